// rt355 — hand-written HIP kernels (gfx950 / MI355X) + C ABI for the per-pixel render path of
// aram-ap/RayTracer (raytracer.py:93-104, 139-140, 152-211, 232-260, 289-325).
//
// This is not a translation of the reference's Python: the recursion is an iterative
// throughput loop, the 13 shadow rays of one shading point are traced together against each
// sphere (shared origin => shared `C - O` and `c` terms, sqrt-free any-hit test), the sphere
// geometry is staged once per workgroup into LDS and read back as wave-uniform broadcasts, and
// pixels are packed to RGB bytes through an LDS tile so global stores are whole dwords.
// DESIGN.md describes the layout and the roofline; include/rt355.h documents the ABI.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>

#include "../../include/rt355.h"

namespace {

// ------------------------------------------------------------------------------------------
// kernel parameters (passed by value: they land in SGPRs via the kernarg segment)
// ------------------------------------------------------------------------------------------
struct DevLights {
  float ppos[3], pcol[3], pint;
  float apos[3], au[3], av[3], acol[3], aint;
  int ns;
};

struct RenderParams {
  rt_camera cam;
  DevLights L;
  int width, height;     // full image, defines the primary rays (raytracer.py:290-295)
  int x0, nx;            // column window [x0, x0+nx)
  int n_rows;            // local rows rendered by this launch
  int tile_rows, tile_first, tile_stride;  // local row lr -> image row, see row_of()
  int y0;                // extra row offset (render_chunk)
  int samples, max_depth, rng_mode;
  uint32_t seed32;
  int n_spheres;
  const float4* geom;    // cx cy cz r^2
  const float4* mat;     // R G B reflection
  const float* spec;     // specular
  uint8_t* out_rgb;      // [n_rows][nx][3] or null
  float* out_f32;        // [n_rows][nx][3] or null (render_chunk)
  int32_t* out_hit;      // [n_rows][nx] or null
  unsigned long long* counters;  // [kCounterSlots][8]
};

constexpr int kBlock = 256;        // 4 waves
constexpr int kTileW = 32;         // pixels per workgroup tile: 32 x 8, one 8x8 square per wave
constexpr int kTileH = 8;
constexpr int kCounterSlots = 64;  // ray counters are spread over 64 cache lines
constexpr int kShadowBatch = 13;   // 1 point-light ray + 12 area-light rays (raytracer.py:80)

// ------------------------------------------------------------------------------------------
// keyed RNG — identical to oracle/ref_harness.py / oracle/rt_oracle.cpp
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t fmix32(uint32_t h) {
  h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
  return h;
}
__device__ __forceinline__ uint32_t mix(uint32_t h, uint32_t v) { return fmix32(h + 0x9E3779B9u + v); }
__device__ __forceinline__ float draw(int mode, uint32_t level_key, uint32_t i) {
  if (mode == RT_RNG_CONSTANT) return 0.5f;
  return float(fmix32(level_key + i * 0x9E3779B9u + 0x7F4A7C15u) >> 8) * (1.0f / 16777216.0f);
}

__device__ __forceinline__ float pow50(float x) {  // `** 50` (raytracer.py:193,208), x >= 0
  float x2 = x * x, x4 = x2 * x2, x8 = x4 * x4, x16 = x8 * x8, x32 = x16 * x16;
  return x32 * x16 * x2;
}

// image row of local row lr (rt_render_rows tiling; full frames use tile_rows = height)
__device__ __forceinline__ int row_of(const RenderParams& p, int lr) {
  int j = lr / p.tile_rows;
  return p.y0 + (p.tile_first + j * p.tile_stride) * p.tile_rows + (lr - j * p.tile_rows);
}

// ------------------------------------------------------------------------------------------
// find_nearest_intersection (raytracer.py:152-172) over the LDS-resident sphere list.
//
// intersect_sphere (raytracer.py:93-104) in a cancellation-free form.  With co = C - O,
// nb = co.D (= -b/2), a = D.D, q = nb/a and the perpendicular offset l = co - q*D:
//     disc/4 = nb^2 - a*c = a*(r^2 - |l|^2)        t = (-b - sqrt(disc))/(2a) = q - sqrt((r^2-|l|^2)/a)
// Evaluating b^2 - 4ac directly in FP32 loses ~|co|^2 * 1e-6 to cancellation (|co| ~ 20 =>
// a silhouette jitter visible as flipped shadow samples on ~0.5 % of pixels); r^2 - |l|^2 keeps
// the error at ~|l| * ulp(|co|).  Strict `<` in index order keeps the lowest index on ties, and
// t == 0 / t < 0 are rejected as the reference's `if t and ...` does (raytracer.py:102,162).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void nearest_hit(const float4* __restrict__ sg, int n, float ox, float oy,
                                            float oz, float dx, float dy, float dz, float inv_a,
                                            float& best_t, int& best_i) {
  best_t = INFINITY;
  best_i = -1;
#pragma unroll 4
  for (int i = 0; i < n; ++i) {
    const float4 s = sg[i];  // wave-uniform address: LDS broadcast read
    const float cx = s.x - ox, cy = s.y - oy, cz = s.z - oz;
    const float q = fmaf(cz, dz, fmaf(cy, dy, cx * dx)) * inv_a;
    const float lx = fmaf(-q, dx, cx), ly = fmaf(-q, dy, cy), lz = fmaf(-q, dz, cz);
    const float dl = fmaf(-lz, lz, fmaf(-ly, ly, fmaf(-lx, lx, s.w)));  // r^2 - |l|^2
    if (dl >= 0.0f) {
      const float t = q - sqrtf(dl * inv_a);
      if (t > 0.0f && t < best_t) {
        best_t = t;
        best_i = i;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// One path: trace_ray (raytracer.py:232-260) as an iterative loop with a throughput
// accumulator:  C_d = (m_d (1-r_d) + r_d C_{d+1}) * l_d   =>
//   acc += T * m_d (1-r_d) * l_d ;  T *= r_d * l_d ;  stop on miss, r_d == 0 or d == D.
// compute_lighting (raytracer.py:174-211) is evaluated in forward order: l_d depends only on the
// hit at level d.
// ------------------------------------------------------------------------------------------
template <int B>
__device__ __forceinline__ void trace_path(const RenderParams& p, const float4* __restrict__ sg,
                                           float ox, float oy, float oz, float dx, float dy, float dz,
                                           uint32_t pixel_key, float& out_r, float& out_g, float& out_b,
                                           int& hit0, uint32_t& n_trace, uint32_t& n_shade) {
  float acc_r = 0.f, acc_g = 0.f, acc_b = 0.f;
  float tp_r = 1.f, tp_g = 1.f, tp_b = 1.f;
  const int n = p.n_spheres;
  const int nrays = 1 + p.L.ns;
  const float inv_ns = 1.0f / float(p.L.ns);
  hit0 = -1;

  for (int level = 0; level <= p.max_depth; ++level) {
    ++n_trace;
    const float inv_a = 1.0f / fmaf(dz, dz, fmaf(dy, dy, dx * dx));  // :95, computed, not assumed 1
    float t;
    int bi;
    nearest_hit(sg, n, ox, oy, oz, dx, dy, dz, inv_a, t, bi);
    if (level == 0) hit0 = bi;
    if (bi < 0) break;  // :237-238 miss -> black
    ++n_shade;

    const float4 s = sg[bi];
    const float4 m = p.mat[bi];
    const float spec = p.spec[bi];
    // hit point and normal (:165-166, :139-140)
    const float px = fmaf(dx, t, ox), py = fmaf(dy, t, oy), pz = fmaf(dz, t, oz);
    float nx = px - s.x, ny = py - s.y, nz = pz - s.z;
    {
      const float inv = 1.0f / sqrtf(fmaf(nz, nz, fmaf(ny, ny, nx * nx)));
      nx *= inv; ny *= inv; nz *= inv;
    }
    // shadow-ray origin (:184, :199) — also the reflection origin (:247)
    const float sx = fmaf(nx, 0.001f, px), sy = fmaf(ny, 0.001f, py), sz = fmaf(nz, 0.001f, pz);
    const float vdn = -(fmaf(dz, nz, fmaf(dy, ny, dx * nx)));  // view . normal, view = -D (:257)

    // ---- compute_lighting: all shadow rays of this point, B at a time -----------------------
    const uint32_t level_key = mix(pixel_key, uint32_t(level + 1));
    float pd = 0.f, ps = 0.f;  // point light: diffuse, specular (unshadowed)
    float ad = 0.f, as = 0.f;  // area light: sums over unshadowed samples
    for (int base = 0; base < nrays; base += B) {
      float Lx[B], Ly[B], Lz[B], Wd[B], Ws[B];
#pragma unroll
      for (int k = 0; k < B; ++k) {
        const int j = base + k;
        float lx, ly, lz;
        if (j == 0) {  // scene['global_light'] (:198)
          lx = p.L.ppos[0]; ly = p.L.ppos[1]; lz = p.L.ppos[2];
        } else {       // area-light sample j-1 (:180-182)
          const float ru = draw(p.rng_mode, level_key, uint32_t(2 * (j - 1))) - 0.5f;
          const float rv = draw(p.rng_mode, level_key, uint32_t(2 * (j - 1) + 1)) - 0.5f;
          lx = fmaf(p.L.av[0], rv, fmaf(p.L.au[0], ru, p.L.apos[0]));
          ly = fmaf(p.L.av[1], rv, fmaf(p.L.au[1], ru, p.L.apos[1]));
          lz = fmaf(p.L.av[2], rv, fmaf(p.L.au[2], ru, p.L.apos[2]));
        }
        lx -= px; ly -= py; lz -= pz;                        // :183 / :198
        const float inv = 1.0f / sqrtf(fmaf(lz, lz, fmaf(ly, ly, lx * lx)));
        lx *= inv; ly *= inv; lz *= inv;
        const float ndl = fmaf(nz, lz, fmaf(ny, ly, nx * lx));
        // view . (L - N*(2 L.N))  (:192-193 / :207-208; the sign of R is the reference's)
        const float vdl = -(fmaf(dz, lz, fmaf(dy, ly, dx * lx)));
        const float vr = fmaf(-2.0f * ndl, vdn, vdl);
        const bool valid = j < nrays;
        Lx[k] = valid ? lx : 0.f; Ly[k] = valid ? ly : 0.f; Lz[k] = valid ? lz : 0.f;
        Wd[k] = valid ? fmaxf(0.f, ndl) : 0.f;                           // :188 / :203
        Ws[k] = (valid && spec > 0.f) ? pow50(fmaxf(0.f, vr)) : 0.f;     // :191-193 / :206-208
      }
      // any-hit scan (:185-187, :200-202 with refraction == 0): the B rays share the origin, so
      // co = C - O and c are per-sphere; hit <=> disc >= 0 and t > 0 <=> r^2 >= |l|^2, nb > 0, c > 0
      // (t > 0 <=> nb > sqrt(disc) <=> nb > 0 and a*c > 0) — no sqrt, no divide.  L is a unit
      // vector (a = 1 +- 2e-7); the residual of q = nb along L enters |l|^2 only to second order.
      bool shadowed[B];
#pragma unroll
      for (int k = 0; k < B; ++k) shadowed[k] = false;
#pragma unroll 2
      for (int i = 0; i < n; ++i) {
        const float4 g = sg[i];
        const float cx = g.x - sx, cy = g.y - sy, cz = g.z - sz;
        const float c = fmaf(cz, cz, fmaf(cy, cy, fmaf(cx, cx, -g.w)));
        const bool cpos = c > 0.0f;
#pragma unroll
        for (int k = 0; k < B; ++k) {
          const float nb = fmaf(cz, Lz[k], fmaf(cy, Ly[k], cx * Lx[k]));
          const float lx = fmaf(-nb, Lx[k], cx), ly = fmaf(-nb, Ly[k], cy), lz = fmaf(-nb, Lz[k], cz);
          const float dl = fmaf(-lz, lz, fmaf(-ly, ly, fmaf(-lx, lx, g.w)));
          shadowed[k] |= (dl >= 0.0f) & (nb > 0.0f) & cpos;
        }
      }
#pragma unroll
      for (int k = 0; k < B; ++k) {
        const float wd = shadowed[k] ? 0.f : Wd[k];
        const float ws = shadowed[k] ? 0.f : Ws[k];
        if (k == 0 && base == 0) { pd = wd; ps = ws; }
        else { ad += wd; as += ws; }
      }
    }
    // :204 + :209 + :189/:194/:196
    const float pk = p.L.pint * pd, psk = p.L.pint * spec * ps;
    const float ak = p.L.aint * ad * inv_ns, ask = p.L.aint * spec * as * inv_ns;
    const float lr = fmaf(m.x * p.L.pcol[0], pk, p.L.pcol[0] * psk) + p.L.acol[0] * (ak + ask);
    const float lg = fmaf(m.y * p.L.pcol[1], pk, p.L.pcol[1] * psk) + p.L.acol[1] * (ak + ask);
    const float lb = fmaf(m.z * p.L.pcol[2], pk, p.L.pcol[2] * psk) + p.L.acol[2] * (ak + ask);

    // :242-249, :258 folded into the accumulator
    const float refl = m.w;
    const float keep = 1.0f - refl;
    acc_r = fmaf(tp_r * (m.x * keep), lr, acc_r);
    acc_g = fmaf(tp_g * (m.y * keep), lg, acc_g);
    acc_b = fmaf(tp_b * (m.z * keep), lb, acc_b);
    if (!(refl > 0.0f)) break;  // :245 no reflection ray
    tp_r *= refl * lr; tp_g *= refl * lg; tp_b *= refl * lb;
    // reflection ray (:246-247), direction not renormalised
    const float dn2 = 2.0f * fmaf(dz, nz, fmaf(dy, ny, dx * nx));
    dx = fmaf(-dn2, nx, dx); dy = fmaf(-dn2, ny, dy); dz = fmaf(-dn2, nz, dz);
    ox = sx; oy = sy; oz = sz;
  }
  out_r = acc_r; out_g = acc_g; out_b = acc_b;
}

// ------------------------------------------------------------------------------------------
// render / render_chunk / render_pixel (raytracer.py:289-325): one thread per pixel, one 8x8
// pixel square per wave, 32x8 per workgroup.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void rt_render_kernel(const RenderParams p) {
  extern __shared__ __align__(16) unsigned char smem[];
  float4* sg = reinterpret_cast<float4*>(smem);
  const int geom_bytes = ((p.n_spheres * 16 + 15) / 16) * 16;
  unsigned char* s_tile = smem + geom_bytes;                       // [8][96] bytes
  unsigned int* s_cnt = reinterpret_cast<unsigned int*>(s_tile + kTileH * kTileW * 3);  // [2]

  const int tid = threadIdx.x;
  for (int i = tid; i < p.n_spheres; i += kBlock) sg[i] = p.geom[i];
  if (tid < 2) s_cnt[tid] = 0u;
  __syncthreads();

  const int wave = tid >> 6, lane = tid & 63;
  const int lx = wave * 8 + (lane & 7), ly = lane >> 3;
  const int col = blockIdx.x * kTileW + lx;   // column inside the window
  const int lrow = blockIdx.y * kTileH + ly;  // local row
  const bool active = (col < p.nx) && (lrow < p.n_rows);

  float cr = 0.f, cg = 0.f, cb = 0.f;
  int hit0 = -1;
  uint32_t n_trace = 0, n_shade = 0;
  if (active) {
    const int x = p.x0 + col;
    const int y = row_of(p, lrow);
    const float fw = float(p.width), fh = float(p.height);
    const float aspect = fw / fh;                                   // :290
    for (int sidx = 0; sidx < p.samples; ++sidx) {                  // :292
      const uint32_t pixel_key = mix(mix(mix(p.seed32, uint32_t(x)), uint32_t(y)), uint32_t(sidx));
      const uint32_t jk = mix(pixel_key, 0u);
      const float u = ((float(x) + draw(p.rng_mode, jk, 0u)) / fw) * 2.0f - 1.0f;            // :293
      const float v = -(((float(y) + draw(p.rng_mode, jk, 1u)) / fh) * 2.0f - 1.0f) / aspect;  // :294
      float dx = fmaf(p.cam.up[0], v, p.cam.right[0] * u) + p.cam.forward[0];
      float dy = fmaf(p.cam.up[1], v, p.cam.right[1] * u) + p.cam.forward[1];
      float dz = fmaf(p.cam.up[2], v, p.cam.right[2] * u) + p.cam.forward[2];
      const float inv = 1.0f / sqrtf(fmaf(dz, dz, fmaf(dy, dy, dx * dx)));                    // :295
      dx *= inv; dy *= inv; dz *= inv;
      float r, g, b;
      trace_path<kShadowBatch>(p, sg, p.cam.origin[0], p.cam.origin[1], p.cam.origin[2], dx, dy, dz,
                               pixel_key, r, g, b, hit0, n_trace, n_shade);
      cr += r; cg += g; cb += b;                                    // :296
    }
    const float inv_s = 1.0f / float(p.samples);                    // :297
    cr = fminf(1.0f, fmaxf(0.0f, cr * inv_s));                      // :318 clamp
    cg = fminf(1.0f, fmaxf(0.0f, cg * inv_s));
    cb = fminf(1.0f, fmaxf(0.0f, cb * inv_s));
    const size_t pix = size_t(lrow) * p.nx + col;
    if (p.out_hit) p.out_hit[pix] = hit0;
    if (p.out_f32) {                                                // render_chunk (:305)
      p.out_f32[pix * 3 + 0] = cr; p.out_f32[pix * 3 + 1] = cg; p.out_f32[pix * 3 + 2] = cb;
    }
  }

  // ---- ray statistics: one pair of atomics per workgroup, spread over kCounterSlots lines ----
  if (n_trace) atomicAdd(&s_cnt[0], n_trace);
  if (n_shade) atomicAdd(&s_cnt[1], n_shade);

  // ---- packed-RGB store (:325  (image*255).astype(uint8): truncation) -------------------------
  if (p.out_rgb) {
    unsigned char* t = s_tile + ly * (kTileW * 3) + lx * 3;
    t[0] = (unsigned char)(cr * 255.0f);
    t[1] = (unsigned char)(cg * 255.0f);
    t[2] = (unsigned char)(cb * 255.0f);
  }
  __syncthreads();
  if (p.out_rgb) {
    const int tile_x = blockIdx.x * kTileW;
    const int rows_here = min(kTileH, p.n_rows - blockIdx.y * kTileH);
    const bool full_w = tile_x + kTileW <= p.nx;
    const bool aligned = ((p.nx & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.out_rgb) & 3) == 0);
    if (full_w && aligned) {
      // 8 rows x 24 dwords; row starts are dword aligned because nx % 4 == 0 and tile_x % 32 == 0
      if (tid < kTileH * 24) {
        const int r = tid / 24, w = tid - r * 24;
        if (r < rows_here) {
          const unsigned int val = *reinterpret_cast<const unsigned int*>(s_tile + r * (kTileW * 3) + w * 4);
          const size_t off = (size_t(blockIdx.y * kTileH + r) * p.nx + tile_x) * 3 + size_t(w) * 4;
          *reinterpret_cast<unsigned int*>(p.out_rgb + off) = val;
        }
      }
    } else {
      for (int b = tid; b < kTileH * kTileW * 3; b += kBlock) {
        const int r = b / (kTileW * 3), cbyte = b - r * (kTileW * 3);
        if (r < rows_here && tile_x * 3 + cbyte < p.nx * 3) {
          p.out_rgb[(size_t(blockIdx.y * kTileH + r) * p.nx + tile_x) * 3 + cbyte] = s_tile[b];
        }
      }
    }
  }
  if (tid == 0 && p.counters) {
    const unsigned int nt = s_cnt[0], nsd = s_cnt[1];
    unsigned long long* slot = p.counters + size_t((blockIdx.x + blockIdx.y * 7u) % kCounterSlots) * 8;
    const int cols_here = min(kTileW, p.nx - int(blockIdx.x) * kTileW);
    const int rows_all = min(kTileH, p.n_rows - int(blockIdx.y) * kTileH);
    if (nt) atomicAdd(slot + 0, (unsigned long long)nt);
    if (nsd) {
      atomicAdd(slot + 1, (unsigned long long)nsd);
      atomicAdd(slot + 3, (unsigned long long)nsd * (unsigned long long)(1 + p.L.ns));
    }
    atomicAdd(slot + 2, (unsigned long long)cols_here * rows_all * p.samples);
  }
}

// out[y] = gathered[rank of y][local row of y]  (rt_scatter_rows)
__global__ __launch_bounds__(256) void rt_scatter_rows_kernel(const uint8_t* __restrict__ gathered,
                                                              long long rank_stride, int n_ranks,
                                                              int width, int height, int tile_rows,
                                                              uint8_t* __restrict__ out, int vec4) {
  const int y = blockIdx.y;
  const int t = y / tile_rows;
  const int r = t % n_ranks, j = t / n_ranks;
  const long long lr = (long long)j * tile_rows + (y - t * tile_rows);
  const long long row_bytes = (long long)width * 3;
  const uint8_t* src = gathered + (long long)r * rank_stride + lr * row_bytes;
  uint8_t* dst = out + (long long)y * row_bytes;
  if (vec4) {
    const int nwords = int(row_bytes / 4);
    for (int w = blockIdx.x * blockDim.x + threadIdx.x; w < nwords; w += gridDim.x * blockDim.x)
      reinterpret_cast<unsigned int*>(dst)[w] = reinterpret_cast<const unsigned int*>(src)[w];
  } else {
    for (long long b = blockIdx.x * blockDim.x + threadIdx.x; b < row_bytes; b += gridDim.x * blockDim.x)
      dst[b] = src[b];
  }
}

thread_local std::string g_last_error;

}  // namespace

// ------------------------------------------------------------------------------------------
// host side: context + C ABI
// ------------------------------------------------------------------------------------------
struct rt_ctx {
  int device = -1;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  float4* d_geom = nullptr;
  float4* d_mat = nullptr;
  float* d_spec = nullptr;
  unsigned long long* d_counters = nullptr;
  int n_spheres = -1;
  DevLights lights{};
  std::string err;
};

namespace {

int fail(rt_ctx* ctx, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (ctx) ctx->err = buf;
  g_last_error = buf;
  return code;
}

#define RT_HIP(ctx, call)                                                                        \
  do {                                                                                           \
    hipError_t e_ = (call);                                                                      \
    if (e_ != hipSuccess) return fail(ctx, RT_EDEVICE, "%s failed: %s", #call, hipGetErrorString(e_)); \
  } while (0)

size_t lds_bytes(int n_spheres) {
  return size_t((n_spheres * 16 + 15) / 16) * 16 + kTileH * kTileW * 3 + 16;
}

struct Window {
  int x0, nx, n_rows, tile_rows, tile_first, tile_stride, y0;
};

int launch(rt_ctx* ctx, const rt_camera* cam, int width, int height, const Window& w, int samples,
           int max_depth, int rng_mode, uint64_t seed, uint8_t* out_rgb, float* out_f32,
           int32_t* out_hit) {
  if (!ctx) return fail(nullptr, RT_EINVAL, "ctx is NULL");
  if (ctx->n_spheres < 0) return fail(ctx, RT_ESTATE, "rt_upload_scene has not been called");
  if (!cam) return fail(ctx, RT_EINVAL, "camera is NULL");
  if (width <= 0 || height <= 0 || width > (1 << 20) || height > (1 << 20))
    return fail(ctx, RT_EINVAL, "width/height out of range: %dx%d", width, height);
  if (samples < 1 || samples > 65536) return fail(ctx, RT_EINVAL, "samples out of range: %d", samples);
  if (max_depth < 0 || max_depth > RT_MAX_DEPTH)
    return fail(ctx, RT_EINVAL, "max_depth out of range [0,%d]: %d", RT_MAX_DEPTH, max_depth);
  if (rng_mode != RT_RNG_CONSTANT && rng_mode != RT_RNG_KEYED)
    return fail(ctx, RT_EINVAL, "unknown rng_mode %d", rng_mode);
  if (!out_rgb && !out_f32 && !out_hit) return fail(ctx, RT_EINVAL, "no output buffer given");
  for (int i = 0; i < 12; ++i)
    if (!std::isfinite((&cam->origin[0])[i])) return fail(ctx, RT_EINVAL, "camera has a non-finite value");
  if (w.n_rows == 0 || w.nx == 0) return RT_OK;  // empty window: nothing to do
  if (w.nx < 0 || w.n_rows < 0) return fail(ctx, RT_EINVAL, "negative window");

  RenderParams p{};
  p.cam = *cam;
  p.L = ctx->lights;
  p.width = width; p.height = height;
  p.x0 = w.x0; p.nx = w.nx; p.n_rows = w.n_rows;
  p.tile_rows = w.tile_rows; p.tile_first = w.tile_first; p.tile_stride = w.tile_stride; p.y0 = w.y0;
  p.samples = samples; p.max_depth = max_depth; p.rng_mode = rng_mode;
  p.seed32 = uint32_t(seed ^ (seed >> 32));
  p.n_spheres = ctx->n_spheres;
  p.geom = ctx->d_geom; p.mat = ctx->d_mat; p.spec = ctx->d_spec;
  p.out_rgb = out_rgb; p.out_f32 = out_f32; p.out_hit = out_hit;
  p.counters = ctx->d_counters;

  const unsigned gx = unsigned((w.nx + kTileW - 1) / kTileW), gy = unsigned((w.n_rows + kTileH - 1) / kTileH);
  if (gy > 65535u * 16u) return fail(ctx, RT_EINVAL, "too many rows for one launch");
  RT_HIP(ctx, hipSetDevice(ctx->device));
  hipLaunchKernelGGL(rt_render_kernel, dim3(gx, gy, 1), dim3(kBlock), lds_bytes(ctx->n_spheres),
                     ctx->stream, p);
  RT_HIP(ctx, hipGetLastError());
  return RT_OK;
}

}  // namespace

extern "C" {

const char* rt_version(void) { return "rt355 0.1 (gfx950)"; }

const char* rt_last_error(const rt_ctx* ctx) { return ctx ? ctx->err.c_str() : g_last_error.c_str(); }

int rt_create(int device_id, rt_ctx** out) {
  if (!out) return fail(nullptr, RT_EINVAL, "out is NULL");
  *out = nullptr;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev == 0)
    return fail(nullptr, RT_EDEVICE, "no HIP device available (%s); rt355 has no CPU path",
                e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
  if (device_id < 0 || device_id >= ndev)
    return fail(nullptr, RT_EINVAL, "device_id %d out of range [0,%d)", device_id, ndev);
  hipDeviceProp_t prop;
  RT_HIP(nullptr, hipGetDeviceProperties(&prop, device_id));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(nullptr, RT_EDEVICE, "device %d is %s; rt355 is built for gfx950 (MI355X) only",
                device_id, prop.gcnArchName);
  rt_ctx* ctx = new (std::nothrow) rt_ctx;
  if (!ctx) return fail(nullptr, RT_EDEVICE, "out of host memory");
  ctx->device = device_id;
  auto bail = [&](hipError_t err, const char* what) {
    int rc = fail(nullptr, RT_EDEVICE, "%s failed: %s", what, hipGetErrorString(err));
    rt_destroy(ctx);
    return rc;
  };
  if ((e = hipSetDevice(device_id)) != hipSuccess) return bail(e, "hipSetDevice");
  if ((e = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking)) != hipSuccess)
    return bail(e, "hipStreamCreate");
  ctx->stream = ctx->own_stream;
  if ((e = hipMalloc(&ctx->d_geom, sizeof(float4) * RT_MAX_SPHERES)) != hipSuccess) return bail(e, "hipMalloc");
  if ((e = hipMalloc(&ctx->d_mat, sizeof(float4) * RT_MAX_SPHERES)) != hipSuccess) return bail(e, "hipMalloc");
  if ((e = hipMalloc(&ctx->d_spec, sizeof(float) * RT_MAX_SPHERES)) != hipSuccess) return bail(e, "hipMalloc");
  if ((e = hipMalloc(&ctx->d_counters, sizeof(unsigned long long) * kCounterSlots * 8)) != hipSuccess)
    return bail(e, "hipMalloc");
  if ((e = hipMemset(ctx->d_counters, 0, sizeof(unsigned long long) * kCounterSlots * 8)) != hipSuccess)
    return bail(e, "hipMemset");
  if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(rt_render_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize,
                               int(lds_bytes(RT_MAX_SPHERES)))) != hipSuccess)
    return bail(e, "hipFuncSetAttribute");
  *out = ctx;
  return RT_OK;
}

void rt_destroy(rt_ctx* ctx) {
  if (!ctx) return;
  if (ctx->device >= 0) (void)hipSetDevice(ctx->device);
  if (ctx->own_stream) { (void)hipStreamSynchronize(ctx->own_stream); (void)hipStreamDestroy(ctx->own_stream); }
  if (ctx->d_geom) (void)hipFree(ctx->d_geom);
  if (ctx->d_mat) (void)hipFree(ctx->d_mat);
  if (ctx->d_spec) (void)hipFree(ctx->d_spec);
  if (ctx->d_counters) (void)hipFree(ctx->d_counters);
  delete ctx;
}

int rt_set_stream(rt_ctx* ctx, void* hip_stream, int use_own_stream) {
  if (!ctx) return fail(nullptr, RT_EINVAL, "ctx is NULL");
  // hip_stream == NULL with use_own_stream == 0 is HIP's legacy default stream (what
  // torch.cuda.current_stream().cuda_stream is unless the caller switched streams).
  ctx->stream = use_own_stream ? ctx->own_stream : static_cast<hipStream_t>(hip_stream);
  return RT_OK;
}

int rt_synchronize(rt_ctx* ctx) {
  if (!ctx) return fail(nullptr, RT_EINVAL, "ctx is NULL");
  RT_HIP(ctx, hipSetDevice(ctx->device));
  RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return RT_OK;
}

int rt_upload_scene(rt_ctx* ctx, const float* geom4, const float* mat4, const float* spec,
                    int n_spheres, const rt_lights* lights) {
  if (!ctx) return fail(nullptr, RT_EINVAL, "ctx is NULL");
  if (n_spheres < 0 || n_spheres > RT_MAX_SPHERES)
    return fail(ctx, RT_EINVAL, "n_spheres %d out of range [0,%d]", n_spheres, RT_MAX_SPHERES);
  if (n_spheres > 0 && (!geom4 || !mat4 || !spec)) return fail(ctx, RT_EINVAL, "scene array is NULL");
  if (!lights) return fail(ctx, RT_EINVAL, "lights is NULL");
  if (lights->area_samples < 1 || lights->area_samples > RT_MAX_AREA_SAMPLES)
    return fail(ctx, RT_EINVAL, "area_samples %d out of range [1,%d]", lights->area_samples,
                RT_MAX_AREA_SAMPLES);
  for (int i = 0; i < n_spheres * 4; ++i)
    if (!std::isfinite(geom4[i]) || !std::isfinite(mat4[i]))
      return fail(ctx, RT_EINVAL, "scene has a non-finite value (sphere %d)", i / 4);
  for (int i = 0; i < n_spheres; ++i) {
    if (!std::isfinite(spec[i])) return fail(ctx, RT_EINVAL, "scene has a non-finite specular (sphere %d)", i);
    if (geom4[4 * i + 3] < 0.f) return fail(ctx, RT_EINVAL, "sphere %d has negative radius^2", i);
  }
  const float* lf = lights->point_pos;
  for (int i = 0; i < 20; ++i)
    if (!std::isfinite(lf[i])) return fail(ctx, RT_EINVAL, "lights have a non-finite value");
  RT_HIP(ctx, hipSetDevice(ctx->device));
  // the previous scene may still be in use by launches in flight on the stream
  RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (n_spheres > 0) {
    RT_HIP(ctx, hipMemcpy(ctx->d_geom, geom4, sizeof(float) * 4 * n_spheres, hipMemcpyHostToDevice));
    RT_HIP(ctx, hipMemcpy(ctx->d_mat, mat4, sizeof(float) * 4 * n_spheres, hipMemcpyHostToDevice));
    RT_HIP(ctx, hipMemcpy(ctx->d_spec, spec, sizeof(float) * n_spheres, hipMemcpyHostToDevice));
  }
  DevLights& L = ctx->lights;
  memcpy(L.ppos, lights->point_pos, 12); memcpy(L.pcol, lights->point_color, 12);
  L.pint = lights->point_intensity;
  memcpy(L.apos, lights->area_pos, 12); memcpy(L.au, lights->area_u, 12); memcpy(L.av, lights->area_v, 12);
  memcpy(L.acol, lights->area_color, 12);
  L.aint = lights->area_intensity; L.ns = lights->area_samples;
  ctx->n_spheres = n_spheres;
  return RT_OK;
}

int rt_render(rt_ctx* ctx, const rt_camera* cam, int width, int height, int samples, int max_depth,
              int rng_mode, uint64_t seed, uint8_t* out_rgb, int32_t* out_hit) {
  if (!out_rgb) return fail(ctx, RT_EINVAL, "out_rgb is NULL");
  Window w{0, width, height, height > 0 ? height : 1, 0, 1, 0};
  return launch(ctx, cam, width, height, w, samples, max_depth, rng_mode, seed, out_rgb, nullptr, out_hit);
}

int rt_render_rows(rt_ctx* ctx, const rt_camera* cam, int width, int height, int samples,
                   int max_depth, int rng_mode, uint64_t seed, int tile_rows, int tile_first,
                   int tile_stride, uint8_t* out_rgb, int32_t* out_hit, int* n_local_rows_out) {
  if (n_local_rows_out) *n_local_rows_out = 0;
  if (tile_rows < 1 || tile_stride < 1 || tile_first < 0 || tile_first >= tile_stride)
    return fail(ctx, RT_EINVAL, "bad tiling: tile_rows=%d tile_first=%d tile_stride=%d", tile_rows,
                tile_first, tile_stride);
  if (width <= 0 || height <= 0) return fail(ctx, RT_EINVAL, "width/height out of range: %dx%d", width, height);
  // rows owned: tiles t = tile_first + j*tile_stride with t*tile_rows < height
  long long rows = 0;
  const int n_tiles = (height + tile_rows - 1) / tile_rows;
  for (int t = tile_first; t < n_tiles; t += tile_stride) {
    const int r0 = t * tile_rows;
    rows += (r0 + tile_rows <= height) ? tile_rows : (height - r0);
  }
  if (n_local_rows_out) *n_local_rows_out = int(rows);
  if (rows == 0) return RT_OK;
  if (!out_rgb) return fail(ctx, RT_EINVAL, "out_rgb is NULL");
  Window w{0, width, int(rows), tile_rows, tile_first, tile_stride, 0};
  return launch(ctx, cam, width, height, w, samples, max_depth, rng_mode, seed, out_rgb, nullptr, out_hit);
}

int rt_render_chunk(rt_ctx* ctx, const rt_camera* cam, int x_start, int x_end, int y_start, int y_end,
                    int width, int height, int samples, int max_depth, int rng_mode, uint64_t seed,
                    float* out_rgbf) {
  if (x_end < x_start || y_end < y_start)
    return fail(ctx, RT_EINVAL, "empty/negative chunk [%d,%d)x[%d,%d)", x_start, x_end, y_start, y_end);
  if (x_end > x_start && y_end > y_start && !out_rgbf) return fail(ctx, RT_EINVAL, "out_rgbf is NULL");
  const int rows = y_end - y_start;
  // pixel coordinates outside the image are legal in the reference (render_pixel just evaluates
  // the formula), so no clipping against width/height here either.
  Window w{x_start, x_end - x_start, rows, rows > 0 ? rows : 1, 0, 1, y_start};
  if (rows == 0 || x_end == x_start) return RT_OK;
  return launch(ctx, cam, width, height, w, samples, max_depth, rng_mode, seed, nullptr, out_rgbf, nullptr);
}

int rt_render_frames(rt_ctx* ctx, const rt_camera* cams, const uint64_t* seeds, int n_frames,
                     int width, int height, int samples, int max_depth, int rng_mode, uint8_t* out_rgb) {
  if (n_frames < 0) return fail(ctx, RT_EINVAL, "n_frames < 0");
  if (n_frames == 0) return RT_OK;
  if (!cams || !seeds || !out_rgb) return fail(ctx, RT_EINVAL, "cams/seeds/out_rgb is NULL");
  const size_t frame_bytes = size_t(width) * size_t(height) * 3;
  for (int f = 0; f < n_frames; ++f) {
    int rc = rt_render(ctx, &cams[f], width, height, samples, max_depth, rng_mode, seeds[f],
                       out_rgb + frame_bytes * f, nullptr);
    if (rc != RT_OK) return rc;
  }
  return RT_OK;
}

int rt_scatter_rows(rt_ctx* ctx, const uint8_t* gathered, int64_t rank_stride_bytes, int n_ranks,
                    int width, int height, int tile_rows, uint8_t* out_rgb) {
  if (!ctx) return fail(nullptr, RT_EINVAL, "ctx is NULL");
  if (!gathered || !out_rgb) return fail(ctx, RT_EINVAL, "gathered/out_rgb is NULL");
  if (n_ranks < 1 || width <= 0 || height <= 0 || tile_rows < 1 || height > 65535 * 16)
    return fail(ctx, RT_EINVAL, "bad scatter arguments");
  // every rank's slab must hold its rows
  const int n_tiles = (height + tile_rows - 1) / tile_rows;
  const long long max_rows = (long long)((n_tiles + n_ranks - 1) / n_ranks) * tile_rows;
  if (rank_stride_bytes < 0 || (n_ranks > 1 && rank_stride_bytes < max_rows * width * 3 &&
                                rank_stride_bytes < (long long)height * width * 3))
    return fail(ctx, RT_EINVAL, "rank_stride_bytes %lld too small", (long long)rank_stride_bytes);
  const long long row_bytes = (long long)width * 3;
  const int vec4 = (row_bytes % 4 == 0) && (rank_stride_bytes % 4 == 0) &&
                   ((reinterpret_cast<uintptr_t>(gathered) | reinterpret_cast<uintptr_t>(out_rgb)) % 4 == 0);
  const int per_row = int(vec4 ? row_bytes / 4 : row_bytes);
  const unsigned gx = unsigned(std::min(8, (per_row + 255) / 256));
  RT_HIP(ctx, hipSetDevice(ctx->device));
  hipLaunchKernelGGL(rt_scatter_rows_kernel, dim3(gx, unsigned(height), 1), dim3(256), 0, ctx->stream,
                     gathered, (long long)rank_stride_bytes, n_ranks, width, height, tile_rows, out_rgb, vec4);
  RT_HIP(ctx, hipGetLastError());
  return RT_OK;
}

int rt_get_ray_counts(rt_ctx* ctx, uint64_t out[4]) {
  if (!ctx || !out) return fail(ctx, RT_EINVAL, "ctx/out is NULL");
  unsigned long long host[kCounterSlots * 8];
  RT_HIP(ctx, hipSetDevice(ctx->device));
  RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
  RT_HIP(ctx, hipMemcpy(host, ctx->d_counters, sizeof host, hipMemcpyDeviceToHost));
  unsigned long long trace = 0, shade = 0, primary = 0, shadow = 0;
  for (int s = 0; s < kCounterSlots; ++s) {
    trace += host[s * 8]; shade += host[s * 8 + 1]; primary += host[s * 8 + 2]; shadow += host[s * 8 + 3];
  }
  out[0] = primary;
  out[1] = trace - primary;
  out[2] = shadow;
  out[3] = shade;
  return RT_OK;
}

int rt_reset_ray_counts(rt_ctx* ctx) {
  if (!ctx) return fail(nullptr, RT_EINVAL, "ctx is NULL");
  RT_HIP(ctx, hipSetDevice(ctx->device));
  RT_HIP(ctx, hipMemsetAsync(ctx->d_counters, 0, sizeof(unsigned long long) * kCounterSlots * 8, ctx->stream));
  return RT_OK;
}

}  // extern "C"
