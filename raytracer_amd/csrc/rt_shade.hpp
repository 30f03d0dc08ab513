// rt_shade.hpp — part of the ONE translation unit rt355.hip (included there, inside its anonymous namespace; not a
// header for anyone else).  compute_lighting (raytracer.py:174-211) and one level of trace_ray (:240-258) per shaded hit; primary rays (:289-295),
// quantisation (:318, :325) and the tile stores shared by the kernels.
#pragma once

// ------------------------------------------------------------------------------------------
// compute_lighting (raytracer.py:174-211) at one hit: point light + L.ns area-light samples,
// B shadow rays at a time.  Returns the light colour l (to be multiplied into the surface colour).
//
// Any-hit scan (:185-187, :200-202 with refraction == 0).  The B rays share their origin S, so
// co = C - S and c = |co|^2 - r^2 are per-sphere; per (ray, sphere):
//     hit <=> disc >= 0 and t > 0 <=> r^2 >= |l|^2 and nb > 0 and c > 0        (no sqrt, no divide)
// (t > 0 <=> nb > sqrt(disc) <=> nb > 0 and a*c > 0; with c > 0 and nb == 0, r^2 - |l|^2 = -c < 0,
// so `nb >= 0` is equivalent).  c <= 0 is folded in by replacing r^2 with -inf for that sphere, and
// the verdict is accumulated branch-free as  m_k = max(m_k, min(r^2 - |l|^2, nb))  (shadowed <=>
// m_k >= 0).  L is a unit vector (a = 1 +- 2e-7): the residual of q = nb along L enters |l|^2 only
// to second order.
// ------------------------------------------------------------------------------------------
#if RT_CULL_STATS
static __device__ unsigned long long g_dbg[4];
#define g_dbg_rel dbg_rel_
#define g_dbg_lanes dbg_lanes_
#define g_dbg_iters dbg_iters_
#define g_dbg_turns dbg_turns_
static __device__ __forceinline__ void dbg_flush(unsigned r, unsigned l, unsigned i, unsigned t) {
  atomicAdd(&g_dbg[0], (unsigned long long)r); atomicAdd(&g_dbg[1], (unsigned long long)l); atomicAdd(&g_dbg[2], (unsigned long long)i);
  atomicAdd(&g_dbg[3], (unsigned long long)t);
}
#endif
// What the shadow rays of one shading point leave behind: unshadowed diffuse / specular weights of the point
// light and their sums over the area light's samples.  light_colour() turns them into the light colour once
// the material is at hand — the stream kernel fetches it behind the shadow loops, so that it holds no
// registers across them.
struct LightSums {
  float pd, ps, ad, as;
  float rx, ry, rz;   // D - 2 (D.N) N: the mirrored direction (:246), which lighting() forms for the specular terms anyway
};
__device__ __forceinline__ void light_colour(const DevLights& L, const LightSums& w, float mr, float mg, float mb,
                                             float spec, float& lr, float& lg, float& lb) {
  float ps = w.ps, as = w.as;
  if (!(spec > 0.0f)) { ps = 0.f; as = 0.f; }                        // :191 / :206
  // :204 + :209 + :189/:194/:196
  const float inv_ns = L.inv_ns;
  const float pk = L.pint * w.pd, psk = L.pint * spec * ps;
  const float ak = L.aint * w.ad * inv_ns, ask = L.aint * spec * as * inv_ns;
  lr = fmaf(mr * L.pcol[0], pk, L.pcol[0] * psk) + L.acol[0] * (ak + ask);
  lg = fmaf(mg * L.pcol[1], pk, L.pcol[1] * psk) + L.acol[1] * (ak + ask);
  lb = fmaf(mb * L.pcol[2], pk, L.pcol[2] * psk) + L.acol[2] * (ak + ask);
}

// (ALL: the caller has checked that the lights ask for exactly the reference's 1 + 12 rays, so the batches and their rays are known
// when the kernel is compiled: no ray needs the "beyond the light's sample count" guard — three scalar instructions and a select
// per ray — and where two batches of 7 hold the 13 rays, the fourteenth does not exist)
template <int MODE, int B, bool REFR = false, bool CUBES = false, bool LIST = false, bool ALL = false>
__device__ __forceinline__ LightSums lighting(const DevLights& L, const float4* __restrict__ sg,
                                         const float4* __restrict__ gg, const float2* __restrict__ refr,
                                         const uint32_t* __restrict__ glass_bits,
                                         const uint32_t* __restrict__ occl_base, uint32_t occl_off, const uint2* __restrict__ occl_items, int n, int nc, float px, float py, float pz, float nx, float ny, float nz,
                                         float dx, float dy, float dz, float sx, float sy, float sz,
                                         uint32_t level_key) {
#if RT_CULL_STATS
  unsigned dbg_rel_ = 0, dbg_lanes_ = 0, dbg_iters_ = 0, dbg_turns_ = 0;
#endif
  const int nrays = ALL ? kReferenceShadowRays : 1 + L.ns;
  [[maybe_unused]] const float vdn_early = LIST ? 0.f : -(fmaf(dz, nz, fmaf(dy, ny, dx * nx)));
  float pd = 0.f, ps = 0.f;  // point light: diffuse, specular (unshadowed)
  float ad = 0.f, as = 0.f;  // area light: sums over unshadowed samples
  // Per point, not per ray: the area light's corner seen from P — sample j is  A + u (r_u - 0.5) + v (r_v - 0.5) - P
  // = (A - P - 1.5 (u + v)) + u (1 + r_u) + v (1 + r_v), and the draws come as 1 + r — and the mirrored view vector:
  // view . (L - N (2 L.N))  (:192-193 / :207-208) = -(D - 2 (D.N) N) . L.
  // (explicit fused multiply-adds: left to the compiler, `a - 1.5 b` is contracted in one inlining context and not in
  // another, and the stream and pixel kernels then differ in a last bit somewhere — found by tools/fuzz_culls.py)
  const float cax = fmaf(-1.5f, L.au[0] + L.av[0], L.apos[0] - px), cay = fmaf(-1.5f, L.au[1] + L.av[1], L.apos[1] - py),
              caz = fmaf(-1.5f, L.au[2] + L.av[2], L.apos[2] - pz);
  // Rays in several batches (REFR, scenes with boxes): the spheres 0..63 / boxes 0..31 that the FIRST batch's culls let through.  That
  // batch holds the point-light ray, so its survivors are a superset of every later batch's: the later batches test these
  // without culling again, and a ray that hits objects of BOTH kinds finds its nearest hit among them (no other object can
  // be hit by any of the point's shadow rays).
  constexpr bool SHARE = REFR || B < kReferenceShadowRays;     // (the reference's 13 rays in one batch: nothing to share, no registers held)
  [[maybe_unused]] uint32_t kept0 = 0u, kept1 = 0u, keptc = 0u;
  // One batch of B rays at a time (rt_shade_batch.hpp: the text of the batch, in a scope that defines `base`).
  if constexpr (ALL && B < kReferenceShadowRays) {
    static_assert(2 * B >= kReferenceShadowRays, "at most two batches");
    {
      constexpr int base = 0;
#include "rt_shade_batch.hpp"
    }
    {
      constexpr int base = B;
#include "rt_shade_batch.hpp"
    }
  } else {
    for (int base = 0; base < nrays; base += B) {      // (ALL with one batch: nrays == B, one turn)
#include "rt_shade_batch.hpp"
    }
  }
#if RT_CULL_STATS
  dbg_flush(dbg_rel_, dbg_lanes_, dbg_iters_, dbg_turns_);
#endif
  const float dn2 = 2.0f * fmaf(dz, nz, fmaf(dy, ny, dx * nx));
  return LightSums{pd, ps, ad, as, fmaf(-dn2, nx, dx), fmaf(-dn2, ny, dy), fmaf(-dn2, nz, dz)};
}

// One shaded hit (trace_ray :240-258 without the recursion): hit point P on sphere `idx`, incoming
// direction D.  Folds the level into the throughput accumulator
//     C_d = (m_d (1-r_d) + r_d C_{d+1}) * l_d   =>   acc += T * m_d (1-r_d) * l_d ;  T *= r_d * l_d
// and produces the reflection ray (:246-247, direction not renormalised).  Returns true if the
// path continues (r_d > 0).
// `late.request(idx)` is called before the lighting loops and `late.load(m, spec)` after them, before T / acc
// are touched: the stream kernel has the material fetched into LDS meanwhile and keeps T, acc and the other
// fields lighting does not need in its LDS record until then, so they do not occupy registers — or scratch —
// across the 13 x N shadow tests.
struct MaterialInRegisters {   // the pixel kernel: plain loads, issued early
  const RenderParams& p;
  float4 m;
  float spec;
  __device__ __forceinline__ void request(int idx) { m = p.mat[idx]; spec = p.spec[idx]; }
  __device__ __forceinline__ void load(float4& m_out, float& spec_out) { m_out = m; spec_out = spec; }
};
template <int MODE, bool CUBES = false, bool LIST = false, typename LateState>
__device__ __forceinline__ bool shade_hit(const RenderParams& p, const float4* __restrict__ sg, int idx,
                                          float px, float py, float pz, float& dx, float& dy, float& dz,
                                          uint32_t level_key, float& tr, float& tg, float& tb,
                                          float& ar, float& ag, float& ab, float& ox, float& oy,
                                          float& oz, LateState& late) {
  late.request(idx);
  float nx, ny, nz;
  surface_normal<CUBES>(sg, p.n_spheres, idx, px, py, pz, nx, ny, nz);   // :139-140 / :142-150
  // shadow-ray origin (:184, :199) — also the reflection origin (:247)
  const float sx = fmaf(nx, 0.001f, px), sy = fmaf(ny, 0.001f, py), sz = fmaf(nz, 0.001f, pz);
  // (the row of the occluder table as a wave-uniform base and a 32-bit offset of the lane's own: the 64-bit address, formed here,
  // was two registers live through the ray set-up)
  // (the list kernels keep the address: with the offset form they spill 7 registers instead of 3, config 4 -4 %)
  const uint32_t* occl_base = LIST ? p.occl_start + occluder_row(p.occl_list_dim, idx, nx, ny, nz) : p.occl;
  const uint32_t occl_off = LIST ? 0u : uint32_t(occluder_row(p.occl_dim, idx, nx, ny, nz)) * uint32_t(p.occl_words);
  constexpr int kB = CUBES ? kCubeShadowBatch : kShadowBatch;
  LightSums w;
  if (RT_LIGHT_ALL && 2 * kB >= kReferenceShadowRays && 1 + p.L.ns == kReferenceShadowRays)      // (wave-uniform) the reference's lights: all rays in one batch, none of them idle
    w = lighting<MODE, kB, false, CUBES, LIST, true>(p.L, sg, p.geom, nullptr, nullptr, occl_base, occl_off, p.occl_items, p.n_spheres, p.n_cubes,
                                                       px, py, pz, nx, ny, nz, dx, dy, dz, sx, sy, sz, level_key);
  else
    w = lighting<MODE, kB, false, CUBES, LIST>(p.L, sg, p.geom, nullptr, nullptr, occl_base, occl_off, p.occl_items, p.n_spheres, p.n_cubes,
                                                px, py, pz, nx, ny, nz, dx, dy, dz, sx, sy, sz, level_key);
  float4 m;
  float spec;
  late.load(m, spec);
  float lr, lg, lb;
  light_colour(p.L, w, m.x, m.y, m.z, spec, lr, lg, lb);
  const float refl = m.w;
  const float keep = 1.0f - refl;                                    // :249
  ar = fmaf(tr * (m.x * keep), lr, ar);
  ag = fmaf(tg * (m.y * keep), lg, ag);
  ab = fmaf(tb * (m.z * keep), lb, ab);
  if (!(refl > 0.0f)) return false;                                  // :245
  tr *= refl * lr; tg *= refl * lg; tb *= refl * lb;
  dx = w.rx; dy = w.ry; dz = w.rz;                                   // :246 (formed in lighting(): the same expression)
  ox = sx; oy = sy; oz = sz;                                         // :247
  return true;
}

// primary ray of pixel (x, y), sample `sidx` (render_pixel, raytracer.py:290-295)
template <int MODE>
__device__ __forceinline__ void primary_ray(const RenderParams& p, const rt_camera& cam, uint32_t seed32,
                                            int x, int y, int sidx, uint32_t& pixel_key, float& dx,
                                            float& dy, float& dz) {
  // (x + r) / W and (y + r) / H stay true (correctly rounded) quotients: with a plain reciprocal
  // multiply the centre column of an odd-width frame is no longer exactly u = 0, and mirror-to-mirror
  // paths amplify that 3e-8 by ~10x per bounce (tests: facing mirrors).  1/aspect is a multiply.
  const float fw = float(p.width), fh = float(p.height);
  const float inv_aspect = p.inv_aspect;                                                      // :290
  pixel_key = pixel_key_of(seed32, uint32_t(x), uint32_t(y), uint32_t(sidx));
  float ju1, jv1;
  draw_pair1<MODE>(pixel_key, 0u, ju1, jv1);                    // the two jitter draws: the pixel's own stream
  const float u = div_exact(float(x) + (ju1 - 1.0f), fw, p.inv_w) * 2.0f - 1.0f;                  // :293
  const float v = -(div_exact(float(y) + (jv1 - 1.0f), fh, p.inv_h) * 2.0f - 1.0f) * inv_aspect;  // :294
  dx = fmaf(cam.up[0], v, cam.right[0] * u) + cam.forward[0];
  dy = fmaf(cam.up[1], v, cam.right[1] * u) + cam.forward[1];
  dz = fmaf(cam.up[2], v, cam.right[2] * u) + cam.forward[2];
  const float inv = frsq(fmaf(dz, dz, fmaf(dy, dy, dx * dx)));                                // :295
  dx *= inv; dy *= inv; dz *= inv;
}

__device__ __forceinline__ uint32_t pack_rgbx(float r, float g, float b) {
  // :318 clamp, :325 (image*255).astype(uint8): truncation toward zero
  r = fminf(1.0f, fmaxf(0.0f, r)); g = fminf(1.0f, fmaxf(0.0f, g)); b = fminf(1.0f, fmaxf(0.0f, b));
  return uint32_t(r * 255.0f) | (uint32_t(g * 255.0f) << 8) | (uint32_t(b * 255.0f) << 16);
}

// Finished sample of a pixel: fold it into the running mean over samples (render_pixel :296-297)
// and, on the last sample, emit the pixel.  `gpix` indexes out_*, `apix` accum (frame offsets included).
__device__ __forceinline__ void finish_pixel(const RenderParams& p, long long gpix, long long apix,
                                             float r, float g, float b, uint32_t* __restrict__ stage_slot) {
  if (p.n_samples > 1) {
    float* a = p.accum + apix * 3;
    if (p.sample_index > 0) { r += a[0]; g += a[1]; b += a[2]; }       // color = color + trace (:296)
    if (p.sample_index + 1 < p.n_samples) { a[0] = r; a[1] = g; a[2] = b; return; }
    const float inv_s = p.inv_samples;                                  // :297
    r *= inv_s; g *= inv_s; b *= inv_s;
  }
  if (p.out_f32) {                                                      // render_chunk (:305)
    float* o = p.out_f32 + gpix * 3;
    o[0] = fminf(1.0f, fmaxf(0.0f, r)); o[1] = fminf(1.0f, fmaxf(0.0f, g)); o[2] = fminf(1.0f, fmaxf(0.0f, b));
  }
  const uint32_t rgbx = pack_rgbx(r, g, b);
  if (stage_slot) {
    *stage_slot = rgbx;
  } else if (p.out_rgb) {   // stream kernel: only hit pixels end here, scattered over the frame
    uint8_t* o = p.out_rgb + gpix * 3;
    o[0] = uint8_t(rgbx); o[1] = uint8_t(rgbx >> 8); o[2] = uint8_t(rgbx >> 16);
  }
}

__device__ __forceinline__ void flush_counters(const RenderParams& p, unsigned int* s_cnt, unsigned slot_id,
                                               unsigned long long primaries) {
  const unsigned int nt = s_cnt[0], nsd = s_cnt[1];
  unsigned long long* slot = p.counters + size_t(slot_id % kCounterSlots) * kCounterWords;
  if (nt) atomicAdd(slot + 0, (unsigned long long)nt);
  if (nsd) {
    atomicAdd(slot + 1, (unsigned long long)nsd);
    atomicAdd(slot + 3, (unsigned long long)nsd * (unsigned long long)(1 + p.L.ns));
  }
  if (primaries) atomicAdd(slot + 2, primaries);
}

// Store a staged tile of `rows` x `cols` pixels (LDS, one RGBX word per pixel, row pitch
// `pitch` words) as packed RGB bytes at out[(row0 + r) * nx + col0 + c].  Whole dwords when the
// tile is full width and the rows are dword aligned, bytes otherwise.
template <int TILE_W>
__device__ __forceinline__ void store_tile_rgb(uint8_t* __restrict__ out, const uint32_t* __restrict__ stage,
                                               int nx, int row0, int col0, int rows, int cols, int tid) {
  const bool aligned = (cols == TILE_W) && ((nx & 3) == 0) && ((reinterpret_cast<uintptr_t>(out) & 3) == 0);
  if (aligned) {
    constexpr int kWordsPerRow = TILE_W * 3 / 4;
    for (int i = tid; i < rows * kWordsPerRow; i += kBlock) {
      const int r = i / kWordsPerRow, w = i - r * kWordsPerRow;
      uint32_t val = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int byte = 4 * w + k, c = byte / 3, ch = byte - 3 * c;
        val |= ((stage[r * TILE_W + c] >> (8 * ch)) & 0xFFu) << (8 * k);
      }
      *reinterpret_cast<uint32_t*>(out + (size_t(row0 + r) * nx + col0) * 3 + size_t(w) * 4) = val;
    }
  } else {
    for (int i = tid; i < rows * cols * 3; i += kBlock) {
      const int r = i / (cols * 3), byte = i - r * (cols * 3), c = byte / 3, ch = byte - 3 * c;
      out[(size_t(row0 + r) * nx + col0) * 3 + byte] = uint8_t(stage[r * TILE_W + c] >> (8 * ch));
    }
  }
}

// The kernel's argument block, re-read where it is used.  A persistent kernel that takes RenderParams by value
// has the compiler load all of it at entry and carry it — ~70 scalar registers — through the whole loop: the
// stream kernel spilled 148 SGPRs (and, for want of the VGPRs that hold them, 22 VGPRs to scratch) that way.
// The block lies in the kernarg segment, which is constant memory: params_here() hands out a reference to it
// whose pointer the compiler cannot see through, so fields are fetched by scalar loads after the call and are
// dead at the end of the phase that called it.  (The kernel's parameter itself is never touched.)
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ const RenderParams& params_here() {
  typedef const RenderParams __attribute__((address_space(4)))* kparams_t;
  kparams_t q = (kparams_t)__builtin_amdgcn_kernarg_segment_ptr();   // RenderParams is the only kernel argument: offset 0
  asm volatile("" : "+s"(q));
  return *(const RenderParams*)q;
}
#else   // host pass only parses the kernels
__device__ __forceinline__ const RenderParams& params_here() { static RenderParams host_never; return host_never; }
#endif

// One 8x8 tile of packed RGB (render :318, :325 -> the array glTexImage2D takes, :346).  Lane (lane & 7, lane >> 3)
// holds pixel (col0 + lx, row0 + ly) as an RGBX word.  A full-width tile of a dword-aligned frame leaves as whole
// dwords: lane l < 48 stores dword w = l % 6 of tile row r = l / 6 — bytes 4w..4w+3 of the row's 24 come from
// pixels a = 4w / 3 and a + 1 (RGBX words of lanes 8r + a, 8r + a + 1, fetched with ds_bpermute: the LDS
// crossbar, no LDS memory).  Anything else goes byte by byte.
__device__ __forceinline__ void store_tile8(uint8_t* __restrict__ out, int nx, int row0, int col0, int rows, int cols,
                                            int lane, uint32_t rgbx, bool dwords) {
  if (cols == 8 && dwords) {
    const int r = (lane * 43) >> 8, w = lane - 6 * r;            // l / 6 for l < 64
    const int hi3 = w >= 3 ? 1 : 0, a = w + hi3, m = w - 3 * hi3;
    const int src = (8 * r + a) & 63;
    const uint32_t lo = __builtin_amdgcn_ds_bpermute(src * 4, int(rgbx));
    const uint32_t hi = __builtin_amdgcn_ds_bpermute(((src + 1) & 63) * 4, int(rgbx));
    const unsigned long long six = (unsigned long long)lo | ((unsigned long long)hi << 24);
    if (lane < 48 && r < rows)
      *reinterpret_cast<uint32_t*>(out + ((size_t)(row0 + r) * nx + col0) * 3 + size_t(w) * 4) = uint32_t(six >> (8 * m));
  } else if ((lane & 7) < cols && (lane >> 3) < rows) {
    uint8_t* o = out + ((size_t)(row0 + (lane >> 3)) * nx + col0 + (lane & 7)) * 3;
    o[0] = uint8_t(rgbx); o[1] = uint8_t(rgbx >> 8); o[2] = uint8_t(rgbx >> 16);
  }
}
