// rt355 — hand-written HIP kernels (gfx950 / MI355X) + C ABI for the per-pixel render path of
// aram-ap/RayTracer (raytracer.py:93-104, 139-140, 152-211, 232-260, 289-325).
//
// This is not a translation of the reference's Python:
//   * primary rays and shading are two kernels: a visibility pass (four pixels per lane, per-wave
//     sphere culling for large scenes) writes per-tile hit indices and occupancy bits; persistent
//     waves then shade — recursion became an iterative throughput accumulator, and further per-wave
//     work lists in LDS: primary hits and reflected hits are compacted into dense 64-lane passes,
//     so the shadow-ray loop (13 of every 14 rays) runs with full waves whatever the hit pattern;
//   * the 13 shadow rays of one shading point are traced together (shared origin => shared
//     `C - O`, sqrt-free any-hit test, sign-bit verdicts), and only against the few spheres that a
//     per-object table (FP64, scene upload) and a per-point cone test leave as candidates;
//   * sphere geometry is staged once per workgroup into LDS; up to RT_FRAMES_PER_LAUNCH (1024) frames share
//     one launch pair so that the persistent kernel's fixed cost is paid once;
//   * finished pixels leave as whole 8x8 tiles (per-wave tile slots in LDS), and up to 64 samples per pixel are
//     shaded by one launch of the persistent kernel, summed per tile in LDS.
// DESIGN.md describes the layout and the roofline; include/rt355.h documents the ABI.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <atomic>
#include <mutex>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include "../../include/rt355.h"

#ifndef RT_CULL_STATS
#define RT_CULL_STATS 0   // diagnostic build: candidate / drain-turn counts of the culled shadow scan (tools/stamps.py)
#endif
#ifndef RT_STAMPS
#define RT_STAMPS 0       // diagnostic build: per-phase s_memtime sums into counters[.][8..15]
#endif
#ifndef RT_WAVES_PER_SIMD
#define RT_WAVES_PER_SIMD 4   // __launch_bounds__ occupancy target: caps VGPRs at 512/4 = 128
#endif
#ifndef RT_CUBE_SHADOW_CULL
#define RT_CUBE_SHADOW_CULL 1   // lighting(): per-lane candidate boxes for the shadow rays (0: all boxes, for fuzzing)
#endif

namespace {
#include "rt_core.hpp"
#include "rt_shade.hpp"
#include "rt_visibility.hpp"
#include "rt_stream.hpp"
#include "rt_kernels_misc.hpp"

thread_local std::string g_last_error;

}  // namespace

// rt_render_noalloc: everything rt_render's kernels keep in device memory besides the frame itself — the frame
// record, the primary-ray table, the stream kernel's work counters and tile occupancy bits, the visibility
// pass's tile records — lies in this array instead of buffers grown with hipMalloc.  One copy per GPU, held
// until the process ends.  Host code never uses the array's name as an address: rt_create asks HIP for the
// address on its device (hipGetSymbolAddress) and the kernels get that pointer as they get the hipMalloc'ed
// buffers of the other entry points.
static __device__ __align__(256) unsigned char scratch[RT_NOALLOC_SCRATCH_BYTES];
namespace {
constexpr int kMaxDevices = 64;              // (devices with a higher id get no scratch array: rt_render_noalloc refuses there)
unsigned char* g_scratch_ptr[kMaxDevices];   // per device: address of ITS copy of `scratch` (rt_create, under g_scratch_mutex)
std::mutex g_scratch_mutex;
}

// REFR path: every pixel of the window from its sum over samples and tree nodes in `accum`
// (render_pixel :297, render :318, :325).
__global__ __launch_bounds__(256) void rt_resolve_kernel(const RenderParams p) {
  const long long win_px = (long long)p.nx * p.n_rows;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int frame = blockIdx.y;
  if (i >= win_px) return;
  const float* a = p.accum + ((long long)frame * p.accum_stride_px + i) * 3;
  const float r = a[0] * p.inv_samples, g = a[1] * p.inv_samples, b = a[2] * p.inv_samples;
  const long long gpix = (long long)frame * p.frame_stride_px + i;
  if (p.out_f32) {
    float* o = p.out_f32 + gpix * 3;
    o[0] = fminf(1.0f, fmaxf(0.0f, r)); o[1] = fminf(1.0f, fmaxf(0.0f, g)); o[2] = fminf(1.0f, fmaxf(0.0f, b));
  }
  if (p.out_rgb) {
    const uint32_t rgbx = pack_rgbx(r, g, b);
    uint8_t* o = p.out_rgb + gpix * 3;
    o[0] = uint8_t(rgbx); o[1] = uint8_t(rgbx >> 8); o[2] = uint8_t(rgbx >> 16);
  }
}

// ------------------------------------------------------------------------------------------
// host side: context + C ABI
// ------------------------------------------------------------------------------------------
struct rt_ctx {
  int device = -1;
  unsigned char* scratch = nullptr;   // this device's copy of the `scratch` array (rt_render_noalloc), from rt_create
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  float4* d_geom = nullptr;
  float4* d_mat = nullptr;
  float* d_spec = nullptr;
  float2* d_refr = nullptr;      // refraction, refractive_index per object (rt_set_refraction, rt_set_cubes)
  bool has_refr = false;         // some sphere has refraction > 0: the REFR kernels run
  bool has_refr_cubes = false;   // some cube has
  // no colour, intensity or coefficient of the scene is negative and reflection / refraction lie in [0, 1], so no node of a
  // pixel's tree contributes a negative share: the refractive kernels may sum in saturating fixed point (spheres + lights /
  // their refraction values / the cubes)
  bool nonneg_scene = true, nonneg_refr = true, nonneg_cubes = true;
  bool sums_never_negative() const { return nonneg_scene && nonneg_refr && nonneg_cubes; }
  uint2* d_refl_tab = nullptr;        // direction table of reflection rays (build_reflect_table; grown on demand)
  size_t refl_tab_bytes = 0;
  int refl_dim = 0, refl_pairs = 0, refl_cpairs = 0;
  int refl_boxes = 0;                 // boxes the table was built with (it serves a scene with cubes only if they are these)
  std::vector<float> h_geom;          // the uploaded spheres (cx cy cz r^2), kept for rt_set_cubes' rebuild of the table
  uint32_t* d_occl_start = nullptr;   // occluder table rows as lists (scenes with >= kGridFrom spheres): [RT_MAX_SPHERES + 1]
  uint2* d_occl_items = nullptr;      // (grown on demand)
  size_t occl_items_bytes = 0;
  uint32_t* d_list_start = nullptr;   // the table as lists (build_reflect_lists; grown on demand)
  uint2* d_list_items = nullptr;
  size_t list_start_bytes = 0, list_items_bytes = 0;
  int list_dim = 0;
  bool reflect_table = true;          // RT_OPT_REFLECT_TABLE
  int queue_spill = 1;                // RT_OPT_QUEUE_SPILL: 0 off, 1 automatic depth, 2.. = at most that many blocks per wave
  uint32_t* d_glass_bits = nullptr;   // bit c: sphere c has refraction > 0 (rt_set_refraction; zero after rt_upload_scene)
  uint32_t* d_occl = nullptr;    // occluder table: [objects][buckets of the normal][ceil(n_spheres / 32)] (upload_occluder_table; grown on demand)
  size_t occl_bytes = 0;
  int occl_dim = 0;              // buckets per cube-map face edge of the bit-mask rows (0: one row per object)
  int occl_list_dim = 0;         // ... of the list rows (scenes with >= kGridFrom spheres)
  size_t occl_start_bytes = 0;
  float4* d_cubes = nullptr;     // [n_cubes][2] min, max (rt_set_cubes); objects n_spheres.. of d_mat / d_spec / d_refr
  int n_cubes = 0;
  unsigned long long* d_counters = nullptr;
  FrameRec* d_frames = nullptr;  // [kMaxBatch] cameras and seeds of the batch in flight
  // batches of more than kFrameChunk frames: their records go up in ONE copy from a ring of pinned staging slots (an event
  // per slot says when the copy that read it is done) instead of one tiny launch per 64 frames — sixteen launches, each with
  // its gap, ahead of a 1024-frame batch
  static constexpr int kFrameRing = 4;
  FrameRec* h_frames = nullptr;  // [kFrameRing][kMaxBatch], pinned
  hipEvent_t h_frames_ev[kFrameRing] = {};
  int h_frames_next = 0;
  CamGroup* d_camtab = nullptr;  // [kMaxBatch][cam_groups_stride(n)] primary-ray tables
  int16_t* d_vis = nullptr;      // visibility pass output, [batch][window pixels] (grown on demand)
  unsigned long long* d_present = nullptr;   // samples in one stream launch: per tile, the samples that hit (grown on demand)
  size_t present_words = 0;
  int* d_chunk = nullptr;        // stream kernel's tile counters (kTileCounters, one per 128-byte line), followed
  size_t chunk_bytes = 0;        // by the occupancy words of the batch's tiles (grown on demand)
  size_t vis_elems = 0;
  int n_cus = 256;
  float* d_spill = nullptr;      // stream kernel, refractive scenes: queue records set aside (grown on demand)
  size_t spill_bytes = 0;
  float* d_accum = nullptr;      // running sums for samples > 1 (grown on demand)
  size_t accum_floats = 0;
  int n_spheres = -1;
  int kernel = RT_KERNEL_STREAM;
  bool occluder_table = true;    // RT_OPT_OCCLUDER_TABLE: false = every sphere is a candidate for every object
  bool tile_flush = true;        // RT_OPT_TILE_FLUSH: false = the stream kernel stores finished pixels one by one
  unsigned long long host_primaries = 0;  // primary rays of the visibility pass (one per pixel and sample), counted here
  // the last render call (rt_compact_tiles): its frames, their tiling, and whether the occupancy bits of ONE launch
  // describe the whole batch
  struct LastBatch {
    const uint8_t* out_rgb = nullptr;
    TileGeom g{};
    int occ_words = 0;
    bool occ_valid = false;
  } last;
  DevLights lights{};
  void* comm = nullptr;          // RCCL communicator of this context (rt_comm_init; rt_comm.hpp), one rank per process
  int comm_ranks = 0, comm_rank = -1;
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

// Every sample of the area light, A + u (r_u - 0.5) + v (r_v - 0.5) with r in [0, 1) (raytracer.py:180-183), lies in the
// parallelogram spanned by u and v about A: its farthest points are corners, at half a diagonal — max(|u + v|, |u - v|) / 2
// (0.71 for the reference's unit square; 0.5 (|u| + |v|) = 1.0, the bound used until round 3, made every shadow cone 41 % wider).
float area_light_radius(const float* u, const float* v) {
  double p = 0.0, m = 0.0;
  for (int k = 0; k < 3; ++k) {
    p += (double(u[k]) + v[k]) * (double(u[k]) + v[k]);
    m += (double(u[k]) - v[k]) * (double(u[k]) - v[k]);
  }
  return float(0.5 * std::sqrt(std::max(p, m)) * (1.0 + 1e-6));
}
size_t geom_lds(int n_spheres) { return size_t((n_spheres * 16 + 15) / 16) * 16; }
size_t stream_lds(int n_spheres, int refr = 0, int out = 0, int waves = kBlock / 64) {
  return geom_lds(n_spheres) + size_t(wave_lds_bytes(refr, out)) * size_t(waves);
}
// resident workgroups of the stream kernel per CU: one per wave slot of a SIMD, or as many as 160 KB of LDS hold
unsigned workgroups_per_cu(size_t lds, unsigned waves_per_simd = RT_WAVES_PER_SIMD) {
  return unsigned(std::max<size_t>(1, std::min<size_t>(waves_per_simd, size_t(160 * 1024) / lds)));
}
size_t pixel_lds(int n_spheres) { return geom_lds(n_spheres) + size_t(kTileW) * kTileH * 4 + 16; }

// RN(1/n) for div_exact(): the float nearest to 1/n, chosen by the exact residual |1 - n*r| (n < 2^21
// and a 24-bit r multiply exactly in double), so double rounding cannot bite.
float rcp_rn(int n) {
  float best = float(1.0 / double(n));
  const float cand[2] = {std::nextafterf(best, 0.0f), std::nextafterf(best, 2.0f)};
  for (float c : cand)
    if (std::fabs(1.0 - double(n) * double(c)) < std::fabs(1.0 - double(n) * double(best))) best = c;
  return best;
}

struct Window {
  int x0, nx, n_rows, tile_rows, tile_first, tile_stride, y0;
};

#include "rt_tables.hpp"

// kernel variants: RNG mode x refraction in the scene x cubes in the scene
using RenderKernel = void (*)(const RenderParams);
RenderKernel pick_visibility_kernel(int rng_mode, bool cubes, bool multi = false, bool many = false) {
  if (many && !cubes && RT_VIS_SKIP && kVisRows > 1) {      // more than 64 spheres: the region culls them all, chunk by chunk
    if (multi) return rng_mode == RT_RNG_KEYED ? rt_visibility_kernel<RT_RNG_KEYED, false, true, true> : rt_visibility_kernel<RT_RNG_CONSTANT, false, true, true>;
    return rng_mode == RT_RNG_KEYED ? rt_visibility_kernel<RT_RNG_KEYED, false, false, true> : rt_visibility_kernel<RT_RNG_CONSTANT, false, false, true>;
  }
  if (multi) {
    if (rng_mode == RT_RNG_KEYED) return cubes ? rt_visibility_kernel<RT_RNG_KEYED, true, true> : rt_visibility_kernel<RT_RNG_KEYED, false, true>;
    return cubes ? rt_visibility_kernel<RT_RNG_CONSTANT, true, true> : rt_visibility_kernel<RT_RNG_CONSTANT, false, true>;
  }
  if (rng_mode == RT_RNG_KEYED) return cubes ? rt_visibility_kernel<RT_RNG_KEYED, true> : rt_visibility_kernel<RT_RNG_KEYED, false>;
  return cubes ? rt_visibility_kernel<RT_RNG_CONSTANT, true> : rt_visibility_kernel<RT_RNG_CONSTANT, false>;
}
// The stream kernel's variants, ONE table: pick_stream_kernel() looks a variant up in it and rt_create() walks it to
// allow every variant its dynamic LDS — a variant that is not listed here cannot be launched, and one that is listed
// cannot be forgotten by either.
//   refr / cubes: the scene has refractive materials / boxes       grid: 0 = bit tables, 1 = lists (256 spheres and
//   more), 2 / 3 = lists in workgroups of eight / sixteen waves    out: 0 = pixels one by one (float output, running
//   sums), 1 = tile slots, 2 = all samples of a pixel in one launch, 3 / 4 = the same with three slots (four-wave /
//   eight-wave workgroups)
struct StreamVariant {
  int refr;      // 0 = opaque scene, 1 = refractive (max_depth <= kRefrShallowDepth), 2 = refractive, any depth
  bool cubes;
  int grid, out;
  RenderKernel keyed, constant;
};
#define RT_STREAM_VARIANT(R, C, G, O) \
  {R, C, G, O, rt_stream_kernel<RT_RNG_KEYED, R, C, G, O>, rt_stream_kernel<RT_RNG_CONSTANT, R, C, G, O>}
const StreamVariant kStreamVariants[] = {
    // opaque scenes
    RT_STREAM_VARIANT(0, false, 0, 0), RT_STREAM_VARIANT(0, true, 0, 0),
    RT_STREAM_VARIANT(0, false, 1, 0), RT_STREAM_VARIANT(0, false, 2, 0), RT_STREAM_VARIANT(0, false, 3, 0),
    RT_STREAM_VARIANT(0, false, 0, 1), RT_STREAM_VARIANT(0, true, 0, 1),
    RT_STREAM_VARIANT(0, false, 1, 1), RT_STREAM_VARIANT(0, false, 2, 1), RT_STREAM_VARIANT(0, false, 3, 1),
    RT_STREAM_VARIANT(0, false, 0, 2), RT_STREAM_VARIANT(0, true, 0, 2), RT_STREAM_VARIANT(0, false, 1, 2),
    RT_STREAM_VARIANT(0, false, 0, 3), RT_STREAM_VARIANT(0, true, 0, 3),
    RT_STREAM_VARIANT(0, false, 0, 4), RT_STREAM_VARIANT(0, true, 0, 4),
    // refractive scenes: float sums in device memory (0), fixed-point sums in tile slots (1: one sample per pixel, 2: 2 .. 64
    // samples in one launch); queues of 160 / 192 records
    RT_STREAM_VARIANT(1, false, 0, 0), RT_STREAM_VARIANT(1, true, 0, 0), RT_STREAM_VARIANT(2, false, 0, 0), RT_STREAM_VARIANT(2, true, 0, 0),
    RT_STREAM_VARIANT(1, false, 0, 1), RT_STREAM_VARIANT(1, true, 0, 1), RT_STREAM_VARIANT(2, false, 0, 1), RT_STREAM_VARIANT(2, true, 0, 1),
    RT_STREAM_VARIANT(1, false, 0, 2), RT_STREAM_VARIANT(1, true, 0, 2), RT_STREAM_VARIANT(2, false, 0, 2), RT_STREAM_VARIANT(2, true, 0, 2),
};
#undef RT_STREAM_VARIANT
RenderKernel pick_stream_kernel(int rng_mode, int refr, bool cubes, int grid = 0, int out = 0) {
  for (const StreamVariant& v : kStreamVariants)
    if (v.refr == refr && v.cubes == cubes && v.grid == grid && v.out == out)
      return rng_mode == RT_RNG_KEYED ? v.keyed : v.constant;
  return nullptr;     // (launch() refuses: a combination the host logic must not ask for)
}

// Validates, then launches ceil(n_frames / RT_FRAMES_PER_LAUNCH) x samples kernel pairs on ctx->stream.
// `ws` (rt_render_noalloc): the device's `scratch` array; the call then allocates nothing — its frame record, table,
// counters and tile records lie there — and fails where it would have had to.
int launch(rt_ctx* ctx, const rt_camera* cams, const uint64_t* seeds, int n_frames, int width,
           int height, const Window& w, int samples, int max_depth, int rng_mode, uint8_t* out_rgb,
           float* out_f32, int32_t* out_hit, long long out_frame_stride_px = 0, unsigned char* ws = nullptr) {
  if (!ctx) return fail(nullptr, RT_EINVAL, "ctx is NULL");
  if (ctx->n_spheres < 0) return fail(ctx, RT_ESTATE, "rt_upload_scene has not been called");
  if (!cams || !seeds) return fail(ctx, RT_EINVAL, "camera/seed is NULL");
  if (width <= 0 || height <= 0 || width > (1 << 20) || height > (1 << 20))
    return fail(ctx, RT_EINVAL, "width/height out of range: %dx%d", width, height);
  if (samples < 1 || samples > 65536) return fail(ctx, RT_EINVAL, "samples out of range: %d", samples);
  if (max_depth < 0 || max_depth > RT_MAX_DEPTH)
    return fail(ctx, RT_EINVAL, "max_depth out of range [0,%d]: %d", RT_MAX_DEPTH, max_depth);
  if (rng_mode != RT_RNG_CONSTANT && rng_mode != RT_RNG_KEYED)
    return fail(ctx, RT_EINVAL, "unknown rng_mode %d", rng_mode);
  if (!out_rgb && !out_f32 && !out_hit) return fail(ctx, RT_EINVAL, "no output buffer given");
  for (int f = 0; f < n_frames; ++f)
    for (int i = 0; i < 12; ++i)
      if (!std::isfinite((&cams[f].origin[0])[i]))
        return fail(ctx, RT_EINVAL, "camera %d has a non-finite value", f);
  if (w.nx < 0 || w.n_rows < 0) return fail(ctx, RT_EINVAL, "negative window");
  if (w.n_rows == 0 || w.nx == 0 || n_frames == 0) return RT_OK;  // empty: nothing to do
  const long long win_px = (long long)w.nx * w.n_rows;
  if (win_px > (1ll << 31)) return fail(ctx, RT_EINVAL, "window of %lld pixels is too large", win_px);
  if (out_frame_stride_px != 0 && out_frame_stride_px < win_px)
    return fail(ctx, RT_EINVAL, "out_frame_stride_px %lld is smaller than a frame (%lld pixels)",
                out_frame_stride_px, win_px);
  const long long frame_px = out_frame_stride_px ? out_frame_stride_px : win_px;

  RT_HIP(ctx, hipSetDevice(ctx->device));
  // frames per launch pair: at most RT_FRAMES_PER_LAUNCH, and few enough that tile ids stay below 2^30
  // (the kernels count tiles of a batch in 32-bit integers)
  const long long frame_tiles = (long long)(((w.nx + 7) / 8) | 1) * ((w.n_rows + 7) / 8);   // tile ids per frame (odd pitch)
  const bool refr = ctx->has_refr || ctx->has_refr_cubes;
  const int refr_kind = refr ? (max_depth > kRefrShallowDepth ? 2 : 1) : 0;   // which refractive kernels: queue of 160 / 192 records
  // Refractive scenes, 8-bit frames at up to 64 samples per pixel: the stream kernel sums a pixel's trees in its tile's slot
  // (fixed point: needs shares that are never negative, which rt_upload_scene / rt_set_refraction / rt_set_cubes have
  // checked).  Otherwise (float output, several samples, RT_OPT_TILE_FLUSH off, a scene with negative colours) every node
  // adds into a float slab per frame (12 B per pixel) with float atomics, and a launch holds as many frames as fit
  // kRefrSumBytes of slabs (1024 at 1080p, 321 at 4K).
  const bool refr_slots = refr && ctx->kernel == RT_KERNEL_STREAM && ctx->tile_flush && ctx->sums_never_negative() && out_rgb &&
                          !out_f32 && samples <= 64;
  const int frames_cap = (refr && !refr_slots) ? int(std::max<long long>(1, std::min<long long>(kMaxBatch, kRefrSumBytes / (12 * std::max<long long>(win_px, 1)))))
                                               : kMaxBatch;
  const bool stream_kernel_early = ctx->kernel == RT_KERNEL_STREAM;
  const bool multi_early = stream_kernel_early && ctx->tile_flush && (!refr || refr_slots) && out_rgb && !out_f32 && samples >= 2 && samples <= 64;
  // all samples of a pixel in one launch: the tile records have a sample dimension — 144 B per tile, frame and sample, 4.7 MB
  // per 1080p frame and sample —, so a launch holds as many frames as fit kVisRecordBytes of them (1024 at 2 .. 5 samples,
  // 81 at 64)
  const long long multi_cap = multi_early ? std::max<long long>(1, kVisRecordBytes / (frame_tiles * kVisTile * 2 * samples)) : kMaxBatch;
  const int max_batch = int(std::max<long long>(1, std::min<long long>(std::min<long long>(frames_cap, multi_cap), (1ll << 30) / std::max<long long>(frame_tiles, 1))));
  const int batch = std::min(n_frames, max_batch);
  const bool cubes = ctx->n_cubes > 0;
  if ((refr || cubes) && ctx->kernel != RT_KERNEL_STREAM)
    return fail(ctx, RT_ESTATE, "refractive materials and cubes need the default kernel (RT_KERNEL_STREAM)");
  if (ws && (samples != 1 || refr || n_frames != 1 || ctx->kernel != RT_KERNEL_STREAM))
    return fail(ctx, RT_EINVAL, "rt_render_noalloc renders one frame at one sample per pixel of a scene without refraction "
                                "with the default kernel (samples %d, refractive %d): the other cases need buffers "
                                "that are allocated on demand", samples, int(refr));
  if ((samples > 1 && !multi_early) || (refr && !refr_slots)) {  // running sums over samples (and tree nodes), one slab per frame of a batch
    const size_t need = size_t(win_px) * 3 * size_t(batch);
    if (need > ctx->accum_floats) {
      RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
      if (ctx->d_accum) RT_HIP(ctx, hipFree(ctx->d_accum));
      ctx->d_accum = nullptr; ctx->accum_floats = 0;
      RT_HIP(ctx, hipMalloc(&ctx->d_accum, need * sizeof(float)));
      ctx->accum_floats = need;
    }
  }

  // refractive scenes: per wave of the stream kernel a stack of 64-record blocks for queue records set aside
  // (a level keeps at most ~128 records waiting while the levels below it are worked through)
  const int spill_blocks = (refr && ctx->queue_spill) ? std::min({kSpillBlocksMax, 2 * (max_depth + 1), ctx->queue_spill == 1 ? kSpillBlocksMax : ctx->queue_spill}) : 0;
  if (spill_blocks > 0) {
    const size_t waves = size_t(ctx->n_cus) * RT_WAVES_PER_SIMD * (kBlock / 64);   // the most the kernel is launched with
    const size_t need = waves * size_t(spill_blocks) * kSpillFields * 64 * sizeof(float);
    if (need > ctx->spill_bytes) {
      RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
      if (ctx->d_spill) RT_HIP(ctx, hipFree(ctx->d_spill));
      ctx->d_spill = nullptr; ctx->spill_bytes = 0;
      RT_HIP(ctx, hipMalloc(&ctx->d_spill, need));
      ctx->spill_bytes = need;
    }
  }

  const bool stream_kernel = ctx->kernel == RT_KERNEL_STREAM;
  // 8-bit frames at one sample per pixel: the stream kernel stages finished pixels per tile and stores whole tiles
  // ... and with 2..64 samples per pixel all of them are traced in ONE launch pair and summed in the slots
  int slots = ((refr ? refr_slots : stream_kernel && ctx->tile_flush) && out_rgb && !out_f32) ? (samples == 1 ? 1 : samples <= 64 ? 2 : 0) : 0;
  // ... unless the slots' LDS would cost a resident workgroup per CU (many-sphere scenes, whose sphere list takes
  // 16 B per sphere of every workgroup's LDS): config 4 measured 88 Grays/s on three workgroups per CU with
  // scattered stores against 65 on two with tile stores
  if (slots == 1 && !refr && workgroups_per_cu(stream_lds(ctx->n_spheres + 2 * ctx->n_cubes, refr, 1)) <
                        workgroups_per_cu(stream_lds(ctx->n_spheres + 2 * ctx->n_cubes, refr, 0)))
    slots = 0;
  // all samples in one launch: the variant with three slots per wave where that buys a fourth workgroup per CU (scenes of
  // up to 36 spheres: config 2 at 2 / 4 samples 513 -> 545 / 531 -> 568 Grays/s; on larger scenes, where three workgroups
  // fit either way, it measured 9 % slower)
  const bool listed = ctx->n_spheres >= kGridFrom && !refr && !cubes && stream_kernel;   // the list kernels (lighting<LIST>, nearest_list)
  // Many spheres (the list / grid kernels): every workgroup holds a copy of the sphere list, 16 KB at 1024 spheres, so only
  // three workgroups of four waves fit a CU.  Workgroups of EIGHT waves share one copy: two of them (80 KB each at 1024
  // spheres) are 16 waves per CU.  Taken where that is more waves than the four-wave form holds, if need be without
  // tile slots (config 4: 308 -> 334 Grays/s).
  // (and from ~1100 spheres on ONE workgroup of sixteen waves: 2000 spheres 125 -> see DESIGN)
  int wg_waves = 4;
  if (stream_kernel && !refr && !cubes && listed && slots < 2) {
    unsigned best = 4u * workgroups_per_cu(stream_lds(ctx->n_spheres, false, slots));
    const int slots4 = slots;
    for (int wv : {8, 16})
      for (int sl : {slots4, 0}) {
        const unsigned waves = unsigned(wv) * workgroups_per_cu(stream_lds(ctx->n_spheres, false, sl, wv), unsigned(16 / wv));
        if (stream_lds(ctx->n_spheres, false, sl, wv) <= size_t(160 * 1024) && waves > best) { best = waves; wg_waves = wv; slots = sl; }
      }
  }
  if (slots == 2 && !refr && !listed && workgroups_per_cu(stream_lds(ctx->n_spheres + 2 * ctx->n_cubes, refr, 3)) >
                                   workgroups_per_cu(stream_lds(ctx->n_spheres + 2 * ctx->n_cubes, refr, 2)))
    slots = 3;
  // ... or, up to 72 spheres, as two workgroups of eight waves around one copy of the spheres (the 64-sphere target scene)
  if (slots == 2 && !refr && !listed && 2 * stream_lds(ctx->n_spheres + 2 * ctx->n_cubes, refr, 3, 8) <= size_t(160 * 1024)) {
    slots = 4;
    wg_waves = 8;
  }
  const bool multi = slots >= 2;
  const size_t vis_need = size_t(((w.nx + 7) / 8) | 1) * size_t((w.n_rows + 7) / 8) * kVisTile * size_t(batch) *
                          size_t(multi ? samples : 1);   // whole 8x8 tiles, odd pitch; MULTI: a record per sample
  if (stream_kernel && !ws) {  // primary-hit indices handed from the visibility pass to the stream kernel
    const size_t need = vis_need;
    if (need > ctx->vis_elems) {
      RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
      if (ctx->d_vis) RT_HIP(ctx, hipFree(ctx->d_vis));
      ctx->d_vis = nullptr; ctx->vis_elems = 0;
      RT_HIP(ctx, hipMalloc(&ctx->d_vis, need * sizeof(int16_t)));
      ctx->vis_elems = need;
    }
  }

  ctx->last = rt_ctx::LastBatch();
  RenderParams p{};
  p.L = ctx->lights;
  p.width = width; p.height = height;
  p.x0 = w.x0; p.nx = w.nx; p.n_rows = w.n_rows;
  p.tile_rows = w.tile_rows; p.tile_first = w.tile_first; p.tile_stride = w.tile_stride; p.y0 = w.y0;
  p.n_samples = samples; p.max_depth = max_depth;
  p.inv_w = rcp_rn(width); p.inv_h = rcp_rn(height);
  p.inv_aspect = float(height) / float(width);
  p.inv_samples = 1.0f / float(samples);
  p.n_spheres = ctx->n_spheres;
  p.geom = ctx->d_geom; p.mat = ctx->d_mat; p.spec = ctx->d_spec; p.refr = ctx->d_refr;
  p.cubes = ctx->d_cubes; p.n_cubes = ctx->n_cubes;
  p.occl = ctx->d_occl; p.occl_words = (ctx->n_spheres + 31) / 32; p.occl_dim = ctx->occl_dim;
  p.occl_start = ctx->d_occl_start; p.occl_items = ctx->d_occl_items; p.occl_list_dim = ctx->occl_list_dim;
  p.glass_bits = ctx->d_glass_bits;
  p.list_start = ctx->d_list_start; p.list_items = ctx->d_list_items;
  p.list_dim = (refr || cubes || !stream_kernel) ? 0 : ctx->list_dim;
  p.refl_tab = ctx->d_refl_tab; p.refl_pairs = ctx->refl_pairs; p.refl_cpairs = ctx->refl_cpairs;
  p.refl_dim = (!stream_kernel || listed || ctx->refl_boxes != ctx->n_cubes) ? 0 : ctx->refl_dim;
  p.defer_output = (refr && !refr_slots) ? 1 : 0;
  p.tile_flush = slots ? 1 : 0;
  p.accum = ctx->d_accum;
  p.spill = ctx->d_spill; p.spill_blocks = spill_blocks;
  FrameRec* d_frames = ctx->d_frames;
  CamGroup* d_camtab = ctx->d_camtab;
  int16_t* d_vis = ctx->d_vis;
  int* d_chunk = ctx->d_chunk;
  // tile occupancy: per counter one bit per slot, rounded up to words, plus one word of padding (a claim's
  // bits are read as two words)
  const long long batch_tiles = frame_tiles * batch;
  const int occ_words = int(((batch_tiles + kTileCounters - 1) / kTileCounters + 31) / 32) + 1;
  const size_t chunk_need = size_t(kTileCounters) * 128 + size_t(kTileCounters) * size_t(occ_words) * 4;
  if (multi) {   // (never with `ws`: rt_render_noalloc takes one sample per pixel)
    const size_t need = size_t(batch_tiles) + 8;   // (+8: a wave reads the words of its four tiles, also past the last id)
    if (need > ctx->present_words) {
      RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
      if (ctx->d_present) RT_HIP(ctx, hipFree(ctx->d_present));
      ctx->d_present = nullptr; ctx->present_words = 0;
      RT_HIP(ctx, hipMalloc(&ctx->d_present, need * sizeof(unsigned long long)));
      ctx->present_words = need;
    }
  }
  p.present = ctx->d_present;
  if (stream_kernel && !ws) {
    const size_t need = chunk_need;
    if (need > ctx->chunk_bytes) {
      RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
      if (ctx->d_chunk) RT_HIP(ctx, hipFree(ctx->d_chunk));
      ctx->d_chunk = nullptr; ctx->chunk_bytes = 0;
      RT_HIP(ctx, hipMalloc(&ctx->d_chunk, need));
      ctx->chunk_bytes = need;
    }
    d_chunk = ctx->d_chunk;
    d_vis = ctx->d_vis;
  }
  if (ws) {   // rt_render_noalloc: the same buffers, carved out of the device's scratch array (256-byte aligned)
    auto up = [](size_t n) { return (n + 255) & ~size_t(255); };
    const size_t o_frames = 0, o_camtab = up(sizeof(FrameRec));
    const size_t o_chunk = o_camtab + up(sizeof(CamGroup) * size_t(cam_groups_stride(ctx->n_spheres)));
    const size_t o_vis = o_chunk + up(chunk_need);
    const size_t total = o_vis + vis_need * sizeof(int16_t);
    if (total > size_t(RT_NOALLOC_SCRATCH_BYTES))
      return fail(ctx, RT_EINVAL, "rt_render_noalloc: a %dx%d frame needs %zu bytes of scratch, the library holds %zu "
                                  "(RT_NOALLOC_SCRATCH_BYTES); use rt_render", w.nx, w.n_rows, total, size_t(RT_NOALLOC_SCRATCH_BYTES));
    d_frames = reinterpret_cast<FrameRec*>(ws + o_frames);
    d_camtab = reinterpret_cast<CamGroup*>(ws + o_camtab);
    d_chunk = reinterpret_cast<int*>(ws + o_chunk);
    d_vis = reinterpret_cast<int16_t*>(ws + o_vis);
  }
  p.camtab = d_camtab;
  p.vis = d_vis;
  p.chunk_counter = d_chunk;
  p.occ = reinterpret_cast<uint32_t*>(d_chunk) + kTileCounters * 32;
  p.occ_words = occ_words;
  p.tiles8_x = (w.nx + 7) / 8; p.tiles8_y = (w.n_rows + 7) / 8;
  p.counters = ctx->d_counters;
  p.frame_stride_px = frame_px;
  p.accum_stride_px = win_px;
  p.tile_pitch = p.tiles8_x | 1;
  p.vis_stride = (long long)p.tile_pitch * p.tiles8_y * kVisTile;
  p.vis_sample_stride = p.vis_stride * batch;
  unsigned per_frame;
  if (stream_kernel) {
    p.strips_x = p.tiles8_x; p.strips_y = p.tiles8_y;   // work units of the stream kernel: 8x8 tiles
    per_frame = unsigned(p.strips_x) * unsigned(p.strips_y);
  } else {
    per_frame = unsigned((w.nx + kTileW - 1) / kTileW) * unsigned((w.n_rows + kTileH - 1) / kTileH);
  }

  if (out_rgb && !ws) {   // what rt_compact_tiles needs to know about this batch
    ctx->last.out_rgb = out_rgb;
    ctx->last.g = TileGeom{w.nx, w.n_rows, n_frames, p.tiles8_x, p.tiles8_y, p.tile_pitch, frame_px};
    ctx->last.occ_words = occ_words;
    // (one launch pair for the whole batch, or MULTI's one stream launch: the bits cover every sample's hits)
    ctx->last.occ_valid = stream_kernel && n_frames <= max_batch && (samples == 1 || multi);
  }
  for (int f0 = 0; f0 < n_frames; f0 += max_batch) {
    const int nf = std::min(max_batch, n_frames - f0);
    const size_t lds = stream_kernel ? stream_lds(ctx->n_spheres + 2 * ctx->n_cubes, refr_kind, slots, wg_waves) : pixel_lds(ctx->n_spheres);
    p.n_frames = nf;
    p.frames = d_frames;
    if (!ws && nf > kFrameChunk) {                   // cameras and seeds of a big batch -> device: one copy from pinned staging
      const int slot = ctx->h_frames_next;
      ctx->h_frames_next = (slot + 1) % rt_ctx::kFrameRing;
      RT_HIP(ctx, hipEventSynchronize(ctx->h_frames_ev[slot]));   // (the copy that last read this slot: four batches ago)
      FrameRec* h = ctx->h_frames + size_t(slot) * kMaxBatch;
      for (int f = 0; f < nf; ++f) {
        h[f].cam = cams[f0 + f];
        h[f].seed = uint32_t(seeds[f0 + f] ^ (seeds[f0 + f] >> 32));
        h[f].pad[0] = h[f].pad[1] = h[f].pad[2] = 0u;
      }
      RT_HIP(ctx, hipMemcpyAsync(d_frames, h, sizeof(FrameRec) * size_t(nf), hipMemcpyHostToDevice, ctx->stream));
      RT_HIP(ctx, hipEventRecord(ctx->h_frames_ev[slot], ctx->stream));
    } else
    for (int c0 = 0; c0 < nf; c0 += kFrameChunk) {   // cameras and seeds of the batch -> device (stream order
      FrameChunk chunk;                              // protects the previous batch's readers)
      const int nc = std::min(kFrameChunk, nf - c0);
      for (int f = 0; f < nc; ++f) {
        chunk.cam[f] = cams[f0 + c0 + f];
        chunk.seed[f] = uint32_t(seeds[f0 + c0 + f] ^ (seeds[f0 + c0 + f] >> 32));
      }
      for (int f = nc; f < kFrameChunk; ++f) { chunk.cam[f] = chunk.cam[0]; chunk.seed[f] = 0u; }
      hipLaunchKernelGGL(rt_store_frames_kernel, dim3(1), dim3(256), 0, ctx->stream, chunk, d_frames + c0, nc);
      RT_HIP(ctx, hipGetLastError());
    }
    p.out_rgb = out_rgb ? out_rgb + size_t(f0) * frame_px * 3 : nullptr;
    p.out_f32 = out_f32 ? out_f32 + size_t(f0) * frame_px * 3 : nullptr;
    p.out_hit = out_hit ? out_hit + size_t(f0) * frame_px : nullptr;
    const dim3 grid(per_frame * unsigned(nf), 1, 1);
    {  // primary-ray tables of this batch
      const int n4 = cam_groups_stride(ctx->n_spheres) * 4;
      {
        hipLaunchKernelGGL(rt_camera_table_kernel, dim3(unsigned((n4 + 255) / 256), unsigned(nf), 1), dim3(256),
                           0, ctx->stream, (const float4*)ctx->d_geom, ctx->n_spheres, (const FrameRec*)d_frames, d_camtab, nf);
        RT_HIP(ctx, hipGetLastError());
      }
    }
    if (p.defer_output)   // every node of every sample adds into the pixel sums
      RT_HIP(ctx, hipMemsetAsync(ctx->d_accum, 0, size_t(win_px) * 3 * size_t(nf) * sizeof(float), ctx->stream));
    if (multi)   // which samples of a tile hit anything: collected by the visibility launches below
      RT_HIP(ctx, hipMemsetAsync(ctx->d_present, 0, (size_t(frame_tiles) * size_t(nf) + 8) * sizeof(unsigned long long), ctx->stream));
    for (int s = 0; s < samples; ++s) {
      p.sample_index = s;
      if (stream_kernel) {
        // one wave per pair of 8x8 tiles, 8 tiles (64 x 8 pixels) per workgroup; tile rows go in x (no 65535 limit)
        const dim3 vgrid(unsigned((w.n_rows + 8 * kVisRows - 1) / (8 * kVisRows)), unsigned((w.nx + 8 * kVisTiles * (RT_VIS_BLOCK / 64) - 1) / (8 * kVisTiles * (RT_VIS_BLOCK / 64))), unsigned(nf));
        const size_t vlds = 0;
        ctx->host_primaries += (unsigned long long)win_px * (unsigned long long)nf;   // one primary ray per pixel and sample
        // persistent stream kernel: as many workgroups as stay resident (4 per CU), never more than chunks
        // (and never more than the LDS lets stay resident)
        // (many-sphere scenes: 1 = the list / grid kernel, 2 / 3 = its forms with eight / sixteen waves per workgroup)
        const int list_kernel = listed ? (wg_waves == 16 ? 3 : wg_waves == 8 ? 2 : 1) : 0;
        const unsigned per_cu = wg_waves > 4 ? workgroups_per_cu(lds, unsigned(16 / wg_waves)) : workgroups_per_cu(lds, (slots == 2 && !refr) ? 3u : unsigned(RT_WAVES_PER_SIMD));
        const unsigned threads = unsigned(wg_waves) * 64u;
        const dim3 sgrid(std::min((grid.x + threads / 64 - 1) / (threads / 64), unsigned(ctx->n_cus) * per_cu), 1, 1);
        static_assert(sizeof(rt_camera) == 48, "the stream kernel copies cameras as 12 floats");
        // (work counters and occupancy bits start every launch at zero — also those in the scratch array, which
        // keeps the last call's contents)
        if (!multi || s == 0) RT_HIP(ctx, hipMemsetAsync(d_chunk, 0, chunk_need, ctx->stream));
        hipLaunchKernelGGL(pick_visibility_kernel(rng_mode, cubes, multi, ctx->n_spheres > 64), vgrid, dim3(RT_VIS_BLOCK), vlds, ctx->stream, p);
        // (MULTI: one visibility launch per sample, then ONE stream-kernel launch that shades them all)
        if (!multi || s + 1 == samples) {
          const RenderKernel shade = pick_stream_kernel(rng_mode, refr_kind, cubes, list_kernel, slots);
          if (!shade) return fail(ctx, RT_ESTATE, "no stream kernel variant for refr %d cubes %d lists %d output %d", int(refr), int(cubes), list_kernel, slots);
          hipLaunchKernelGGL(shade, sgrid, dim3(threads), lds, ctx->stream, p);
        }
        if (p.defer_output && s + 1 == samples) {   // (never MULTI)
          hipLaunchKernelGGL(rt_resolve_kernel, dim3(unsigned((win_px + 255) / 256), unsigned(nf), 1), dim3(256), 0, ctx->stream, p);
        }
      } else {
        if (rng_mode == RT_RNG_KEYED)
          hipLaunchKernelGGL(rt_pixel_kernel<RT_RNG_KEYED>, grid, dim3(kBlock), lds, ctx->stream, p);
        else
          hipLaunchKernelGGL(rt_pixel_kernel<RT_RNG_CONSTANT>, grid, dim3(kBlock), lds, ctx->stream, p);
      }
      RT_HIP(ctx, hipGetLastError());
    }
  }
  return RT_OK;
}

// The occluder table of lighting()'s shadow scan for the context's spheres (ctx->h_geom, ctx->lights) and `n_boxes` boxes
// behind them: bit-mask rows per (object, bucket of the surface normal) — boxes: every sphere, every bucket — and, from
// kGridFrom spheres on, the same rows as lists for the list kernels.  Host build (FP64), one copy to the device.
int upload_occluder_rows(rt_ctx* ctx, int n_boxes);
int upload_occluder_table(rt_ctx* ctx, int n_boxes) {
  // (a failure leaves rows of one layout under the dimensions of another: the context then holds no scene at all, and
  // renders return RT_ESTATE until the next rt_upload_scene)
  const int rc = upload_occluder_rows(ctx, n_boxes);
  if (rc != RT_OK) ctx->n_spheres = -1;
  return rc;
}
int upload_occluder_rows(rt_ctx* ctx, int n_boxes) {
  const int n = int(ctx->h_geom.size() / 4), words = (n + 31) / 32;
  ctx->occl_dim = 0; ctx->occl_list_dim = 0;
  if (n == 0) return RT_OK;
  const float* geom4 = ctx->h_geom.data();
  std::vector<uint32_t> object_rows;
  build_occluder_table(geom4, n, ctx->lights, object_rows, ctx->occluder_table);
  const int B = (ctx->occluder_table && n < kGridFrom) ? occluder_table_dim(n + n_boxes, n) : 0;
  const int nb = B ? 6 * B * B : 1;
  std::vector<uint32_t> rows;
  if (B) build_occluder_buckets(geom4, n, ctx->lights, B, object_rows, rows);
  else rows = object_rows;
  rows.resize(size_t(n + n_boxes) * size_t(nb) * size_t(words), 0u);
  for (size_t r = size_t(n) * size_t(nb); r < size_t(n + n_boxes) * size_t(nb); ++r)
    for (int c = 0; c < n; ++c) rows[r * size_t(words) + size_t(c >> 5)] |= 1u << (c & 31);
  const size_t bytes = rows.size() * sizeof(uint32_t);
  if (bytes > ctx->occl_bytes) {
    if (ctx->d_occl) (void)hipFree(ctx->d_occl);
    ctx->d_occl = nullptr; ctx->occl_bytes = 0;
    RT_HIP(ctx, hipMalloc(&ctx->d_occl, bytes));
    ctx->occl_bytes = bytes;
  }
  RT_HIP(ctx, hipMemcpy(ctx->d_occl, rows.data(), bytes, hipMemcpyHostToDevice));
  ctx->occl_dim = B;
  if (n >= kGridFrom && n_boxes == 0) {   // the list kernels (sphere-only scenes) read rows as lists (lighting<..., LIST>)
    const int LB = ctx->occluder_table ? occluder_table_dim(n, n) : 0, lnb = LB ? 6 * LB * LB : 1;
    std::vector<uint32_t> brows;
    if (LB) build_occluder_buckets(geom4, n, ctx->lights, LB, object_rows, brows);
    const std::vector<uint32_t>& src = LB ? brows : object_rows;
    std::vector<uint32_t> start(size_t(n) * size_t(lnb) + 1, 0u);
    std::vector<uint2> items;
    for (size_t r = 0; r < size_t(n) * size_t(lnb); ++r) {
      start[r] = uint32_t(items.size());
      uint32_t v[4];
      int fill = 0;
      for (int w = 0; w < words; ++w) {
        uint32_t bits = src[r * size_t(words) + size_t(w)];
        while (bits) {
          v[fill++] = uint32_t(32 * w + __builtin_ctz(bits));
          bits &= bits - 1u;
          if (fill == 4) { items.push_back(make_uint2(v[0] | (v[1] << 16), v[2] | (v[3] << 16))); fill = 0; }
        }
      }
      if (fill) {
        for (int q = fill; q < 4; ++q) v[q] = 0xFFFFu;
        items.push_back(make_uint2(v[0] | (v[1] << 16), v[2] | (v[3] << 16)));
      }
    }
    start.back() = uint32_t(items.size());
    items.push_back(make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu));
    const size_t sb = start.size() * sizeof(uint32_t), ib = items.size() * sizeof(uint2);
    if (sb > ctx->occl_start_bytes) {
      if (ctx->d_occl_start) (void)hipFree(ctx->d_occl_start);
      ctx->d_occl_start = nullptr; ctx->occl_start_bytes = 0;
      RT_HIP(ctx, hipMalloc(&ctx->d_occl_start, sb));
      ctx->occl_start_bytes = sb;
    }
    if (ib > ctx->occl_items_bytes) {
      if (ctx->d_occl_items) (void)hipFree(ctx->d_occl_items);
      ctx->d_occl_items = nullptr; ctx->occl_items_bytes = 0;
      RT_HIP(ctx, hipMalloc(&ctx->d_occl_items, ib));
      ctx->occl_items_bytes = ib;
    }
    RT_HIP(ctx, hipMemcpy(ctx->d_occl_start, start.data(), sb, hipMemcpyHostToDevice));
    RT_HIP(ctx, hipMemcpy(ctx->d_occl_items, items.data(), ib, hipMemcpyHostToDevice));
    ctx->occl_list_dim = LB;
  }
  return RT_OK;
}

// The direction table of secondary rays (bit-mask form, scenes of 4 .. 255 spheres) for the context's spheres and `n_boxes`
// boxes: built on the host, copied to the device (the buffer grows on demand).
int upload_reflect_table(rt_ctx* ctx, const float4* boxes, int n_boxes) {
  const int n = int(ctx->h_geom.size() / 4);
  const int B = ctx->reflect_table ? reflect_table_dim(n, n_boxes) : 0;
  ctx->refl_dim = 0; ctx->refl_boxes = 0;
  if (B == 0) return RT_OK;
  std::vector<uint2> tab;
  build_reflect_table(ctx->h_geom.data(), n, B, tab, boxes, n_boxes);
  const size_t bytes = tab.size() * sizeof(uint2);
  if (bytes > ctx->refl_tab_bytes) {
    if (ctx->d_refl_tab) (void)hipFree(ctx->d_refl_tab);
    ctx->d_refl_tab = nullptr; ctx->refl_tab_bytes = 0;
    RT_HIP(ctx, hipMalloc(&ctx->d_refl_tab, bytes));
    ctx->refl_tab_bytes = bytes;
  }
  RT_HIP(ctx, hipMemcpy(ctx->d_refl_tab, tab.data(), bytes, hipMemcpyHostToDevice));
  ctx->refl_dim = B; ctx->refl_pairs = (n + 63) / 64; ctx->refl_cpairs = (n_boxes + 63) / 64; ctx->refl_boxes = n_boxes;
  return RT_OK;
}

template <typename K>
hipError_t allow_big_lds(K kernel, size_t bytes) {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                             hipFuncAttributeMaxDynamicSharedMemorySize, int(bytes));
}

}  // namespace

#include "rt_comm.hpp"

extern "C" {

const char* rt_version(void) { return "rt355 0.2 (gfx950)"; }

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
  if (device_id < kMaxDevices) {
    // this device's copy of the scratch array of rt_render_noalloc (every GPU has its own, at its own address); contexts
    // may be created from several threads
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device_id]) {
      unsigned char* ptr = nullptr;
      if ((e = hipGetSymbolAddress(reinterpret_cast<void**>(&ptr), HIP_SYMBOL(scratch))) != hipSuccess)
        return bail(e, "hipGetSymbolAddress(scratch)");
      g_scratch_ptr[device_id] = ptr;
    }
    ctx->scratch = g_scratch_ptr[device_id];
  }
  if ((e = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking)) != hipSuccess)
    return bail(e, "hipStreamCreate");
  ctx->stream = ctx->own_stream;
  if ((e = hipMalloc(&ctx->d_geom, sizeof(float4) * (RT_MAX_SPHERES + 8))) != hipSuccess) return bail(e, "hipMalloc");
  if ((e = hipMalloc(&ctx->d_mat, sizeof(float4) * RT_MAX_SPHERES)) != hipSuccess) return bail(e, "hipMalloc");
  if ((e = hipMalloc(&ctx->d_spec, sizeof(float) * RT_MAX_SPHERES)) != hipSuccess) return bail(e, "hipMalloc");
  if ((e = hipMalloc(&ctx->d_refr, sizeof(float2) * RT_MAX_SPHERES)) != hipSuccess) return bail(e, "hipMalloc");
  if ((e = hipMalloc(&ctx->d_cubes, sizeof(float4) * RT_MAX_SPHERES)) != hipSuccess) return bail(e, "hipMalloc");
  if ((e = hipMalloc(&ctx->d_glass_bits, sizeof(uint32_t) * (RT_MAX_SPHERES / 32))) != hipSuccess) return bail(e, "hipMalloc");
  if ((e = hipMemset(ctx->d_glass_bits, 0, sizeof(uint32_t) * (RT_MAX_SPHERES / 32))) != hipSuccess) return bail(e, "hipMemset");
  if ((e = hipMalloc(&ctx->d_frames, sizeof(FrameRec) * kMaxBatch)) != hipSuccess) return bail(e, "hipMalloc");
  if ((e = hipHostMalloc(reinterpret_cast<void**>(&ctx->h_frames), sizeof(FrameRec) * kMaxBatch * rt_ctx::kFrameRing, hipHostMallocDefault)) != hipSuccess)
    return bail(e, "hipHostMalloc");
  for (int k = 0; k < rt_ctx::kFrameRing; ++k)
    if ((e = hipEventCreateWithFlags(&ctx->h_frames_ev[k], hipEventDisableTiming)) != hipSuccess) return bail(e, "hipEventCreate");
  if ((e = hipMalloc(&ctx->d_camtab, sizeof(CamGroup) * size_t(cam_groups_stride(RT_MAX_SPHERES)) * kMaxBatch)) != hipSuccess)
    return bail(e, "hipMalloc");
  ctx->n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  if ((e = hipMalloc(&ctx->d_counters, sizeof(unsigned long long) * kCounterSlots * kCounterWords)) != hipSuccess)
    return bail(e, "hipMalloc");
  if ((e = hipMemset(ctx->d_counters, 0, sizeof(unsigned long long) * kCounterSlots * kCounterWords)) != hipSuccess)
    return bail(e, "hipMemset");
  // every kernel with dynamic LDS may use all of a CU's 160 KB (the launch says how much it takes)
  for (const StreamVariant& v : kStreamVariants)
    if ((e = allow_big_lds(v.keyed, size_t(160 * 1024))) != hipSuccess || (e = allow_big_lds(v.constant, size_t(160 * 1024))) != hipSuccess)
      return bail(e, "hipFuncSetAttribute");
  if ((e = allow_big_lds(rt_pixel_kernel<RT_RNG_KEYED>, stream_lds(RT_MAX_SPHERES))) != hipSuccess ||
      (e = allow_big_lds(rt_pixel_kernel<RT_RNG_CONSTANT>, stream_lds(RT_MAX_SPHERES))) != hipSuccess)
    return bail(e, "hipFuncSetAttribute");
  *out = ctx;
  return RT_OK;
}

void rt_destroy(rt_ctx* ctx) {
  if (!ctx) return;
  if (ctx->device >= 0) (void)hipSetDevice(ctx->device);
  if (ctx->comm) (void)rt_comm_destroy(ctx);
  if (ctx->own_stream) { (void)hipStreamSynchronize(ctx->own_stream); (void)hipStreamDestroy(ctx->own_stream); }
  if (ctx->d_geom) (void)hipFree(ctx->d_geom);
  if (ctx->d_mat) (void)hipFree(ctx->d_mat);
  if (ctx->d_spec) (void)hipFree(ctx->d_spec);
  if (ctx->d_refr) (void)hipFree(ctx->d_refr);
  if (ctx->d_cubes) (void)hipFree(ctx->d_cubes);
  if (ctx->d_occl) (void)hipFree(ctx->d_occl);
  if (ctx->d_glass_bits) (void)hipFree(ctx->d_glass_bits);
  if (ctx->d_refl_tab) (void)hipFree(ctx->d_refl_tab);
  if (ctx->d_occl_start) (void)hipFree(ctx->d_occl_start);
  if (ctx->d_occl_items) (void)hipFree(ctx->d_occl_items);
  if (ctx->d_list_start) (void)hipFree(ctx->d_list_start);
  if (ctx->d_list_items) (void)hipFree(ctx->d_list_items);
  if (ctx->d_counters) (void)hipFree(ctx->d_counters);
  if (ctx->d_camtab) (void)hipFree(ctx->d_camtab);
  if (ctx->d_frames) (void)hipFree(ctx->d_frames);
  for (int k = 0; k < rt_ctx::kFrameRing; ++k)      // (a copy out of the pinned ring may still be queued on a caller's stream)
    if (ctx->h_frames_ev[k]) { (void)hipEventSynchronize(ctx->h_frames_ev[k]); (void)hipEventDestroy(ctx->h_frames_ev[k]); }
  if (ctx->h_frames) (void)hipHostFree(ctx->h_frames);
  if (ctx->d_accum) (void)hipFree(ctx->d_accum);
  if (ctx->d_spill) (void)hipFree(ctx->d_spill);
  if (ctx->d_vis) (void)hipFree(ctx->d_vis);
  if (ctx->d_present) (void)hipFree(ctx->d_present);
  if (ctx->d_chunk) (void)hipFree(ctx->d_chunk);
  delete ctx;
}

int rt_set_stream(rt_ctx* ctx, void* hip_stream, int use_own_stream) {
  if (!ctx) return fail(nullptr, RT_EINVAL, "ctx is NULL");
  // hip_stream == NULL with use_own_stream == 0 is HIP's legacy default stream (what
  // torch.cuda.current_stream().cuda_stream is unless the caller switched streams).
  ctx->stream = use_own_stream ? ctx->own_stream : static_cast<hipStream_t>(hip_stream);
  return RT_OK;
}

int rt_set_option(rt_ctx* ctx, int option, int value) {
  if (!ctx) return fail(nullptr, RT_EINVAL, "ctx is NULL");
  if (option == RT_OPT_KERNEL && (value == RT_KERNEL_STREAM || value == RT_KERNEL_PIXEL)) {
    ctx->kernel = value;
    return RT_OK;
  }
  if (option == RT_OPT_QUEUE_SPILL && value >= 0 && value <= kSpillBlocksMax) {
    ctx->queue_spill = value;           // takes effect at the next render call
    return RT_OK;
  }
  if (option == RT_OPT_TILE_FLUSH && (value == 0 || value == 1)) {
    ctx->tile_flush = value != 0;       // takes effect at the next render call
    return RT_OK;
  }
  if (option == RT_OPT_REFLECT_TABLE && (value == 0 || value == 1)) {
    ctx->reflect_table = value != 0;    // takes effect at the next rt_upload_scene
    return RT_OK;
  }
  if (option == RT_OPT_OCCLUDER_TABLE && (value == 0 || value == 1)) {
    ctx->occluder_table = value != 0;   // takes effect at the next rt_upload_scene
    return RT_OK;
  }
  return fail(ctx, RT_EINVAL, "unknown option %d / value %d", option, value);
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
  // From here on the device holds neither the old scene nor the new one: a failure below leaves the context without a
  // scene (renders return RT_ESTATE) instead of the old sphere count and flags over the new geometry.
  ctx->n_spheres = -1;
  ctx->has_refr = false; ctx->n_cubes = 0; ctx->has_refr_cubes = false;
  ctx->nonneg_refr = true; ctx->nonneg_cubes = true;
  {   // eight never-hit spheres behind the list: nearest_hit reads it in blocks of four, one block ahead
    float pad[8 * 4];
    for (int i = 0; i < 8; ++i) { pad[4 * i] = pad[4 * i + 1] = pad[4 * i + 2] = 0.0f; pad[4 * i + 3] = -INFINITY; }
    RT_HIP(ctx, hipMemcpy(ctx->d_geom + n_spheres, pad, sizeof(pad), hipMemcpyHostToDevice));
  }
  if (n_spheres > 0) {
    RT_HIP(ctx, hipMemcpy(ctx->d_geom, geom4, sizeof(float) * 4 * n_spheres, hipMemcpyHostToDevice));
    RT_HIP(ctx, hipMemcpy(ctx->d_mat, mat4, sizeof(float) * 4 * n_spheres, hipMemcpyHostToDevice));
    RT_HIP(ctx, hipMemcpy(ctx->d_spec, spec, sizeof(float) * n_spheres, hipMemcpyHostToDevice));
    // Material's defaults (raytracer.py:43): refraction 0, refractive_index 1 (read when a cube is refractive)
    const std::vector<float2> defaults(size_t(n_spheres), make_float2(0.0f, 1.0f));
    RT_HIP(ctx, hipMemcpy(ctx->d_refr, defaults.data(), sizeof(float2) * size_t(n_spheres), hipMemcpyHostToDevice));
  }
  DevLights& L = ctx->lights;
  memcpy(L.ppos, lights->point_pos, 12); memcpy(L.pcol, lights->point_color, 12);
  L.pint = lights->point_intensity;
  memcpy(L.apos, lights->area_pos, 12); memcpy(L.au, lights->area_u, 12); memcpy(L.av, lights->area_v, 12);
  memcpy(L.acol, lights->area_color, 12);
  L.aint = lights->area_intensity; L.ns = lights->area_samples; L.inv_ns = 1.0f / float(L.ns);
  L.arad = area_light_radius(L.au, L.av);
  ctx->h_geom.assign(geom4, geom4 + size_t(4) * size_t(n_spheres));   // (rt_set_cubes builds the tables again, with the boxes)
  if (int rc = upload_occluder_table(ctx, 0)) return rc;
  ctx->list_dim = 0;
  if (ctx->reflect_table && n_spheres >= kGridFrom) {   // many spheres: the direction table as lists
    std::vector<uint32_t> start;
    std::vector<uint2> items;
    // 32 x 32 buckets per face up to kReflListFineTo spheres (config 4: lists half as long as with 16 x 16, 377 -> 387 Grays/s,
    // +30 ms of upload), 16 x 16 beyond (the start table alone would be 50 MB at 2048 spheres)
    const int list_dim = n_spheres <= kReflListFineTo ? 2 * kReflListDim : kReflListDim;
    if (build_reflect_lists(geom4, n_spheres, list_dim, start, items)) {
      const size_t sb = start.size() * sizeof(uint32_t), ib = items.size() * sizeof(uint2);
      if (sb > ctx->list_start_bytes) {
        if (ctx->d_list_start) (void)hipFree(ctx->d_list_start);
        ctx->d_list_start = nullptr; ctx->list_start_bytes = 0;
        RT_HIP(ctx, hipMalloc(&ctx->d_list_start, sb));
        ctx->list_start_bytes = sb;
      }
      if (ib > ctx->list_items_bytes) {
        if (ctx->d_list_items) (void)hipFree(ctx->d_list_items);
        ctx->d_list_items = nullptr; ctx->list_items_bytes = 0;
        RT_HIP(ctx, hipMalloc(&ctx->d_list_items, ib));
        ctx->list_items_bytes = ib;
      }
      RT_HIP(ctx, hipMemcpy(ctx->d_list_start, start.data(), sb, hipMemcpyHostToDevice));
      RT_HIP(ctx, hipMemcpy(ctx->d_list_items, items.data(), ib, hipMemcpyHostToDevice));
      ctx->list_dim = list_dim;
    }
  }
  if (int rc = upload_reflect_table(ctx, nullptr, 0)) return rc;
  RT_HIP(ctx, hipMemset(ctx->d_glass_bits, 0, sizeof(uint32_t) * (RT_MAX_SPHERES / 32)));
  {   // can a node of a pixel's tree contribute a negative share?  (see rt_ctx::sums_never_negative)
    bool ok = lights->point_intensity >= 0.f && lights->area_intensity >= 0.f;
    for (int k = 0; k < 3; ++k) ok = ok && lights->point_color[k] >= 0.f && lights->area_color[k] >= 0.f;
    for (int i = 0; i < n_spheres && ok; ++i)
      ok = mat4[4 * i] >= 0.f && mat4[4 * i + 1] >= 0.f && mat4[4 * i + 2] >= 0.f && mat4[4 * i + 3] >= 0.f && mat4[4 * i + 3] <= 1.f &&
           spec[i] >= 0.f;
    ctx->nonneg_scene = ok;
  }
  // (Material's defaults, raytracer.py:43, until rt_set_refraction says otherwise; spheres only until rt_set_cubes does)
  ctx->n_spheres = n_spheres;
  return RT_OK;
}

int rt_set_cubes(rt_ctx* ctx, const float* box6, const float* mat4, const float* spec, const float* refraction,
                 const float* refractive_index, int n_cubes) {
  if (!ctx) return fail(nullptr, RT_EINVAL, "ctx is NULL");
  if (ctx->n_spheres < 0) return fail(ctx, RT_ESTATE, "rt_upload_scene has not been called");
  if (n_cubes < 0 || ctx->n_spheres + 2 * n_cubes > RT_MAX_SPHERES)
    return fail(ctx, RT_EINVAL, "n_cubes %d out of range: n_spheres + 2 n_cubes must not exceed %d", n_cubes,
                RT_MAX_SPHERES);
  if (n_cubes > 0 && (!box6 || !mat4 || !spec)) return fail(ctx, RT_EINVAL, "cube array is NULL");
  std::vector<float4> box(size_t(2 * std::max(n_cubes, 1)));
  std::vector<float2> rf(size_t(std::max(n_cubes, 1)));
  bool any = false, nonneg = true;
  for (int i = 0; i < n_cubes; ++i) {
    for (int k = 0; k < 6; ++k)
      if (!std::isfinite(box6[6 * i + k])) return fail(ctx, RT_EINVAL, "cube %d has a non-finite corner", i);
    for (int k = 0; k < 4; ++k)
      if (!std::isfinite(mat4[4 * i + k])) return fail(ctx, RT_EINVAL, "cube %d has a non-finite material", i);
    if (!std::isfinite(spec[i])) return fail(ctx, RT_EINVAL, "cube %d has a non-finite specular", i);
    const float f = refraction ? refraction[i] : 0.0f, ri = refractive_index ? refractive_index[i] : 1.0f;
    if (!std::isfinite(f) || !std::isfinite(ri)) return fail(ctx, RT_EINVAL, "cube %d: non-finite refraction value", i);
    if (f > 0.0f && !(ri > 0.0f)) return fail(ctx, RT_EINVAL, "cube %d: refractive_index must be > 0", i);
    box[size_t(2 * i)] = make_float4(box6[6 * i], box6[6 * i + 1], box6[6 * i + 2], 0.0f);
    box[size_t(2 * i + 1)] = make_float4(box6[6 * i + 3], box6[6 * i + 4], box6[6 * i + 5], 0.0f);
    rf[size_t(i)] = make_float2(f, ri);
    any = any || f > 0.0f;
    nonneg = nonneg && mat4[4 * i] >= 0.f && mat4[4 * i + 1] >= 0.f && mat4[4 * i + 2] >= 0.f && mat4[4 * i + 3] >= 0.f &&
             mat4[4 * i + 3] <= 1.f && spec[i] >= 0.f && f >= 0.f && f <= 1.f;
  }
  RT_HIP(ctx, hipSetDevice(ctx->device));
  RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (n_cubes > 0) {
    const size_t at = size_t(ctx->n_spheres);   // cubes are objects n_spheres, n_spheres + 1, ...
    RT_HIP(ctx, hipMemcpy(ctx->d_cubes, box.data(), sizeof(float4) * 2 * size_t(n_cubes), hipMemcpyHostToDevice));
    RT_HIP(ctx, hipMemcpy(ctx->d_mat + at, mat4, sizeof(float4) * size_t(n_cubes), hipMemcpyHostToDevice));
    RT_HIP(ctx, hipMemcpy(ctx->d_spec + at, spec, sizeof(float) * size_t(n_cubes), hipMemcpyHostToDevice));
    RT_HIP(ctx, hipMemcpy(ctx->d_refr + at, rf.data(), sizeof(float2) * size_t(n_cubes), hipMemcpyHostToDevice));
    // occluder table rows of the cubes (every sphere may shadow a point of a box), behind the spheres' rows
    if (int rc = upload_occluder_table(ctx, n_cubes)) return rc;
  }
  ctx->n_cubes = 0;
  // secondary rays of a scene with boxes: the direction table again, with the boxes as sources and targets
  if (int rc = upload_reflect_table(ctx, n_cubes > 0 ? box.data() : nullptr, n_cubes)) return rc;
  ctx->n_cubes = n_cubes;
  ctx->has_refr_cubes = any;
  ctx->nonneg_cubes = nonneg;
  return RT_OK;
}

int rt_occluder_table(const float* geom4, int n_spheres, const rt_lights* lights, uint32_t* out_words) {
  if (n_spheres < 0 || n_spheres > RT_MAX_SPHERES || !lights || !out_words || (n_spheres > 0 && !geom4))
    return fail(nullptr, RT_EINVAL, "rt_occluder_table: bad argument");
  DevLights L{};
  memcpy(L.ppos, lights->point_pos, 12);
  memcpy(L.apos, lights->area_pos, 12); memcpy(L.au, lights->area_u, 12); memcpy(L.av, lights->area_v, 12);
  L.arad = area_light_radius(L.au, L.av);
  std::vector<uint32_t> tab;
  build_occluder_table(geom4, n_spheres, L, tab);
  const size_t words = size_t(n_spheres) * size_t((n_spheres + 31) / 32);
  if (words) memcpy(out_words, tab.data(), words * sizeof(uint32_t));
  return RT_OK;
}

int rt_occluder_table_buckets(const float* geom4, int n_spheres, const rt_lights* lights, int dim, uint32_t* out_words) {
  if (n_spheres < 1 || n_spheres > RT_MAX_SPHERES || !lights || !out_words || !geom4 || dim < 1 || dim > 16)
    return fail(nullptr, RT_EINVAL, "rt_occluder_table_buckets: bad argument");
  DevLights L{};
  memcpy(L.ppos, lights->point_pos, 12);
  memcpy(L.apos, lights->area_pos, 12); memcpy(L.au, lights->area_u, 12); memcpy(L.av, lights->area_v, 12);
  L.arad = area_light_radius(L.au, L.av);
  std::vector<uint32_t> object_rows, tab;
  build_occluder_table(geom4, n_spheres, L, object_rows);
  build_occluder_buckets(geom4, n_spheres, L, dim, object_rows, tab);
  memcpy(out_words, tab.data(), tab.size() * sizeof(uint32_t));
  return RT_OK;
}

int rt_reflect_table(const float* geom4, int n_spheres, int dim, uint32_t* out_words) {
  if (n_spheres < 1 || n_spheres > RT_MAX_SPHERES || dim < 1 || dim > 16 || !out_words || !geom4)
    return fail(nullptr, RT_EINVAL, "rt_reflect_table: bad argument");
  std::vector<uint2> tab;
  build_reflect_table(geom4, n_spheres, dim, tab);
  memcpy(out_words, tab.data(), tab.size() * sizeof(uint2));
  return RT_OK;
}

int rt_set_refraction(rt_ctx* ctx, const float* refraction, const float* refractive_index, int n_spheres) {
  if (!ctx) return fail(nullptr, RT_EINVAL, "ctx is NULL");
  if (ctx->n_spheres < 0) return fail(ctx, RT_ESTATE, "rt_upload_scene has not been called");
  if (n_spheres != ctx->n_spheres)
    return fail(ctx, RT_EINVAL, "n_spheres %d does not match the uploaded scene (%d)", n_spheres, ctx->n_spheres);
  if (n_spheres > 0 && (!refraction || !refractive_index)) return fail(ctx, RT_EINVAL, "NULL array");
  std::vector<float2> h(size_t(std::max(n_spheres, 1)));
  bool any = false, nonneg = true;
  for (int i = 0; i < n_spheres; ++i) {
    if (!std::isfinite(refraction[i]) || !std::isfinite(refractive_index[i]))
      return fail(ctx, RT_EINVAL, "sphere %d: non-finite refraction value", i);
    if (refraction[i] > 0.0f && !(refractive_index[i] > 0.0f))
      return fail(ctx, RT_EINVAL, "sphere %d: refractive_index must be > 0", i);
    h[size_t(i)] = make_float2(refraction[i], refractive_index[i]);
    any = any || refraction[i] > 0.0f;
    nonneg = nonneg && refraction[i] >= 0.0f && refraction[i] <= 1.0f;
  }
  RT_HIP(ctx, hipSetDevice(ctx->device));
  RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (n_spheres > 0)
    RT_HIP(ctx, hipMemcpy(ctx->d_refr, h.data(), sizeof(float2) * size_t(n_spheres), hipMemcpyHostToDevice));
  {
    std::vector<uint32_t> bits(RT_MAX_SPHERES / 32, 0u);
    for (int i = 0; i < n_spheres; ++i)
      if (refraction[i] > 0.0f) bits[size_t(i >> 5)] |= 1u << (i & 31);
    RT_HIP(ctx, hipMemcpy(ctx->d_glass_bits, bits.data(), sizeof(uint32_t) * bits.size(), hipMemcpyHostToDevice));
  }
  ctx->has_refr = any;
  ctx->nonneg_refr = nonneg;
  return RT_OK;
}

int rt_render(rt_ctx* ctx, const rt_camera* cam, int width, int height, int samples, int max_depth,
              int rng_mode, uint64_t seed, uint8_t* out_rgb, int32_t* out_hit) {
  if (!out_rgb) return fail(ctx, RT_EINVAL, "out_rgb is NULL");
  Window w{0, width, height, height > 0 ? height : 1, 0, 1, 0};
  return launch(ctx, cam, &seed, 1, width, height, w, samples, max_depth, rng_mode, out_rgb, nullptr, out_hit);
}

int rt_render_noalloc(rt_ctx* ctx, const rt_camera* cam, int width, int height, int samples, int max_depth,
                      int rng_mode, uint64_t seed, uint8_t* out_rgb, int32_t* out_hit) {
  if (!ctx) return fail(nullptr, RT_EINVAL, "ctx is NULL");
  if (!out_rgb) return fail(ctx, RT_EINVAL, "out_rgb is NULL");
  unsigned char* ws = ctx->scratch;
  if (!ws) return fail(ctx, RT_ESTATE, "rt_render_noalloc: no scratch array on device %d", ctx->device);
  Window w{0, width, height, height > 0 ? height : 1, 0, 1, 0};
  return launch(ctx, cam, &seed, 1, width, height, w, samples, max_depth, rng_mode, out_rgb, nullptr, out_hit, 0, ws);
}

static int rows_owned(int height, int tile_rows, int tile_first, int tile_stride) {
  long long rows = 0;
  const int n_tiles = (height + tile_rows - 1) / tile_rows;
  for (int t = tile_first; t < n_tiles; t += tile_stride) {
    const int r0 = t * tile_rows;
    rows += (r0 + tile_rows <= height) ? tile_rows : (height - r0);
  }
  return int(rows);
}

int rt_render_rows(rt_ctx* ctx, const rt_camera* cam, int width, int height, int samples,
                   int max_depth, int rng_mode, uint64_t seed, int tile_rows, int tile_first,
                   int tile_stride, uint8_t* out_rgb, int32_t* out_hit, int* n_local_rows_out) {
  return rt_render_frames_rows(ctx, cam, &seed, 1, width, height, samples, max_depth, rng_mode, tile_rows,
                               tile_first, tile_stride, 0, out_rgb, out_hit, n_local_rows_out);
}

int rt_render_frames_rows(rt_ctx* ctx, const rt_camera* cams, const uint64_t* seeds, int n_frames,
                          int width, int height, int samples, int max_depth, int rng_mode,
                          int tile_rows, int tile_first, int tile_stride, int64_t out_frame_stride_px,
                          uint8_t* out_rgb, int32_t* out_hit, int* n_local_rows_out) {
  if (n_local_rows_out) *n_local_rows_out = 0;
  if (n_frames < 0) return fail(ctx, RT_EINVAL, "n_frames < 0");
  if (tile_rows < 1 || tile_stride < 1 || tile_first < 0 || tile_first >= tile_stride)
    return fail(ctx, RT_EINVAL, "bad tiling: tile_rows=%d tile_first=%d tile_stride=%d", tile_rows,
                tile_first, tile_stride);
  if (width <= 0 || height <= 0) return fail(ctx, RT_EINVAL, "width/height out of range: %dx%d", width, height);
  const int rows = rows_owned(height, tile_rows, tile_first, tile_stride);
  if (n_local_rows_out) *n_local_rows_out = rows;
  if (rows == 0 || n_frames == 0) return RT_OK;
  if (!out_rgb) return fail(ctx, RT_EINVAL, "out_rgb is NULL");
  Window w{0, width, rows, tile_rows, tile_first, tile_stride, 0};
  if (out_frame_stride_px < 0) return fail(ctx, RT_EINVAL, "negative out_frame_stride_px");
  return launch(ctx, cams, seeds, n_frames, width, height, w, samples, max_depth, rng_mode, out_rgb, nullptr,
                out_hit, out_frame_stride_px);
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
  return launch(ctx, cam, &seed, 1, width, height, w, samples, max_depth, rng_mode, nullptr, out_rgbf, nullptr);
}

int rt_render_frames(rt_ctx* ctx, const rt_camera* cams, const uint64_t* seeds, int n_frames,
                     int width, int height, int samples, int max_depth, int rng_mode, uint8_t* out_rgb) {
  if (n_frames < 0) return fail(ctx, RT_EINVAL, "n_frames < 0");
  if (n_frames == 0) return RT_OK;
  if (!cams || !seeds || !out_rgb) return fail(ctx, RT_EINVAL, "cams/seeds/out_rgb is NULL");
  Window w{0, width, height, height > 0 ? height : 1, 0, 1, 0};
  return launch(ctx, cams, seeds, n_frames, width, height, w, samples, max_depth, rng_mode, out_rgb, nullptr,
                nullptr);
}

int rt_scatter_rows(rt_ctx* ctx, const uint8_t* gathered, int64_t rank_stride_bytes, int n_ranks,
                    int width, int height, int tile_rows, uint8_t* out_rgb) {
  return rt_scatter_rows_frames(ctx, gathered, rank_stride_bytes, 0, n_ranks, 1, width, height, tile_rows, out_rgb);
}

int rt_scatter_rows_frames(rt_ctx* ctx, const uint8_t* gathered, int64_t rank_stride_bytes,
                           int64_t frame_stride_bytes, int n_ranks, int n_frames, int width, int height,
                           int tile_rows, uint8_t* out_rgb) {
  if (!ctx) return fail(nullptr, RT_EINVAL, "ctx is NULL");
  if (!gathered || !out_rgb) return fail(ctx, RT_EINVAL, "gathered/out_rgb is NULL");
  if (n_ranks < 1 || width <= 0 || height <= 0 || tile_rows < 1 || height > 65535 || n_frames < 1 || n_frames > 65535)
    return fail(ctx, RT_EINVAL, "bad scatter arguments");
  if (n_frames > 1 && frame_stride_bytes < rows_owned(height, tile_rows, 0, n_ranks) * (long long)width * 3)
    return fail(ctx, RT_EINVAL, "frame_stride_bytes %lld too small", (long long)frame_stride_bytes);
  // every rank's slab must hold its rows
  const long long max_rows = rows_owned(height, tile_rows, 0, n_ranks);
  if (rank_stride_bytes < 0 || (n_ranks > 1 && rank_stride_bytes < max_rows * width * 3))
    return fail(ctx, RT_EINVAL, "rank_stride_bytes %lld too small", (long long)rank_stride_bytes);
  const long long row_bytes = (long long)width * 3;
  const int vec4 = (row_bytes % 4 == 0) && (rank_stride_bytes % 4 == 0) && (frame_stride_bytes % 4 == 0) &&
                   (((long long)height * row_bytes) % 4 == 0) &&
                   ((reinterpret_cast<uintptr_t>(gathered) | reinterpret_cast<uintptr_t>(out_rgb)) % 4 == 0);
  const int per_row = int(vec4 ? row_bytes / 4 : row_bytes);
  const unsigned gx = unsigned(std::min(8, (per_row + 255) / 256));
  RT_HIP(ctx, hipSetDevice(ctx->device));
  hipLaunchKernelGGL(rt_scatter_rows_kernel, dim3(gx, unsigned(height), unsigned(n_frames)), dim3(256), 0, ctx->stream,
                     gathered, (long long)rank_stride_bytes, n_ranks, width, height, tile_rows, out_rgb, vec4,
                     (long long)frame_stride_bytes, (long long)height * row_bytes);
  RT_HIP(ctx, hipGetLastError());
  return RT_OK;
}

int rt_compact_tiles(rt_ctx* ctx, const uint8_t* slab, uint8_t* entries, uint32_t* count_dev, int capacity_tiles) {
  if (!ctx) return fail(nullptr, RT_EINVAL, "ctx is NULL");
  if (!slab || !entries || !count_dev || capacity_tiles < 0) return fail(ctx, RT_EINVAL, "rt_compact_tiles: bad argument");
  if (ctx->last.out_rgb != slab || !slab)
    return fail(ctx, RT_ESTATE, "rt_compact_tiles: `slab` is not the frame buffer the last render call on this context wrote");
  const TileGeom& g = ctx->last.g;
  const long long total = (long long)g.tile_pitch * g.tiles8_y * g.n_frames;
  if (total >= (1ll << 31)) return fail(ctx, RT_EINVAL, "rt_compact_tiles: %lld tile ids in the batch (at most 2^31 - 1)", total);
  RT_HIP(ctx, hipSetDevice(ctx->device));
  RT_HIP(ctx, hipMemsetAsync(count_dev, 0, sizeof(uint32_t), ctx->stream));
  if (total == 0) return RT_OK;
  // a wave takes n_chunks x 64 consecutive slots of one of the kTileCounters counters (rt_compact_tiles_kernel): as many as
  // leave about one wave per wave slot of the GPU — fewer atomics on the one count, yet enough waves to fill the machine
  const long long slots = (total + kTileCounters - 1) / kTileCounters;
  int n_chunks = int(std::min<long long>(kCompactChunks, std::max<long long>(1, total / (64 * 8192))));
  const long long waves = ((slots + 64 * n_chunks - 1) / (64 * n_chunks)) * kTileCounters;
  const dim3 grid(unsigned((waves + kCompactThreads / 64 - 1) / (kCompactThreads / 64)));
  const uint32_t* occ = reinterpret_cast<const uint32_t*>(ctx->d_chunk) + kTileCounters * 32;
  if (ctx->last.occ_valid)
    hipLaunchKernelGGL(rt_compact_tiles_kernel<false>, grid, dim3(kCompactThreads), 0, ctx->stream, slab, g, occ, ctx->last.occ_words,
                       entries, count_dev, (unsigned int)capacity_tiles, n_chunks);
  else
    hipLaunchKernelGGL(rt_compact_tiles_kernel<true>, grid, dim3(kCompactThreads), 0, ctx->stream, slab, g, (const uint32_t*)nullptr, 0,
                       entries, count_dev, (unsigned int)capacity_tiles, n_chunks);
  RT_HIP(ctx, hipGetLastError());
  return RT_OK;
}

int rt_place_tiles(rt_ctx* ctx, const uint8_t* entries, const uint32_t* counts_dev, int n_ranks, int capacity_tiles,
                   int n_frames, int width, int height, int tile_rows, uint32_t* table, int64_t table_words,
                   uint8_t* out_rgb) {
  if (!ctx) return fail(nullptr, RT_EINVAL, "ctx is NULL");
  if (!entries || !counts_dev || !table || !out_rgb) return fail(ctx, RT_EINVAL, "rt_place_tiles: NULL argument");
  if (n_ranks < 1 || capacity_tiles < 0 || n_frames < 1 || n_frames > 65535 || width <= 0 || height <= 0 || height > 65535 ||
      tile_rows < 1)
    return fail(ctx, RT_EINVAL, "rt_place_tiles: bad argument");
  const int rows_max = rows_owned(height, tile_rows, 0, n_ranks);            // rank 0 owns the most rows
  const long long ids_per_rank = (long long)(((width + 7) / 8) | 1) * ((rows_max + 7) / 8) * n_frames;
  if (table_words < ids_per_rank * n_ranks)
    return fail(ctx, RT_EINVAL, "rt_place_tiles: table holds %lld words, %lld needed", (long long)table_words,
                (long long)(ids_per_rank * n_ranks));
  RT_HIP(ctx, hipSetDevice(ctx->device));
  RT_HIP(ctx, hipMemsetAsync(table, 0, size_t(ids_per_rank) * size_t(n_ranks) * sizeof(uint32_t), ctx->stream));
  if (capacity_tiles > 0) {
    hipLaunchKernelGGL(rt_index_tiles_kernel, dim3(unsigned(std::min(1024, (capacity_tiles + 255) / 256)), unsigned(n_ranks)), dim3(256), 0,
                       ctx->stream, entries, counts_dev, capacity_tiles, ids_per_rank, table);
    RT_HIP(ctx, hipGetLastError());
  }
  const long long row_bytes = (long long)width * 3;
  int vec4 = (row_bytes % 4 == 0) && (reinterpret_cast<uintptr_t>(out_rgb) % 4 == 0) && (reinterpret_cast<uintptr_t>(entries) % 4 == 0);
  if (vec4 && row_bytes % 16 == 0 && reinterpret_cast<uintptr_t>(out_rgb) % 16 == 0) vec4 = 2;   // 16-byte stores
  // (the fast path gives a lane 16-byte columns of a band: as few turns as workgroups of up to 384 lanes need, and lanes to match
  // — a 1080p row's 360 columns in one turn of 384; three turns of 128 measured 9 % slower: each turn waits for its table
  // look-ups and then for its tiles)
  unsigned place_threads = 256;
  if (vec4 == 2 && (tile_rows & 7) == 0) {
    const long long cols = row_bytes / 16, turns = (cols + 383) / 384;
    place_threads = unsigned(std::min<long long>(384, ((cols + turns - 1) / turns + 63) / 64 * 64));
  }
  hipLaunchKernelGGL(rt_place_tiles_kernel, dim3(unsigned((height + 7) / 8), unsigned(n_frames)),
                     dim3(place_threads), 0, ctx->stream, entries, (const uint32_t*)table, capacity_tiles, ids_per_rank, n_ranks, width, height,
                     tile_rows, out_rgb, vec4);
  RT_HIP(ctx, hipGetLastError());
  return RT_OK;
}

int rt_get_ray_counts(rt_ctx* ctx, uint64_t out[4]) {
  if (!ctx || !out) return fail(ctx, RT_EINVAL, "ctx/out is NULL");
  unsigned long long host[kCounterSlots * kCounterWords];
  RT_HIP(ctx, hipSetDevice(ctx->device));
  RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
  RT_HIP(ctx, hipMemcpy(host, ctx->d_counters, sizeof host, hipMemcpyDeviceToHost));
  unsigned long long trace = 0, shade = 0, primary = 0, shadow = 0;
  for (int s = 0; s < kCounterSlots; ++s) {
    const unsigned long long* h = host + s * kCounterWords;
    trace += h[0]; shade += h[1]; primary += h[2]; shadow += h[3];
  }
  out[0] = primary + ctx->host_primaries;
  out[1] = trace - primary;
  out[2] = shadow;
  out[3] = shade;
  return RT_OK;
}

int rt_debug_counters(rt_ctx* ctx, uint64_t out[16]) {
  if (!ctx || !out) return fail(ctx, RT_EINVAL, "ctx/out is NULL");
  unsigned long long host[kCounterSlots * kCounterWords];
  RT_HIP(ctx, hipSetDevice(ctx->device));
  RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
  RT_HIP(ctx, hipMemcpy(host, ctx->d_counters, sizeof host, hipMemcpyDeviceToHost));
  for (int k = 0; k < kCounterWords; ++k) {
    out[k] = 0;
    for (int s = 0; s < kCounterSlots; ++s) out[k] += host[s * kCounterWords + k];
  }
  out[0] += ctx->host_primaries; out[2] += ctx->host_primaries;
#if RT_CULL_STATS
  {  // culling statistics: [4] candidate spheres summed over lanes, [5] lane-chunks, [6] drain-loop turns summed over lanes
    unsigned long long h[4];
    RT_HIP(ctx, hipMemcpyFromSymbol(h, HIP_SYMBOL(g_dbg), sizeof h));
    out[4] = h[0]; out[5] = h[1]; out[6] = h[2]; out[7] = h[3];   // [7] cull-loop turns summed over lanes
    unsigned long long z[4] = {0, 0, 0, 0};
    RT_HIP(ctx, hipMemcpyToSymbol(HIP_SYMBOL(g_dbg), z, sizeof z));
  }
#endif
  return RT_OK;
}

int rt_reset_ray_counts(rt_ctx* ctx) {
  if (!ctx) return fail(nullptr, RT_EINVAL, "ctx is NULL");
  RT_HIP(ctx, hipSetDevice(ctx->device));
  RT_HIP(ctx, hipMemsetAsync(ctx->d_counters, 0, sizeof(unsigned long long) * kCounterSlots * kCounterWords, ctx->stream));
  ctx->host_primaries = 0;
  return RT_OK;
}

}  // extern "C"
