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

// ------------------------------------------------------------------------------------------
// kernel parameters (passed by value: the kernarg segment is read with scalar loads)
// ------------------------------------------------------------------------------------------
struct DevLights {
  float ppos[3], pcol[3], pint;
  float apos[3], au[3], av[3], acol[3], aint;
  int ns;
  float arad;   // 0.5 (|au| + |av|): every area-light sample lies within this distance of apos
  float inv_ns; // RN(1 / ns): what `1.0f / float(ns)` gives, formed once on the host instead of per shaded hit (10 instructions)
};

// frames per launch: the fixed cost of a launch pair (ramp-up, the stream kernel's uneven tail, the
// gaps between dependent launches: ~0.2 ms) is paid once per batch (measured on config 2 while the batch grew:
// 16 -> 64 frames +12 %, 64 -> 256 +15 %, 256 -> 1024 +4 %)
constexpr int kMaxBatch = RT_FRAMES_PER_LAUNCH;
constexpr long long kRefrSumBytes = 32ll << 30;   // refractive scenes: most memory the float pixel sums of a launch may take
constexpr long long kVisRecordBytes = 24ll << 30; // 2 .. 64 samples in one launch: most memory the tile records of a launch may take
struct CamGroup;               // primary-ray table entry (four spheres), defined with primary_hit
static_assert(kMaxBatch <= 8192, "frame numbers travel in 13 bits of the work-queue records");
static_assert(RT_MAX_SPHERES <= 4096 && RT_MAX_DEPTH < 127, "object index: 12 bits, level: 7 bits of a record's meta word");
// meta word of a work-queue record: level (7 bits) | frame (13 bits) | object index (12 bits)
__device__ __forceinline__ uint32_t pack_meta(int level, int frame, int object) {
  return uint32_t(level) | (uint32_t(frame) << 7) | (uint32_t(object) << 20);
}
__device__ __forceinline__ int meta_level(uint32_t m) { return int(m & 127u); }
__device__ __forceinline__ int meta_frame(uint32_t m) { return int((m >> 7) & 8191u); }
__device__ __forceinline__ int meta_object(uint32_t m) { return int(m >> 20); }

// Per-frame inputs of a batch.  A launch's kernel arguments are limited to 4 KB, so the cameras of a batch
// (up to RT_FRAMES_PER_LAUNCH frames) live in device memory: rt_store_frames_kernel writes them there from kernel arguments,
// 64 frames per (tiny) launch — no host-to-device copy, so the host never waits for the stream.
struct __attribute__((aligned(64))) FrameRec {
  rt_camera cam;
  uint32_t seed;
  uint32_t pad[3];
};
constexpr int kFrameChunk = 64;
struct FrameChunk {
  rt_camera cam[kFrameChunk];
  uint32_t seed[kFrameChunk];
};

struct RenderParams {
  DevLights L;
  int width, height;     // full image, defines the primary rays (raytracer.py:290-295)
  int x0, nx;            // column window [x0, x0+nx)
  int n_rows;            // local rows rendered by this launch
  int tile_rows, tile_first, tile_stride;  // local row lr -> image row, see row_of()
  int y0;                // extra row offset (render_chunk)
  int sample_index, n_samples;  // this launch traces sample `sample_index` of n_samples
  int max_depth;
  int n_spheres;
  int n_frames, strips_x, strips_y;
  float inv_w, inv_h;    // RN(1/width), RN(1/height): div_exact() turns them into true quotients
  float inv_aspect;      // RN(height/width)
  float inv_samples;     // RN(1/n_samples)
  const float4* geom;    // cx cy cz r^2
  const float4* mat;     // R G B reflection
  const float* spec;     // specular
  const float2* refr;    // refraction, refractive_index (raytracer.py:43); only read by the REFR kernels
  const uint32_t* glass_bits;  // [ceil(n_spheres / 32)] bit c: sphere c is refractive (REFR kernels' shadow scan)
  const uint32_t* occl;  // [objects][occl_words] bit c of row k: sphere c can shadow some point of object k (host table)
  int occl_words;        // ceil(n_spheres / 32)
  // the same table as lists, for scenes with many spheres (the GRID kernels; 32 .. 64 mostly empty words per row otherwise):
  // row k is the run of 4-index chunks occl_start[k] .. occl_start[k + 1] of occl_items (uint16, ascending, padded with 0xFFFF)
  const uint32_t* occl_start;
  const uint2* occl_items;
  // direction table for reflection rays (refl_dim = 0: none): [object][6 refl_dim^2 direction buckets][refl_pairs] 64-bit
  // masks of the spheres a ray that leaves the object in a direction of the bucket can hit (nearest_table)
  const uint2* refl_tab;
  int refl_dim, refl_pairs;
  // the same table as lists, for scenes with too many spheres for masks (list_dim = 0: none): entry [object][bucket] is the
  // run of 4-index chunks list_start[e] .. list_start[e + 1] of list_items (uint16 indices, ascending, padded with 0xFFFF)
  const uint32_t* list_start;
  const uint2* list_items;
  int list_dim;
  const float4* cubes;   // [n_cubes][2] min xyz -, max xyz - (raytracer.py:56-60); objects n_spheres.. of mat/spec/refr
  int n_cubes;           // > 0: the CUBES kernels run
  int defer_output;      // REFR path: pixels are resolved from `accum` after the last sample (rt_resolve_kernel)
  int tile_flush;        // the stream kernel stores every occupied tile whole (its SLOTS variant): the visibility pass
                         // writes the pixels of empty tiles only
  long long vis_sample_stride;  // all samples shaded by one stream-kernel launch (the kernels' MULTI variants): elements
                                // between the tile records of consecutive samples of the batch (sample-major: sample 0's first)
  unsigned long long* present;  // MULTI: [tile ids of the batch] bit s: sample s of the tile hit something (zeroed per batch)
  const CamGroup* camtab;  // [frame][n4 / 4] primary_hit tables
  int16_t* vis;            // [frame][tiles8_y][tiles8_x][kVisTile] per 8x8 tile: 64 primary hit indices (-1 = miss /
                           // outside the window) + header {tx | ty << 16, frame}; visibility pass -> stream kernel
  long long vis_stride;    // elements between frames of vis
  int* chunk_counter;      // dynamic work distribution of the stream kernel (zeroed per launch)
  uint32_t* occ;           // [kTileCounters][occ_words] one bit per tile: some pixel of it hit something (zeroed per
  int occ_words;           // launch, set by the visibility pass): word h * occ_words + (slot >> 5), bit slot & 31
  int tiles8_x, tiles8_y;
  int tile_pitch;          // tile ids per tile row: tiles8_x made odd, so that the tiles of one work counter (ids
                           // congruent mod kTileCounters) form a lattice over the whole frame instead of a few columns
  uint8_t* out_rgb;      // [frame][n_rows][nx][3] or null
  float* out_f32;        // [frame][n_rows][nx][3] or null (render_chunk)
  int32_t* out_hit;      // [frame][n_rows][nx] or null
  float* accum;          // [frame][n_rows][nx][3] running sum over samples (n_samples > 1)
  unsigned long long* counters;  // [kCounterSlots][8]
  long long frame_stride_px;     // pixels between frames in out_*
  long long accum_stride_px;     // pixels between frames in accum
  const FrameRec* frames;  // [n_frames] camera and folded seed of each frame of the batch (device; rt_store_frames_kernel)
  float* spill;            // REFR: [wave][spill_blocks][kSpillFields][64] queue records set aside (stream kernel)
  int spill_blocks;        // blocks of 64 records per wave in `spill`
};

#ifndef RT_REFL_TABLE_FROM
#define RT_REFL_TABLE_FROM 4    // reflection rays: scenes with at least this many spheres get a direction table (nearest_table);
                                // measured on config 2 (8 spheres): +1.6 % over the plain scan, 64 spheres: +33 %
#endif
#ifndef RT_REFL_LIST_DIM
#define RT_REFL_LIST_DIM 16     // scenes with too many spheres for masks (>= the grid's threshold): buckets per face edge of the list form
#endif
#ifndef RT_VIS_TILES
#define RT_VIS_TILES 4     // 8x8 tiles (pixels per lane) of one visibility-pass wave: 2 -> +5 %, 4 -> +8 % on config 2 (87 VGPRs)
#endif
#ifndef RT_VIS_WAVES
#define RT_VIS_WAVES 5     // its __launch_bounds__ occupancy target (waves per SIMD); 8 with 3+ tiles spills
#endif
constexpr int kVisTiles = RT_VIS_TILES;
#ifndef RT_VIS_SKIP
#define RT_VIS_SKIP 1      // the visibility pass culls per wave at any sphere count: blocks that no sphere reaches build no rays
#endif
#ifndef RT_VIS_CULL_FROM
#define RT_VIS_CULL_FROM 4    // scenes with more spheres than this: a block tests only the table groups that hold a sphere its cone reaches (primary_hit_culled; 16 -> 4: +0.9 % on config 2)
#endif
constexpr int kVisCullFrom = RT_VIS_CULL_FROM;
constexpr int kBlock = 256;        // 4 waves
constexpr int kVisTile = 72;       // int16 elements per tile record of RenderParams::vis (144 bytes)
constexpr int kTileCounters = 32;  // work counters of the stream kernel: tile id = slot * kTileCounters + counter
constexpr int kCounterSlots = 64;  // ray counters are spread over 64 cache lines
constexpr int kCounterWords = 16;  // per slot: 0 trace 1 shade 2 primary 3 shadow | 8.. phase stamps
#if RT_STAMPS
#define RT_NOW() ((unsigned long long)clock64())
#else
#define RT_NOW() 0ull
#endif
constexpr int kShadowBatch = 13;   // 1 point-light ray + 12 area-light rays (raytracer.py:80)
#ifndef RT_REFR_BATCH
#define RT_REFR_BATCH 7
#endif
#ifndef RT_REFR_WAVES
#define RT_REFR_WAVES 4     // waves per SIMD the REFR slot variants are compiled for (their LDS holds three workgroups per CU)
#endif
// Scenes with boxes: 13 rays at a time (7 with refraction) left the shading kernels 34 VGPRs short — scratch traffic of
// 1.2e10 bytes per 1024 frames on cube_scene(), 1.4e11 with glass (TCC_EA0_WRREQ; round 2 took the latter for the float
// atomics' doing).  Batches of 7 / 5 fit the registers (no scratch); the later batches reuse the first one's cull.
// cube_scene() 328 -> 376 Grays/s, with glass 235 -> 259 (9 / 7: 339 / 236, 5 / 4: 352 / 247).
#ifndef RT_CUBE_BATCH
#define RT_CUBE_BATCH 7
#endif
#ifndef RT_REFR_CUBE_BATCH
#define RT_REFR_CUBE_BATCH 5
#endif
constexpr int kCubeShadowBatch = RT_CUBE_BATCH;            // scenes with boxes
constexpr int kRefrCubeShadowBatch = RT_REFR_CUBE_BATCH;   // ... and refractive materials
constexpr int kRefrShadowBatch = RT_REFR_BATCH;   // refractive scenes: two verdicts per ray (opaque / glass), so fewer rays per batch fit the registers

// pixel kernel (one thread per pixel): 32 x 8 pixel tile per workgroup
constexpr int kTileW = 32, kTileH = 8;

// ------------------------------------------------------------------------------------------
// keyed RNG — identical to oracle/ref_harness.py / oracle/rt_oracle.cpp
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t fmix32(uint32_t h) {
  h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
  return h;
}
__device__ __forceinline__ uint32_t mix(uint32_t h, uint32_t v) { return fmix32(h + 0x9E3779B9u + v); }
__device__ __forceinline__ uint32_t pixel_key_of(uint32_t seed32, uint32_t x, uint32_t y, uint32_t sample) {
  return fmix32(seed32 + x * 0x9E3779B9u + y * 0x85EBCA6Bu + sample * 0xC2B2AE35u);
}
// Pair j of the stream `key`: ONE hash (two multiplies) gives two 16-bit draws, returned as 1 + draw in [1, 2) —
// the 16 bits go straight into a float's mantissa (shift, and-or: no half-rate integer conversion), and
// `r - 1.5f` / `r - 1.0f` are exact, so the values are those of (h >> 16) * 2^-16 and (h & 0xFFFF) * 2^-16.
template <int MODE>
__device__ __forceinline__ void draw_pair1(uint32_t key, uint32_t j, float& first1, float& second1) {
  if (MODE == RT_RNG_CONSTANT) { first1 = 1.5f; second1 = 1.5f; return; }
  uint32_t h = key + j * 0x9E3779B9u + 0x7F4A7C15u;
  h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
  first1 = __uint_as_float(((h >> 9) & 0x007FFF80u) | 0x3F800000u);
  second1 = __uint_as_float(((h << 7) & 0x007FFF80u) | 0x3F800000u);
}

// 1-ulp hardware reciprocal / reciprocal square root / square root (v_rcp_f32, v_rsq_f32,
// v_sqrt_f32: quarter-rate single instructions) instead of the ~10-25 instruction IEEE expansions;
// the FP64 reference is ~1e-7 away either way.
__device__ __forceinline__ float frcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float frsq(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ float fsqrt(float x) { return __builtin_amdgcn_sqrtf(x); }


// wave-uniform reads of the constant sphere tables: address space 4 makes hipcc emit scalar loads
// (s_load_dwordx4..x16 into SGPRs) — no LDS traffic, no VGPRs, and SGPR operands feed the VALU
#if defined(__HIP_DEVICE_COMPILE__)
typedef const float4 __attribute__((address_space(4))) * cgeom_t;
__device__ __forceinline__ cgeom_t as_const(const float4* p) { return (cgeom_t)(unsigned long long)(p); }
#else   // host pass only parses the kernels
typedef const float4* cgeom_t;
__device__ __forceinline__ cgeom_t as_const(const float4* p) { return p; }
#endif

// max(0, x) in ONE instruction: fmaxf() compiles to a second v_max that only quiets a signalling NaN (which cannot
// occur here), and v_max is a half-rate instruction on gfx950
__device__ __forceinline__ float max0(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  float r;
  asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(x));
  return r;
#else
  return x > 0.f ? x : 0.f;
#endif
}
__device__ __forceinline__ float pow50(float x) {  // `** 50` (raytracer.py:193,208), x >= 0
  float x2 = x * x, x4 = x2 * x2, x8 = x4 * x4, x16 = x8 * x8, x32 = x16 * x16;
  return x32 * x16 * x2;
}

// a / b, correctly rounded, given rb = RN(1/b): q = RN(a*rb) is within an ulp, the fma residual
// a - q*b is exact, and RN(q + r*rb) is then the correctly rounded quotient (Markstein's theorem; b
// is an image size <= 2^20, so its significand is never all ones).  Three full-rate ops instead of
// the ~11 of the IEEE division expansion; tools/check_div.c verifies bit-equality with `/`.
__device__ __forceinline__ float div_exact(float a, float b, float rb) {
  const float q = a * rb;
  return fmaf(fmaf(-q, b, a), rb, q);
}

// image row of local row lr (rt_render_rows tiling; full frames use tile_rows = height)
__device__ __forceinline__ int row_of(const RenderParams& p, int lr) {
  int j = lr / p.tile_rows;
  return p.y0 + (p.tile_first + j * p.tile_stride) * p.tile_rows + (lr - j * p.tile_rows);
}

// image row of window row row0 + ly, row0 a multiple of 8: wave-uniform arithmetic unless a row tile
// is not a multiple of 8 rows (tile_rows_uniform() false)
__device__ __forceinline__ bool tile_rows_uniform(const RenderParams& p) {
  return p.tile_stride == 1 || (p.tile_rows & 7) == 0;
}
__device__ __forceinline__ int tile_row_base(const RenderParams& p, int row0) {
  if (p.tile_stride == 1) return p.y0 + p.tile_first * p.tile_rows + row0;   // contiguous rows
  const int j = row0 / p.tile_rows;                                          // scalar division
  return p.y0 + (p.tile_first + j * p.tile_stride) * p.tile_rows + (row0 - j * p.tile_rows);
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
__device__ __forceinline__ void nearest_hit(const float4* __restrict__ sg, const float4* __restrict__ gg,
                                            int n, float ox, float oy, float oz, float dx, float dy,
                                            float dz, float& best_t, int& best_i) {
  // only the fused multiply-adds written out below: the hit distance must come out the same to the
  // last bit wherever this test is inlined (sphere_t_exact), whatever the optimiser
  // would contract in each context
#pragma clang fp contract(off)
  const float inv_a = frcp(fmaf(dz, dz, fmaf(dy, dy, dx * dx)));  // :95, computed, not assumed 1
  best_t = INFINITY;
  best_i = -1;
  // Scalar loads: four spheres per turn (64 B), the next four requested before these are tested, so the load's
  // latency passes under ~60 vector instructions instead of being waited for sphere by sphere.  The list is
  // padded with eight never-hit spheres (r^2 = -inf, rt_upload_scene), so neither the last block nor the
  // prefetch behind it needs a bound.
  {
    cgeom_t src = as_const(gg);
    auto test = [&](const float4 s, int i) {
      const float cx = s.x - ox, cy = s.y - oy, cz = s.z - oz;
      const float q = fmaf(cz, dz, fmaf(cy, dy, cx * dx)) * inv_a;
      const float lx = fmaf(-q, dx, cx), ly = fmaf(-q, dy, cy), lz = fmaf(-q, dz, cz);
      const float dl = fmaf(-lz, lz, fmaf(-ly, ly, fmaf(-lx, lx, s.w)));  // r^2 - |l|^2
      if (dl >= 0.0f) {
        const float t = q - fsqrt(dl * inv_a);
        if (t > 0.0f && t < best_t) { best_t = t; best_i = i; }
      }
    };
    float4 n0 = src[0], n1 = src[1], n2 = src[2], n3 = src[3];
    for (int i = 0; i < n; i += 4) {
      const float4 c0 = n0, c1 = n1, c2 = n2, c3 = n3;
      n0 = src[i + 4]; n1 = src[i + 5]; n2 = src[i + 6]; n3 = src[i + 7];
      test(c0, i); test(c1, i + 1); test(c2, i + 2); test(c3, i + 3);
    }
  }
}

// ------------------------------------------------------------------------------------------
// Primary rays (render_pixel :295-296 -> trace_ray -> find_nearest_intersection): every ray of a
// frame starts at the camera, so co = C - O and c = |co|^2 - r^2 are the same for all lanes.  The
// workgroup computes them once per sphere (c in FP64, spheres that contain the camera get c = +inf:
// their near root is negative, :102) into LDS, and a test is  nb = co.D, disc' = nb^2 - c  — 4 VALU
// ops.  Candidates (disc' >= 0, nb > 0) of four spheres are detected with one sign-bit test so the
// root is only evaluated where some lane needs it.  D is a unit vector here (a = 1 +- 2e-7).  The
// plain b^2-4ac form is good enough to CHOOSE the sphere (its error, ~|co|^2 * 1e-7, moves a
// silhouette by < 1 % of a pixel and is far inside the grazing-tie margin); the hit distance of
// the chosen sphere is then recomputed once in the cancellation-free form of nearest_hit.
// ------------------------------------------------------------------------------------------
// intersect_sphere (raytracer.py:93-104) for ONE known sphere in the cancellation-free form of
// nearest_hit: returns t (valid iff dl >= 0 and t > 0).
__device__ __forceinline__ float sphere_t_exact(const float4 s, float ox, float oy, float oz, float dx,
                                                float dy, float dz, float inv_a, float& dl) {
#pragma clang fp contract(off)
  const float cx = s.x - ox, cy = s.y - oy, cz = s.z - oz;
  const float q = fmaf(cz, dz, fmaf(cy, dy, cx * dx)) * inv_a;
  const float lx = fmaf(-q, dx, cx), ly = fmaf(-q, dy, cy), lz = fmaf(-q, dz, cz);
  dl = fmaf(-lz, lz, fmaf(-ly, ly, fmaf(-lx, lx, s.w)));
  return q - fsqrt(dl * inv_a);
}

// ------------------------------------------------------------------------------------------
// Cubes: members of scene['spheres'] that are not Sphere objects go through
// intersect_cube / cube_normal (raytracer.py:157-160, :164-169).
//
// intersect_cube (:106-137), slab test.  `inv` = 1/d per axis, only used where |d| >= 1e-8 (:124).
// Returns whether find_nearest_intersection would consider the result (`if t and ...`, :162), i.e. not
// None and not 0.  As in the reference, t is t_max WITHOUT a sign check when t_min <= 0 (:137): a box
// behind the origin yields t < 0 and is accepted by :162 (`t < min_distance`), so it beats every sphere.
// `face` = 2 * axis + (0: min plane, 1: max plane) names the plane that produced t.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ bool cube_t(const float4 lo, const float4 hi, float ox, float oy, float oz, float dx,
                                       float dy, float dz, float ix, float iy, float iz, float& t, int& face) {
  float tmin = -INFINITY, tmax = INFINITY;                        // :107-108
  int fmin = 0, fmax = 0;
  bool ok = true;
  auto axis = [&](float o, float d, float inv, float mn, float mx, int a) {
    if (fabsf(d) < 1e-8f) {                                       // :124-126
      ok = ok && !(o < mn || o > mx);
    } else {
      float t1 = (mn - o) * inv, t2 = (mx - o) * inv;             // :128-129
      int f1 = 2 * a, f2 = 2 * a + 1;
      if (t1 > t2) { const float tt = t1; t1 = t2; t2 = tt; f1 = 2 * a + 1; f2 = 2 * a; }   // :130-131
      if (t1 > tmin) { tmin = t1; fmin = f1; }                    // :132
      if (t2 < tmax) { tmax = t2; fmax = f2; }                    // :133
    }
  };
  axis(ox, dx, ix, lo.x, hi.x, 0);
  axis(oy, dy, iy, lo.y, hi.y, 1);
  axis(oz, dz, iz, lo.z, hi.z, 2);
  // (:134-135 tests t_min > t_max after every axis; t_min only grows and t_max only shrinks, so
  // testing once at the end decides the same)
  if (!ok || tmin > tmax) return false;
  t = tmin > 0.0f ? tmin : tmax;                                  // :137
  face = tmin > 0.0f ? fmin : fmax;
  return t != 0.0f;
}

// Hit point on a cube (:168): o + d t, with the coordinate of the plane that was hit set to the plane
// itself.  In the reference's FP64 that coordinate is the plane's to ~1e-16 and cube_normal's
// `abs(...) < 1e-5` (:144-149) recognises it; FP32's o + d t can be off by more than 1e-5 when the
// origin is far away, so the plane is taken directly.
__device__ __forceinline__ void snap_to_face(const float4 lo, const float4 hi, int face, float& px, float& py,
                                             float& pz) {
  const float4 s = (face & 1) ? hi : lo;
  const int a = face >> 1;
  if (a == 0) px = s.x; else if (a == 1) py = s.y; else pz = s.z;
}

// cube_normal (:142-150): the first face, in the reference's order, within 1e-5 of the point
__device__ __forceinline__ void cube_normal(const float4 lo, const float4 hi, float px, float py, float pz,
                                            float& nx, float& ny, float& nz) {
  const float eps = 1e-5f;
  nx = 0.f; ny = 1.f; nz = 0.f;                                   // :150
  if (fabsf(px - lo.x) < eps) { nx = -1.f; ny = 0.f; }            // :144
  else if (fabsf(px - hi.x) < eps) { nx = 1.f; ny = 0.f; }        // :145
  else if (fabsf(py - lo.y) < eps) { ny = -1.f; }                 // :146
  else if (fabsf(py - hi.y) < eps) { ny = 1.f; }                  // :147
  else if (fabsf(pz - lo.z) < eps) { ny = 0.f; nz = -1.f; }       // :148
  else if (fabsf(pz - hi.z) < eps) { ny = 0.f; nz = 1.f; }        // :149
}

// find_nearest_intersection (:152-172) over spheres AND cubes.  `sg` holds the n spheres followed by
// the cubes (two entries each).  The spheres are scanned first: a cube listed before a sphere in the
// reference's list wins an exact tie in t there and loses it here (ties between a sphere and a cube
// have measure zero; include/rt355.h, rt_set_cubes).
template <bool CUBES>
__device__ __forceinline__ void nearest_object(const float4* __restrict__ sg, const float4* __restrict__ gg, int n,
                                               int nc, float ox, float oy, float oz, float dx, float dy, float dz,
                                               float& best_t, int& best_i, int& best_face) {
  nearest_hit(sg, gg, n, ox, oy, oz, dx, dy, dz, best_t, best_i);
  best_face = 0;
  if constexpr (CUBES) {
    const float ix = frcp(dx), iy = frcp(dy), iz = frcp(dz);
    for (int c = 0; c < nc; ++c) {
      float t;
      int face;
      if (cube_t(sg[n + 2 * c], sg[n + 2 * c + 1], ox, oy, oz, dx, dy, dz, ix, iy, iz, t, face) && t < best_t) {
        best_t = t; best_i = n + c; best_face = face;             // :162-163
      }
    }
  }
}

// find_nearest_intersection for reflection rays of sphere-only opaque scenes with more than a handful of spheres:
// a ray that leaves object k (origin within 0.001 of its surface) and hits sphere j points within
// asin((r_k + r_j) / |J - K|) of the direction K -> J, whatever its origin — so which spheres a ray CAN hit
// follows from (k, direction) alone.  Directions are bucketed on a cube map (6 faces x dim x dim); the host lists,
// per object and bucket, the spheres in reach as a bit mask (build_reflect_table, FP64, margins far above FP32
// error: a superset).  A lane fetches ITS mask (one 8-byte load per 64 spheres) and walks its bits in ascending
// order, testing each with the expressions of nearest_hit() — same distances, same lowest-index rule on ties, so the
// result is the plain scan's (RT_OPT_REFLECT_TABLE off, test).  Only ~5 % of reflection rays hit anything and a
// mask holds 0.5 spheres on average at 64 spheres: the wave loops max-over-lanes(popcount) ~ 4 times instead of 64.
// cube-map bucket of a direction: major axis m (face 2 m + sign), the other two components over |major|
__device__ __forceinline__ int direction_bucket(float dx, float dy, float dz, int B) {
  const float ax = fabsf(dx), ay = fabsf(dy), az = fabsf(dz);
  const bool mx = ax >= ay && ax >= az, my = !mx && ay >= az;
  const float major = mx ? dx : my ? dy : dz;
  const float a = mx ? dy : my ? dz : dx, b = mx ? dz : my ? dx : dy;
  const int m = mx ? 0 : my ? 1 : 2;
  const float inv = frcp(fabsf(major));
  const float hB = 0.5f * float(B);
  const int ix = min(B - 1, max(0, int(fmaf(a * inv, hB, hB))));
  const int iy = min(B - 1, max(0, int(fmaf(b * inv, hB, hB))));
  return ((2 * m + (major < 0.0f ? 1 : 0)) * B + iy) * B + ix;
}
__device__ __forceinline__ void nearest_table(const RenderParams& p, const float4* __restrict__ sg, int obj, float ox,
                                              float oy, float oz, float dx, float dy, float dz, float& best_t,
                                              int& best_i) {
  const float inv_a = frcp(fmaf(dz, dz, fmaf(dy, dy, dx * dx)));
  best_t = INFINITY;
  best_i = -1;
  const int B = p.refl_dim;
  const int bucket = direction_bucket(dx, dy, dz, B);
  const uint2* row = p.refl_tab + (size_t(obj) * size_t(6 * B * B) + size_t(bucket)) * size_t(p.refl_pairs);
  for (int w = 0; w < p.refl_pairs; ++w) {
    const uint2 mask = row[w];
    uint32_t lo = mask.x, hi = mask.y;
    while (__ballot((lo | hi) != 0u) != 0ull) {
      if ((lo | hi) != 0u) {
        const bool in_lo = lo != 0u;
        uint32_t cur = in_lo ? lo : hi;
        const int j = 64 * w + (in_lo ? 0 : 32) + __ffs(int(cur)) - 1;
        cur &= cur - 1u;
        if (in_lo) lo = cur; else hi = cur;
        float dl;
        const float t = sphere_t_exact(sg[j], ox, oy, oz, dx, dy, dz, inv_a, dl);
        if (dl >= 0.0f && t > 0.0f && t < best_t) { best_t = t; best_i = j; }
      }
    }
  }
}

// The same for scenes with 256 spheres or more, where a mask per entry would be mostly zeros: an entry is a short
// list of sphere indices, stored in chunks of four (one 8-byte load; the next chunk is requested before this one is
// tested): two dependent loads per ray.
__device__ __forceinline__ void nearest_list(const RenderParams& p, const float4* __restrict__ sg, int obj, float ox,
                                             float oy, float oz, float dx, float dy, float dz, float& best_t,
                                             int& best_i) {
  const float inv_a = frcp(fmaf(dz, dz, fmaf(dy, dy, dx * dx)));
  best_t = INFINITY;
  best_i = -1;
  const int B = p.list_dim;
  const uint32_t* st = p.list_start + size_t(obj) * size_t(6 * B * B) + size_t(direction_bucket(dx, dy, dz, B));
  uint32_t c = st[0];
  const uint32_t e = st[1];
  uint2 nxt = make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
  if (c < e) nxt = p.list_items[c];
  auto test = [&](uint32_t j) {
    if (j != 0xFFFFu) {
      float dl;
      const float t = sphere_t_exact(sg[j], ox, oy, oz, dx, dy, dz, inv_a, dl);
      if (dl >= 0.0f && t > 0.0f && t < best_t) { best_t = t; best_i = int(j); }
    }
  };
  while (__ballot(c < e) != 0ull) {
    if (c < e) {
      const uint2 cur = nxt;
      ++c;
      if (c < e) nxt = p.list_items[c];
      test(cur.x & 0xFFFFu); test(cur.x >> 16); test(cur.y & 0xFFFFu); test(cur.y >> 16);
    }
  }
}

// point = ray.origin + ray.direction * t (:165 / :168) of object idx
template <bool CUBES>
__device__ __forceinline__ void hit_point(const float4* __restrict__ sg, int n, int idx, int face, float ox,
                                          float oy, float oz, float dx, float dy, float dz, float t, float& px,
                                          float& py, float& pz) {
  px = fmaf(dx, t, ox); py = fmaf(dy, t, oy); pz = fmaf(dz, t, oz);
  if constexpr (CUBES)
    if (idx >= n) snap_to_face(sg[n + 2 * (idx - n)], sg[n + 2 * (idx - n) + 1], face, px, py, pz);
}

// sphere_normal (:139-140) or cube_normal (:142-150) of object idx at P
template <bool CUBES>
__device__ __forceinline__ void surface_normal(const float4* __restrict__ sg, int n, int idx, float px, float py,
                                               float pz, float& nx, float& ny, float& nz) {
  if (CUBES && idx >= n) {
    cube_normal(sg[n + 2 * (idx - n)], sg[n + 2 * (idx - n) + 1], px, py, pz, nx, ny, nz);
  } else {
    const float4 s = sg[idx];
    nx = px - s.x; ny = py - s.y; nz = pz - s.z;
    const float inv = frsq(fmaf(nz, nz, fmaf(ny, ny, nx * nx)));
    nx *= inv; ny *= inv; nz *= inv;
  }
}

#if defined(__HIP_DEVICE_COMPILE__)
typedef const FrameRec __attribute__((address_space(4))) * cframe_t;
__device__ __forceinline__ cframe_t as_const(const FrameRec* p) { return (cframe_t)(unsigned long long)(p); }
#else
typedef const FrameRec* cframe_t;
__device__ __forceinline__ cframe_t as_const(const FrameRec* p) { return p; }
#endif
#if defined(__HIP_DEVICE_COMPILE__)
typedef const uint32_t __attribute__((address_space(4))) * cocc_t;
__device__ __forceinline__ cocc_t as_const(const uint32_t* p) { return (cocc_t)(unsigned long long)(p); }
#else
typedef const uint32_t* cocc_t;
__device__ __forceinline__ cocc_t as_const(const uint32_t* p) { return p; }
#endif

// table entry: four spheres, struct-of-arrays, so one s_load_dwordx16 brings a group and packed
// VALU ops find sphere pairs in adjacent SGPRs.  w holds -c (or -inf where the camera is inside).
struct __attribute__((aligned(64))) CamGroup {
  float x[4], y[4], z[4], negc[4];
};
#if defined(__HIP_DEVICE_COMPILE__)
typedef const CamGroup __attribute__((address_space(4))) * camtab_t;
__device__ __forceinline__ camtab_t as_const(const CamGroup* p) { return (camtab_t)(unsigned long long)(p); }
#else
typedef const CamGroup* camtab_t;
__device__ __forceinline__ camtab_t as_const(const CamGroup* p) { return p; }
#endif

// groups of a frame's table: the sphere groups rounded up to an even count (the scan below is
// unrolled by two) plus one never-hit group for its last, unused, prefetch
__host__ __device__ __forceinline__ int cam_groups_scanned(int n) { return (((n + 3) >> 2) + 1) & ~1; }
__host__ __device__ __forceinline__ int cam_groups_stride(int n) { return cam_groups_scanned(n) + 1; }

// NP rays per lane against the same table entries: the scalar loads, the loop and the candidate
// branch are shared (the visibility pass traces several pixels per lane: it is issue-bound, and scalar
// instructions are ~40 % of what a one-pixel wave issues).
template <int NP>
__device__ __forceinline__ void test_cam_group(const CamGroup& grp, int g, cgeom_t geom, float ox, float oy, float oz,
                                               const float (&dx)[NP], const float (&dy)[NP], const float (&dz)[NP],
                                               const float (&inv_a)[NP], float (&best_t)[NP], int (&best_i)[NP]) {
  float nb[NP][4], disc[NP][4];
  uint32_t any = 0xFFFFFFFFu;
#pragma unroll
  for (int q = 0; q < NP; ++q)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      nb[q][j] = fmaf(grp.z[j], dz[q], fmaf(grp.y[j], dy[q], grp.x[j] * dx[q]));
      disc[q][j] = fmaf(nb[q][j], nb[q][j], grp.negc[j]);
      any &= __float_as_uint(disc[q][j]) | __float_as_uint(nb[q][j]);
    }
  if (int(any) >= 0) {
    // some lane may hit one of these four spheres: decide and measure with the cancellation-free
    // form of nearest_hit (the table's c carries a safety margin, so this is a superset).  One wave-uniform
    // branch and one scalar load per SPHERE that has a candidate in some lane and pixel (all four requested
    // at once); its NP pixels are then tested without further branches — a branch, a load and a wait per
    // (sphere, pixel) pair made blocks full of hits cost several times what their arithmetic does.
    const float4 s4[4] = {geom[4 * g], geom[4 * g + 1], geom[4 * g + 2], geom[4 * g + 3]};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      uint32_t cj = 0xFFFFFFFFu;
#pragma unroll
      for (int q = 0; q < NP; ++q) cj &= __float_as_uint(disc[q][j]) | __float_as_uint(nb[q][j]);
      if (__ballot(int(cj) >= 0) != 0ull) {
#pragma unroll
        for (int q = 0; q < NP; ++q) {
          float dl;
          const float t = sphere_t_exact(s4[j], ox, oy, oz, dx[q], dy[q], dz[q], inv_a[q], dl);
          // (`<` between spheres of one group and of groups in index order; the culled scan below may
          // visit groups in the same increasing order only, so ties still go to the lowest index)
          if (disc[q][j] >= 0.0f && nb[q][j] > 0.0f && dl >= 0.0f && t > 0.0f && t < best_t[q]) {
            best_t[q] = t;
            best_i[q] = 4 * g + j;
          }
        }
      }
    }
  }
}

template <int NP>
__device__ __forceinline__ void primary_hit_n(const float4* __restrict__ gg, const CamGroup* __restrict__ camtab,
                                              int n, float ox, float oy, float oz, const float (&dx)[NP],
                                              const float (&dy)[NP], const float (&dz)[NP], float (&best_t)[NP],
                                              int (&best_i)[NP]) {
  float inv_a[NP];
#pragma unroll
  for (int q = 0; q < NP; ++q) {
    best_t[q] = INFINITY;
    best_i[q] = -1;
    inv_a[q] = frcp(fmaf(dz[q], dz[q], fmaf(dy[q], dy[q], dx[q] * dx[q])));
  }
  const int ngroups = cam_groups_scanned(n);   // padded with never-hit entries
  camtab_t tab = as_const(camtab);
  cgeom_t geom = as_const(gg);   // candidates are rare: their sphere comes by scalar load
  // Two groups per turn in two SGPR sets (no register-to-register rotation); each group's scalar
  // load is issued one group ahead of its arithmetic and the (rare) candidate branch.
  CamGroup a = tab[0];
  for (int g = 0; g < ngroups; g += 2) {
    const CamGroup b = tab[g + 1];
    test_cam_group<NP>(a, g, geom, ox, oy, oz, dx, dy, dz, inv_a, best_t, best_i);
    a = tab[g + 2];
    test_cam_group<NP>(b, g + 1, geom, ox, oy, oz, dx, dy, dz, inv_a, best_t, best_i);
  }
}

// Bounding cone of a bundle of primary rays: unit axis, sin and cos^2 of the half-angle (with margin).
struct RayCone {
  float ax, ay, az, sin_t, cos2_t;
};

// The same scan for many spheres: the rays of a wave are a narrow bundle (a block of 8 x 8 NP pixels),
// so first the WAVE culls — lane l tests sphere base + l against the bundle's bounding cone (the test
// of lighting()'s per-lane cull: the sphere reaches into the cone iff perp cos(t) - along sin(t) <= r,
// with the same slack) — and only table groups that hold a surviving sphere are tested per ray.
// ~25 VALU per 64 spheres instead of 16 NP per 4.  Groups are still visited in increasing order.
// Does sphere s reach into the cone?  (the wave's cull: one sphere per lane)
__device__ __forceinline__ bool sphere_in_cone(const float4 s, const RayCone& cone, float ox, float oy, float oz) {
  const float cx = s.x - ox, cy = s.y - oy, cz = s.z - oz;
  const float d2 = fmaf(cz, cz, fmaf(cy, cy, cx * cx)), r = fsqrt(s.w);
  const float along = fmaf(cz, cone.az, fmaf(cy, cone.ay, cx * cone.ax));
  const float perp2 = fmaxf(0.0f, fmaf(-along, along, d2));
  const float rhs = fmaf(along, cone.sin_t, fmaf(r, 1.00390625f, fmaf(d2, 1.0f / 65536.0f, 1.0f / 1048576.0f)));
  return rhs >= 0.0f && rhs * rhs >= perp2 * cone.cos2_t;
}
// Can a ray of the cone "hit" the box?  intersect_cube accepts a box that the ray's LINE crosses, also behind the origin
// (cube_t: t < 0 counts), so the box is tested against the cone and its mirror image, through its bounding sphere.
__device__ __forceinline__ bool box_in_double_cone(const float4 lo, const float4 hi, const RayCone& cone, float ox, float oy,
                                                   float oz) {
  const float hx = 0.5f * (hi.x - lo.x), hy = 0.5f * (hi.y - lo.y), hz = 0.5f * (hi.z - lo.z);
  const float4 s = make_float4(lo.x + hx, lo.y + hy, lo.z + hz, fmaf(hz, hz, fmaf(hy, hy, hx * hx)) * 1.0001f);
  const RayCone back{-cone.ax, -cone.ay, -cone.az, cone.sin_t, cone.cos2_t};
  return sphere_in_cone(s, cone, ox, oy, oz) || sphere_in_cone(s, back, ox, oy, oz);
}
// (`first` = the ballot of sphere_in_cone over spheres 0..63, which the caller has taken already; `kept`: the caller has
// all the ballots — lane c of keep_lo / keep_hi holds that of spheres 64 c .. 64 c + 63)
__device__ __forceinline__ unsigned long long mask_of_chunk(uint32_t keep_lo, uint32_t keep_hi, int c) {
  return (unsigned long long)uint32_t(__builtin_amdgcn_readlane(int(keep_lo), c)) |
         ((unsigned long long)uint32_t(__builtin_amdgcn_readlane(int(keep_hi), c)) << 32);
}
template <int NP>
__device__ __forceinline__ void primary_hit_culled(const float4* __restrict__ gg, const CamGroup* __restrict__ camtab,
                                                   int n, int lane, const RayCone& cone, unsigned long long first,
                                                   bool kept, uint32_t keep_lo, uint32_t keep_hi,
                                                   float ox, float oy, float oz, const float (&dx)[NP],
                                                   const float (&dy)[NP], const float (&dz)[NP], float (&best_t)[NP],
                                                   int (&best_i)[NP]) {
  float inv_a[NP];
#pragma unroll
  for (int q = 0; q < NP; ++q) {
    best_t[q] = INFINITY;
    best_i[q] = -1;
    inv_a[q] = frcp(fmaf(dz[q], dz[q], fmaf(dy[q], dy[q], dx[q] * dx[q])));
  }
  camtab_t tab = as_const(camtab);
  cgeom_t geom = as_const(gg);
  for (int base = 0; base < n; base += 64) {
    unsigned long long m = first;
    if (kept) {
      m = mask_of_chunk(keep_lo, keep_hi, base >> 6);
    } else if (base != 0) {
      const int i = base + lane;
      m = __ballot(i < n && sphere_in_cone(gg[i < n ? i : n - 1], cone, ox, oy, oz));
    }
    m |= m >> 1; m |= m >> 2;
    m &= 0x1111111111111111ull;              // bit 4 k: group k of this chunk holds a sphere to test
    while (m != 0ull) {
      const int g = (base >> 2) + ((__ffsll((long long)m) - 1) >> 2);
      m &= m - 1ull;
      const CamGroup grp = tab[g];
      test_cam_group<NP>(grp, g, geom, ox, oy, oz, dx, dy, dz, inv_a, best_t, best_i);
    }
  }
}

__device__ __forceinline__ void primary_hit(const float4* __restrict__ gg, const CamGroup* __restrict__ camtab,
                                            int n, float ox, float oy, float oz, float dx, float dy,
                                            float dz, float& best_t, int& best_i) {
  const float ax[1] = {dx}, ay[1] = {dy}, az[1] = {dz};
  float t[1];
  int i[1];
  primary_hit_n<1>(gg, camtab, n, ox, oy, oz, ax, ay, az, t, i);
  best_t = t[0];
  best_i = i[0];
}

// Per-frame table for primary_hit (one tiny launch per batch, ahead of the render kernel on the
// same stream): per sphere C - O and -(c - margin), c = |C - O|^2 - r^2 in FP64.  The margin
// (2^-17 |C - O|^2, ~30x the FP32 error of nb^2 - c) makes the cheap test a superset of the exact
// one; spheres that contain the camera can never be hit from it (near root < 0, :102) -> -inf.
__global__ __launch_bounds__(256) void rt_store_frames_kernel(const FrameChunk chunk, FrameRec* __restrict__ dst, int n) {
  const int i = threadIdx.x;
  if (i < n) {
    FrameRec r;
    r.cam = chunk.cam[i];
    r.seed = chunk.seed[i];
    r.pad[0] = r.pad[1] = r.pad[2] = 0u;
    dst[i] = r;
  }
}

__global__ __launch_bounds__(256) void rt_camera_table_kernel(const float4* __restrict__ geom, int n,
                                                              const FrameRec* __restrict__ frames,
                                                              CamGroup* __restrict__ camtab, int n_frames) {
  const int n4 = cam_groups_stride(n) * 4;   // every entry of the padded table is written
  const int frame = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4 || frame >= n_frames) return;
  float ex = 0.f, ey = 0.f, ez = 0.f, ew = -INFINITY;
  if (i < n) {
    const float4 g = geom[i];
    const float* origin = frames[frame].cam.origin;
    const double cx = double(g.x) - double(origin[0]), cy = double(g.y) - double(origin[1]),
                 cz = double(g.z) - double(origin[2]);
    const double d2 = cx * cx + cy * cy + cz * cz;
    const double c = d2 - double(g.w);
    ex = float(cx); ey = float(cy); ez = float(cz);
    ew = c > 0.0 ? -float(c - d2 * (1.0 / 131072.0)) : -INFINITY;
  }
  CamGroup& grp = camtab[size_t(frame) * (n4 >> 2) + (i >> 2)];
  grp.x[i & 3] = ex; grp.y[i & 3] = ey; grp.z[i & 3] = ez; grp.negc[i & 3] = ew;
}

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

template <int MODE, int B, bool REFR = false, bool CUBES = false, bool LIST = false>
__device__ __forceinline__ LightSums lighting(const DevLights& L, const float4* __restrict__ sg,
                                         const float4* __restrict__ gg, const float2* __restrict__ refr,
                                         const uint32_t* __restrict__ glass_bits,
                                         const uint32_t* __restrict__ occl_row, const uint2* __restrict__ occl_items, int n, int nc, float px, float py, float pz, float nx, float ny, float nz,
                                         float dx, float dy, float dz, float sx, float sy, float sz,
                                         uint32_t level_key) {
#if RT_CULL_STATS
  unsigned dbg_rel_ = 0, dbg_lanes_ = 0, dbg_iters_ = 0, dbg_turns_ = 0;
#endif
  const int nrays = 1 + L.ns;
  const float vdn = -(fmaf(dz, nz, fmaf(dy, ny, dx * nx)));  // view . normal, view = -D (:257)
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
  constexpr bool SHARE = REFR || B < kShadowBatch;     // (the reference's 13 rays in one batch: nothing to share, no registers held)
  [[maybe_unused]] uint32_t kept0 = 0u, kept1 = 0u, keptc = 0u;
  for (int base = 0; base < nrays; base += B) {
    float Lx[B], Ly[B], Lz[B], m[B];
    float mrf[REFR ? B : 1];   // REFR: the same verdicts for the REFRACTIVE spheres (m: the opaque ones)
#pragma unroll
    for (int k = 0; k < (REFR ? B : 1); ++k) mrf[k] = -1.0f;
#pragma unroll
    for (int k = 0; k < B; ++k) {
      const int j = base + k;
      float lx, ly, lz;
      if (j == 0) {  // scene['global_light'] (:198)
        lx = L.ppos[0] - px; ly = L.ppos[1] - py; lz = L.ppos[2] - pz;
      } else {       // area-light sample j-1 (:180-183)
        float ru1, rv1;
        draw_pair1<MODE>(level_key, uint32_t(j - 1), ru1, rv1);
        lx = fmaf(L.av[0], rv1, fmaf(L.au[0], ru1, cax));
        ly = fmaf(L.av[1], rv1, fmaf(L.au[1], ru1, cay));
        lz = fmaf(L.av[2], rv1, fmaf(L.au[2], ru1, caz));
      }
      // a ray beyond the light's sample count gets L = 0: it can never be shadowed and every
      // weight below (N.L, V.R) is then 0 too
      const float inv = (j < nrays) ? frsq(fmaf(lz, lz, fmaf(ly, ly, lx * lx))) : 0.0f;
      Lx[k] = lx * inv; Ly[k] = ly * inv; Lz[k] = lz * inv;
      m[k] = -1.0f;   // sign bit set = "no sphere hit yet" in both accumulation forms
    }
    // One sphere against the B rays.  Verdict accumulation: with RT_ACC_BITS the sign bits do the
    // work at full VALU rate — hit <=> sign(dl) == 0 and sign(nb) == 0, so `acc &= (dl | nb)`
    // clears acc's sign bit as soon as any sphere is hit (v_min/v_max/v_cmp are half rate on
    // gfx950, v_and/v_or full rate: profiles/r1_ubench_valu_rate.txt).
    // (REFR: `glass` = 0x80000000 for a refractive sphere, 0 for an opaque one — a set sign bit leaves the other
    // class's verdicts as they are)
    auto test_sphere = [&](const float4 g, uint32_t glass = 0u) {
      const float cx = g.x - sx, cy = g.y - sy, cz = g.z - sz;
      const float c = fmaf(cz, cz, fmaf(cy, cy, fmaf(cx, cx, -g.w)));
      // Two-tier test.  Tier 1 is the plain discriminant nb^2 - c (4 ops per ray): its FP32 error is
      // below 2^-20 |co|^2 (3 roundings in nb, 3 in c, |L|^2 = 1 +- 4 ulp), so with thr = 2^-18 |co|^2
      //   dh = nb^2 - (c + thr) >= 0  is a certain hit, and  dh + 2 thr < 0  a certain miss.
      // Only a verdict inside that band (a ray within ~1 % of a silhouette) needs tier 2, the
      // cancellation-free r^2 - |l|^2 form, and it is evaluated for the lanes that saw such a ray
      // (~1 % of sphere iterations).  Certain hits re-evaluated by tier 2 stay hits.
      const float thr = (c + g.w) * (1.0f / 262144.0f);
      const float chi = (c > 0.0f) ? c + thr : INFINITY;   // origin inside the sphere: never hit
      const float thr2 = thr + thr;
      uint32_t amb = 0u;
#pragma unroll
      for (int k = 0; k < B; ++k) {
        const float nb = fmaf(cz, Lz[k], fmaf(cy, Ly[k], cx * Lx[k]));
        const float dh = fmaf(nb, nb, -chi);
        const float dlo = dh + thr2;
        const uint32_t verdict = __float_as_uint(dh) | __float_as_uint(nb);
        m[k] = __uint_as_float(__float_as_uint(m[k]) & (REFR ? verdict | glass : verdict));
        if constexpr (REFR) mrf[k] = __uint_as_float(__float_as_uint(mrf[k]) & (verdict | (glass ^ 0x80000000u)));
        amb |= __float_as_uint(dh) & ~__float_as_uint(dlo);   // sign set: dh < 0 <= dh + 2 thr
        asm volatile("" : "+v"(m[k]), "+v"(amb));   // keep both chains sequential (no reassociation tree)
      }
      if (int(amb) < 0)
      {
        const float r2 = (c > 0.0f) ? g.w : -INFINITY;
#pragma unroll
        for (int k = 0; k < B; ++k) {
          const float nb = fmaf(cz, Lz[k], fmaf(cy, Ly[k], cx * Lx[k]));
          const float lx = fmaf(-nb, Lx[k], cx), ly = fmaf(-nb, Ly[k], cy), lz = fmaf(-nb, Lz[k], cz);
          const float dl = fmaf(-lz, lz, fmaf(-ly, ly, fmaf(-lx, lx, r2)));
          const uint32_t verdict = __float_as_uint(dl) | __float_as_uint(nb);
          m[k] = __uint_as_float(__float_as_uint(m[k]) & (REFR ? verdict | glass : verdict));
          if constexpr (REFR) mrf[k] = __uint_as_float(__float_as_uint(mrf[k]) & (verdict | (glass ^ 0x80000000u)));
          asm volatile("" : "+v"(m[k]));   // keep the and-chain sequential (no reassociation tree)
        }
      }
    };
    // Per-lane sphere culling.  The B rays of a lane leave one point towards one small light, so
    // almost every sphere is far from all of them; but lanes of a wave are unrelated pixels, so no
    // sphere can be skipped for the whole wave.  Instead each lane first collects, 32 spheres at
    // a time, a bit mask of the spheres that can matter to it (~20 ops per sphere instead of
    // 7 B + 11) and then the wave loops max-over-lanes(popcount) times, every lane testing ITS next
    // candidate, fetched from LDS by per-lane index.  A sphere can matter to a lane if
    //   * the shadow origin is outside it (c > 0; inside: the near root is negative, :102), and
    //   * it reaches into the cone {apex S, axis towards the light's centre, half-angle
    //     asin(R / |A - P|)} that holds every area-light ray (R bounds the light's half extent; the
    //     test  perp cos(t) - along sin(t) <= r  carries ~0.4 % of slack, far above FP32 error), or
    //   * (first batch) the point-light ray's plain discriminant says "hit or too close to call" —
    //     exactly tier 1's own condition for touching the verdict of ray 0.
    // A culled sphere would have left every verdict as it was, so results are unchanged.
    {
      const float ax0 = L.apos[0] - px, ay0 = L.apos[1] - py, az0 = L.apos[2] - pz;
      const float dA2 = fmaf(az0, az0, fmaf(ay0, ay0, ax0 * ax0));
      const float inv_dA = frsq(dA2);
      const float axx = ax0 * inv_dA, axy = ay0 * inv_dA, axz = az0 * inv_dA;
      const float sin_t = fminf(1.0f, L.arad * inv_dA * 1.00390625f);
      const float cos2_t = fmaf(-sin_t, sin_t, 1.0f);
      const float sin2_t = sin_t * sin_t * 1.0078125f;
      const bool wide = !(sin_t < 0.99f);   // P within (about) the light's bounding sphere: rays go anywhere
      // one candidate sphere against this lane's cone (and its point-light ray): the verdict's sign bit is CLEAR if the
      // sphere can matter
      auto cone_cull = [&](const float4 g) -> uint32_t {
        const float cx = g.x - sx, cy = g.y - sy, cz = g.z - sz;
        const float c = fmaf(cz, cz, fmaf(cy, cy, fmaf(cx, cx, -g.w)));
        const float d2 = c + g.w, r = fsqrt(g.w);
        const float along = fmaf(cz, axz, fmaf(cy, axy, cx * axx));
        const float al2 = along * along, perp2 = d2 - al2;
        const float rhs = fmaf(along, sin_t, fmaf(r, 1.00390625f, fmaf(d2, 1.0f / 65536.0f, 1.0f / 1048576.0f)));
        const float t2 = fmaf(rhs, rhs, -(perp2 * cos2_t));
        const float e = fmaf(al2, cos2_t, -(perp2 * sin2_t));
        uint32_t no = __float_as_uint(rhs) | __float_as_uint(t2) | (__float_as_uint(along) & ~__float_as_uint(e));
        if (wide) no = 0u;
        if (base == 0) {   // the point-light ray (k = 0): nb > 0 and nb^2 - c + 1.5 thr >= 0
          const float nb = fmaf(cz, Lz[0], fmaf(cy, Ly[0], cx * Lx[0]));
          const float q = fmaf(nb, nb, fmaf(d2, 1.5f / 262144.0f, -c));
          no &= __float_as_uint(nb) | __float_as_uint(q);
        }
        return no | __float_as_uint(c);      // origin inside the sphere: never hit
      };
      if constexpr (LIST) {
        // Many spheres: the object's candidates come as a list (chunks of four indices, the next chunk requested
        // before this one is looked at) instead of 32 .. 64 mostly empty mask words, each a load of its own.
        // Up to 32 candidates are culled at a time; the survivors' positions go into `rel` and are tested below.
        uint32_t c = occl_row[0];
        const uint32_t e = occl_row[1];
        while (__ballot(c < e) != 0ull) {
          const uint32_t c0 = c;
          uint32_t rel = 0u;
          uint2 nxt = make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
          if (c < e) nxt = occl_items[c];
#pragma nounroll
          for (int grp = 0; grp < 8; ++grp) {
            if (__ballot(c < e) == 0ull) break;
            if (c < e) {
              const uint2 cur = nxt;
              ++c;
              if (c < e) nxt = occl_items[c];
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                const uint32_t j = ((i & 2) ? cur.y : cur.x) >> ((i & 1) * 16) & 0xFFFFu;
                if (j != 0xFFFFu) rel |= (~cone_cull(sg[j]) >> 31) << (4 * grp + i);
              }
            }
          }
          while (__ballot(rel != 0u) != 0ull) {
            if (rel != 0u) {
              const int pos = __ffs(int(rel)) - 1;
              rel &= rel - 1u;
              const uint2 ch = occl_items[c0 + uint32_t(pos >> 2)];
              const uint32_t j = ((pos & 2) ? ch.y : ch.x) >> ((pos & 1) * 16) & 0xFFFFu;
              test_sphere(sg[j]);
            }
          }
        }
      } else
      for (int cbase = 0, word = 0; cbase < n; cbase += 32, ++word) {
        // candidates: the spheres that can shadow SOME point of the object this lane shades (a table
        // built at scene upload: every ray from the object towards a light lies in one cone, and few
        // spheres reach into it).  Lanes shade different objects, so each walks its own bit list.
        uint32_t cand = occl_row[word];
        uint32_t rel = 0u;
        if (SHARE && base > 0 && n <= 64) { cand = 0u; rel = word == 0 ? kept0 : kept1; }   // (the first batch's survivors)
        while (__ballot(cand != 0u) != 0ull) {
#if RT_CULL_STATS
          g_dbg_turns += 1;
#endif
          if (cand != 0u) {
            const int j = __ffs(int(cand)) - 1;
            cand &= cand - 1u;
            const float4 g = sg[cbase + j];   // per-lane LDS read
            const float cx = g.x - sx, cy = g.y - sy, cz = g.z - sz;
            const float c = fmaf(cz, cz, fmaf(cy, cy, fmaf(cx, cx, -g.w)));
            const float d2 = c + g.w, r = fsqrt(g.w);
            const float along = fmaf(cz, axz, fmaf(cy, axy, cx * axx));
            const float al2 = along * along, perp2 = d2 - al2;
            // sign bits do the logic (full-rate bit ops instead of half-rate compares): a set sign = "no"
            //   rhs < 0 or rhs^2 < perp^2 cos^2 : out of the cone's reach
            //   along < 0 and along^2 cos^2 > perp^2 sin^2 : every ray of the cone points away (nb < 0)
            const float rhs = fmaf(along, sin_t, fmaf(r, 1.00390625f, fmaf(d2, 1.0f / 65536.0f, 1.0f / 1048576.0f)));
            const float t2 = fmaf(rhs, rhs, -(perp2 * cos2_t));
            const float e = fmaf(al2, cos2_t, -(perp2 * sin2_t));
            uint32_t no = __float_as_uint(rhs) | __float_as_uint(t2) | (__float_as_uint(along) & ~__float_as_uint(e));
            if (wide) no = 0u;
            if (base == 0) {   // the point-light ray (k = 0): nb > 0 and nb^2 - c + 1.5 thr >= 0
              const float nb = fmaf(cz, Lz[0], fmaf(cy, Ly[0], cx * Lx[0]));
              const float q = fmaf(nb, nb, fmaf(d2, 1.5f / 262144.0f, -c));   // tier 1 acts on nb^2 - c + 2^-18 d2 >= 0
              no &= __float_as_uint(nb) | __float_as_uint(q);
            }
            no |= __float_as_uint(c);           // origin inside the sphere: never hit
            rel |= (~no >> 31) << j;
          }
        }
#if RT_CULL_STATS
        g_dbg_rel += __popc(rel); g_dbg_lanes += 1;
#endif
        if constexpr (SHARE) { if (base == 0) { if (word == 0) kept0 = rel; else if (word == 1) kept1 = rel; } }
        while (__ballot(rel != 0u) != 0ull) {
#if RT_CULL_STATS
          g_dbg_iters += 1;
#endif
          if (rel != 0u) {
            const int j = __ffs(int(rel)) - 1;
            rel &= rel - 1u;
            if constexpr (REFR) test_sphere(sg[cbase + j], ((glass_bits[word] >> j) & 1u) << 31);
            else test_sphere(sg[cbase + j]);
          }
        }
      }
    }
    if constexpr (CUBES) {
      // any-hit over the cubes: the ray is blocked if intersect_cube gives a non-zero t of EITHER sign
      // (:137, :162 — see cube_t): the ray's LINE decides.  Per lane, the cubes whose bounding sphere reaches into the
      // cone of the area light's rays or its mirror image (apex S; the test of the sphere cull with |along|), or — first
      // batch — comes within reach of the point-light ray's line, are collected in a mask (32 cubes at a time); the
      // rays are then tested, ray by ray (three reciprocals per ray), against the lane's own candidates only: the wave
      // loops max-over-lanes(candidates) times per ray instead of once per cube.
      const float ax0 = L.apos[0] - px, ay0 = L.apos[1] - py, az0 = L.apos[2] - pz;
      const float inv_dA = frsq(fmaf(az0, az0, fmaf(ay0, ay0, ax0 * ax0)));
      const float axx = ax0 * inv_dA, axy = ay0 * inv_dA, axz = az0 * inv_dA;
      const float sin_t = fminf(1.0f, L.arad * inv_dA * 1.00390625f);
      const float cos2_t = fmaf(-sin_t, sin_t, 1.0f);
      const bool wide = !(sin_t < 0.99f);
      for (int c0 = 0; c0 < nc; c0 += 32) {
        uint32_t cm = 0u;
        const int cn = min(32, nc - c0);
        const bool again = SHARE && base > 0 && nc <= 32;      // (the first batch's candidate boxes)
        if (again) cm = keptc;
        for (int c = 0; c < (again ? 0 : cn); ++c) {
          const float4 lo = sg[n + 2 * (c0 + c)], hi = sg[n + 2 * (c0 + c) + 1];      // wave-uniform: LDS broadcast
          const float hx = 0.5f * (hi.x - lo.x), hy = 0.5f * (hi.y - lo.y), hz = 0.5f * (hi.z - lo.z);
          const float cx = (lo.x + hx) - sx, cy = (lo.y + hy) - sy, cz = (lo.z + hz) - sz;
          const float d2 = fmaf(cz, cz, fmaf(cy, cy, cx * cx));
          const float r = fsqrt(fmaf(hz, hz, fmaf(hy, hy, hx * hx)));
          const float reach = fmaf(r, 1.00390625f, fmaf(d2, 1.0f / 65536.0f, 1.0f / 1048576.0f));
          const float along = fmaf(cz, axz, fmaf(cy, axy, cx * axx));
          const float perp2 = fmaxf(0.0f, fmaf(-along, along, d2));
          const float rhs = fmaf(fabsf(along), sin_t, reach);
          bool cand = wide || rhs * rhs >= perp2 * cos2_t;
          if (base == 0) {                                         // the point-light ray: distance of its line from the centre
            const float al0 = fmaf(cz, Lz[0], fmaf(cy, Ly[0], cx * Lx[0]));
            cand = cand || fmaf(-al0, al0, d2) <= reach * reach;
          }
          cm |= (cand ? 1u : 0u) << c;
        }
#if !RT_CUBE_SHADOW_CULL
        cm = cn >= 32 ? ~0u : (1u << cn) - 1u;      // (diagnostic build: every box for every lane — tools/fuzz_cubes.py)
#endif
        if constexpr (SHARE) { if (c0 == 0 && base == 0) keptc = cm; }
        while (__ballot(cm != 0u) != 0ull) {
          if (cm != 0u) {
            const int c = c0 + __ffs(int(cm)) - 1;
            cm &= cm - 1u;
            const float4 lo = sg[n + 2 * c], hi = sg[n + 2 * c + 1];
            bool glass_cube = false;
            if constexpr (REFR) glass_cube = refr[n + c].x > 0.0f;
            (void)glass_cube;
            // A shadow ray is blocked by a box iff its LINE meets the box (cube_t accepts either sign of t; t = 0 exactly —
            // the origin in an exit plane — has measure zero).  Line against box by the three separating axes d x e_i:
            //   |(d x m)_i| <= h_j |d_k| + h_k |d_j|,  m = S - centre, h = half extents
            // (a zero component of d turns its two tests into the slab containment of :124-126) — 21 issue slots per ray
            // and box instead of 52 for the slab test with its three reciprocals.
            const float hx = 0.5f * (hi.x - lo.x), hy = 0.5f * (hi.y - lo.y), hz = 0.5f * (hi.z - lo.z);
            const float mx = sx - (lo.x + hx), my = sy - (lo.y + hy), mz = sz - (lo.z + hz);
#pragma unroll
            for (int k = 0; k < B; ++k) {
              const float lx = Lx[k], ly = Ly[k], lz = Lz[k];
              const float c0 = fmaf(ly, mz, -(lz * my)), c1 = fmaf(lz, mx, -(lx * mz)), c2 = fmaf(lx, my, -(ly * mx));
              const float b0 = fmaf(hy, fabsf(lz), hz * fabsf(ly)), b1 = fmaf(hz, fabsf(lx), hx * fabsf(lz)),
                          b2 = fmaf(hx, fabsf(ly), hy * fabsf(lx));
              if (fabsf(c0) <= b0 && fabsf(c1) <= b1 && fabsf(c2) <= b2) {
                if constexpr (REFR) {                              // (which kind of hit: see below)
                  if (glass_cube) mrf[k] = 0.0f; else m[k] = 0.0f;
                } else {
                  m[k] = 0.0f;
                }
              }
            }
          }
        }
      }
    }
    if constexpr (REFR) {
      // Scenes with refractive materials: a shadow ray is blocked only if its NEAREST hit is opaque
      // (raytracer.py:187, :202).  The scan above has told, per ray, whether it hits SOME opaque sphere (m) and
      // whether it hits SOME refractive one (mrf) — the any-hit tests of the opaque path, over the same culled
      // candidates, no root taken.  No opaque hit: lit.  Opaque hits only: blocked.  Both (a ray through glass
      // towards something solid, or the other way round: rare): the nearest hit decides, found by the full scan
      // of find_nearest_intersection for that ray.  Cubes (their line rule, :137, :162) have been entered into m and mrf above.
      // (sign bits as truth values, as in the scan: m / mrf have their sign CLEAR where an opaque / a refractive object was
      // hit; lit <=> no opaque hit or some refractive one <=> sign of m | ~mrf set: two operations per ray.  Whether any ray
      // of the lane was hit by objects of BOTH kinds — sign of m | mrf clear — is asked once per batch.)
      uint32_t both_any = 0u;
#pragma unroll
      for (int k = 0; k < B; ++k) {
        const uint32_t ms = __float_as_uint(m[k]), gs = __float_as_uint(mrf[k]);
        m[k] = __uint_as_float(ms | ~gs);                        // sign set: lit (what the weights below look at)
        mrf[k] = __uint_as_float(ms | gs);                       // sign clear: both kinds
        both_any |= ~(ms | gs);
        asm volatile("" : "+v"(both_any));
      }
      if (__ballot(int(both_any) < 0) != 0ull) {
#pragma unroll
        for (int k = 0; k < B; ++k) {
          if (int(__float_as_uint(mrf[k])) >= 0 && base + k < nrays) {
            float t = INFINITY;
            int hi = -1, face;
            if (n <= 64 && nc <= 32) {
              // (the expressions and the order of nearest_object(): the same distances, the lowest index on a tie, spheres
              // before boxes — over the point's surviving candidates instead of every object of the scene)
              const float inv_a = frcp(fmaf(Lz[k], Lz[k], fmaf(Ly[k], Ly[k], Lx[k] * Lx[k])));
              uint32_t lo = kept0, up = kept1;
              while ((lo | up) != 0u) {
                const bool in_lo = lo != 0u;
                uint32_t cur = in_lo ? lo : up;
                const int j = (in_lo ? 0 : 32) + __ffs(int(cur)) - 1;
                cur &= cur - 1u;
                if (in_lo) lo = cur; else up = cur;
                float dl;
                const float tt = sphere_t_exact(sg[j], sx, sy, sz, Lx[k], Ly[k], Lz[k], inv_a, dl);
                if (dl >= 0.0f && tt > 0.0f && tt < t) { t = tt; hi = j; }
              }
              if constexpr (CUBES) {
                const float ix = frcp(Lx[k]), iy = frcp(Ly[k]), iz = frcp(Lz[k]);
                uint32_t cmk = keptc;
                while (cmk != 0u) {
                  const int c = __ffs(int(cmk)) - 1;
                  cmk &= cmk - 1u;
                  float tc;
                  if (cube_t(sg[n + 2 * c], sg[n + 2 * c + 1], sx, sy, sz, Lx[k], Ly[k], Lz[k], ix, iy, iz, tc, face) && tc < t) { t = tc; hi = n + c; }
                }
              }
            } else {
              nearest_object<CUBES>(sg, gg, n, nc, sx, sy, sz, Lx[k], Ly[k], Lz[k], t, hi, face);
            }
            const bool blocked = hi >= 0 && !(refr[hi >= 0 ? hi : 0].x > 0.0f);
            m[k] = blocked ? 0.0f : -1.0f;
          }
        }
      }
    }
#pragma unroll
    for (int k = 0; k < B; ++k) {
      const float ndl = fmaf(nz, Lz[k], fmaf(ny, Ly[k], nx * Lx[k]));
      // view . (L - N*(2 L.N))  (:192-193 / :207-208; the sign of R is the reference's)
      const float vdl = -(fmaf(dz, Lz[k], fmaf(dy, Ly[k], dx * Lx[k])));
      const float vr = fmaf(-2.0f * ndl, vdn, vdl);
      const bool lit = int(__float_as_uint(m[k])) < 0;               // :187 / :202
      const float wd = lit ? max0(ndl) : 0.f;                        // :188 / :203
      const float ws = lit ? pow50(max0(vr)) : 0.f;                  // :193 / :208
      if (k == 0) {
        if (base == 0) { pd = wd; ps = ws; } else { ad += wd; as += ws; }
      } else { ad += wd; as += ws; }
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
  const LightSums w = lighting<MODE, (CUBES ? kCubeShadowBatch : kShadowBatch), false, CUBES, LIST>(p.L, sg, p.geom, nullptr, nullptr,
                               LIST ? p.occl_start + idx : p.occl + size_t(idx) * size_t(p.occl_words), p.occl_items, p.n_spheres, p.n_cubes, px, py, pz, nx, ny, nz, dx, dy, dz, sx, sy, sz,
                               level_key);
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

// ------------------------------------------------------------------------------------------
// VISIBILITY PASS (first half of the default path).  One thread per pixel, one wave per 8x8 tile
// (workgroup = 4 tiles side by side; grid = tile rows x tile columns/4 x frames, so no index
// arithmetic beyond blockIdx): primary ray (render_pixel :293-295) and its
// find_nearest_intersection over all N spheres.  Output per pixel: the hit sphere index as int16
// (2 B through L2/HBM; the consumer recomputes the hit distance with the same expressions), and
// the frame itself — black where the ray misses (:237-238), a placeholder where the stream kernel
// will write.  Per tile: one occupancy bit (`occ`), so the stream kernel visits only tiles with work.
// The kernel is issue-bound (profiles/r1e_*: ~500 instructions per wave, scalar ones included, and
// gfx950 issues scalar and vector instructions of a SIMD's waves from one ~1.2 instr/ns budget:
// profiles/r1_ubench_issue_mix.txt), so it has no LDS staging, no barrier and no per-lane divisions:
// the packed RGB rows of a tile (24 B each) are assembled across lanes with ds_bpermute.
// ------------------------------------------------------------------------------------------
// Bounding cone of the primary rays of window columns [xa, xb) x image rows [ya, yb): pixel x + r, y + r with r in
// [0, 1) -> u in [u0, u1], v in [v1, v0] (render_pixel :293-294); the directions right u + up v + forward form a
// planar patch, so the largest angle to the centre direction is at a corner.  Lanes 0..3 take a corner each, the
// others the centre.
__device__ __forceinline__ RayCone bundle_cone(const RenderParams& p, const rt_camera& cam, int lane, int xa, int xb,
                                               int ya, int yb) {
  const float u0 = float(xa) * p.inv_w * 2.0f - 1.0f, u1 = float(xb) * p.inv_w * 2.0f - 1.0f;
  const float v0 = -(float(ya) * p.inv_h * 2.0f - 1.0f) * p.inv_aspect, v1 = -(float(yb) * p.inv_h * 2.0f - 1.0f) * p.inv_aspect;
  const float ul = lane < 4 ? ((lane & 1) ? u1 : u0) : 0.5f * (u0 + u1);
  const float vl = lane < 4 ? ((lane & 2) ? v1 : v0) : 0.5f * (v0 + v1);
  float ex = fmaf(cam.up[0], vl, cam.right[0] * ul) + cam.forward[0];
  float ey = fmaf(cam.up[1], vl, cam.right[1] * ul) + cam.forward[1];
  float ez = fmaf(cam.up[2], vl, cam.right[2] * ul) + cam.forward[2];
  const float einv = frsq(fmaf(ez, ez, fmaf(ey, ey, ex * ex)));
  ex *= einv; ey *= einv; ez *= einv;
  auto from_lane = [](float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); };
  RayCone cone;
  cone.ax = from_lane(ex, 4); cone.ay = from_lane(ey, 4); cone.az = from_lane(ez, 4);
  const float cl = fmaf(ez, cone.az, fmaf(ey, cone.ay, ex * cone.ax));   // corners: cos of the angle to the axis
  float cmin = fminf(fminf(from_lane(cl, 0), from_lane(cl, 1)), fminf(from_lane(cl, 2), from_lane(cl, 3)));
  cmin = fmaxf(0.0f, fminf(1.0f, cmin) - 1e-5f);      // FP32 error of the above is ~1e-7
  cone.cos2_t = cmin * cmin;
  cone.sin_t = fsqrt(fmaxf(0.0f, 1.0f - cone.cos2_t));
  return cone;
}

// One block of the visibility pass: kVisTiles 8x8 tiles side by side, rows row0 .. row0 + 7 of the window, one wave.
template <int MODE, bool CUBES, bool MULTI, bool MANY = false>
__device__ __forceinline__ void visibility_block(const RenderParams& p, const int frame, const int row0, const int col0,
                                                 const int lane, const uint32_t region_lo = 0u, const uint32_t region_hi = 0u) {
  constexpr bool region_kept = MANY;      // (scenes of more than 64 spheres without cubes)
  // lane (lx, ly) traces pixel (lx, ly) of each tile
  constexpr int NT = kVisTiles;
  const int lx = lane & 7, ly = lane >> 3;
  const int rows = min(8, p.n_rows - row0);
  int cols[NT];
#pragma unroll
  for (int q = 0; q < NT; ++q) cols[q] = min(8, p.nx - col0 - 8 * q);
  if (cols[0] <= 0) return;                               // wave-uniform: both tiles beyond the right edge
  const long long frame_px = (long long)frame * p.frame_stride_px;
  const long long accum_px = (long long)frame * p.accum_stride_px;
  const bool last_sample = p.sample_index + 1 == p.n_samples;
  bool live[NT];
#pragma unroll
  for (int q = 0; q < NT; ++q) live[q] = lx < cols[q] && ly < rows;
  const long long wpix0 = (long long)(row0 + ly) * p.nx + (col0 + lx);      // second pixel: + 8
  int idx[NT];
  uint32_t rgbx[NT];                                      // hit pixels: placeholder
#pragma unroll
  for (int q = 0; q < NT; ++q) { idx[q] = -1; rgbx[q] = 0u; }
  // The common case: a whole block of an 8-bit frame with dword-aligned rows, one sample per pixel, nothing else
  // asked for.  Every pixel this kernel writes is then black (misses are final, hits get a placeholder), which
  // the short forms of the epilogue below rely on.
  const bool simple = !MULTI && p.n_samples == 1 && !p.out_f32 && !p.out_hit && !p.defer_output && p.out_rgb && rows == 8 &&
                      cols[NT - 1] == 8 && (p.nx & 3) == 0 && (frame_px & 3) == 0 &&
                      (reinterpret_cast<uintptr_t>(p.out_rgb) & 3) == 0;
  // The wave culls the spheres against the bounding cone of its rays first (primary_hit_culled): with many
  // spheres only the table groups that hold a survivor are tested per ray, and with ANY number of spheres a
  // block that no sphere reaches (sky: 81 % of config 2's tiles) builds no rays at all.  The first 64 spheres
  // are requested now, so the load is covered by the cone's arithmetic.
  const bool cull = RT_VIS_SKIP ? p.n_spheres > 0 : p.n_spheres > kVisCullFrom;
  float4 first_sphere = make_float4(0.f, 0.f, 0.f, 0.f);
  if (cull && !region_kept) first_sphere = p.geom[lane < p.n_spheres ? lane : p.n_spheres - 1];
  if (live[0] || cull) {                                  // (a later tile's pixel is live only if the first one's is;
    const FrameRec fr = as_const(p.frames)[frame];        //  with the cull all lanes take part in its ballots)
    const rt_camera& cam = fr.cam;
    const int y = tile_rows_uniform(p) ? tile_row_base(p, row0) + ly : row_of(p, row0 + ly);
    const CamGroup* table = p.camtab + size_t(frame) * cam_groups_stride(p.n_spheres);
    RayCone cone{0.f, 0.f, 0.f, 0.f, 0.f};
    unsigned long long first = 0ull;                      // spheres 0..63 that reach into the block's cone
    // Many spheres (region_kept): the region has culled ALL spheres against its own cone (lane c of region_lo / _hi: the
    // ballot of spheres 64 c ..); the block looks only at chunks that kept something, tests only the kept spheres against
    // its narrower cone, and keeps its ballots the same way — so it knows whether ANY sphere reaches it before it
    // builds a ray, as blocks of small scenes do.
    uint32_t keep_lo = 0u, keep_hi = 0u;
    unsigned long long reached = 0ull;
    if (cull) {
      const int ya = tile_rows_uniform(p) ? tile_row_base(p, row0) : row_of(p, row0);
      const int yb = (tile_rows_uniform(p) ? tile_row_base(p, row0) + 7 : row_of(p, min(row0 + 7, p.n_rows - 1))) + 1;
      cone = bundle_cone(p, cam, lane, p.x0 + col0, p.x0 + col0 + 8 * NT, ya, yb);
      if (region_kept) {
        for (int c = 0; c * 64 < p.n_spheres; ++c) {
          const unsigned long long rm = mask_of_chunk(region_lo, region_hi, c);
          if (rm == 0ull) continue;
          bool in = false;
          if ((rm >> lane) & 1ull) in = sphere_in_cone(p.geom[c * 64 + lane], cone, cam.origin[0], cam.origin[1], cam.origin[2]);
          const unsigned long long m = __ballot(in);
          if (lane == c) { keep_lo = uint32_t(m); keep_hi = uint32_t(m >> 32); }
          reached |= m;
        }
      } else {
        first = __ballot(lane < p.n_spheres && sphere_in_cone(first_sphere, cone, cam.origin[0], cam.origin[1], cam.origin[2]));
        reached = first;
      }
    }
    // a block that no sphere reaches builds no rays: all its pixels miss (with more than 64 spheres the later
    // chunks are not looked at here, and with cubes every ray is needed for them)
    bool cube_reach = false;
    if constexpr (CUBES) {   // (one cube per lane; beyond 64 cubes, or without a cone — no spheres at all —, every ray is built)
      cube_reach = true;
      if (cull && p.n_cubes <= 64) {
        const int c = lane < p.n_cubes ? lane : p.n_cubes - 1;
        cube_reach = __ballot(lane < p.n_cubes && box_in_double_cone(p.cubes[2 * c], p.cubes[2 * c + 1], cone, cam.origin[0],
                                                                     cam.origin[1], cam.origin[2])) != 0ull;
      }
    }
    const bool need_rays = !RT_VIS_SKIP || cube_reach || reached != 0ull || (p.n_spheres > 64 && !region_kept);
    // (nothing to write at all: a sample before the last of the MULTI path leaves only records and `present` bits
    // of tiles it hit; on the deferred-output path — refractive scenes — every pixel comes from rt_resolve_kernel)
    if (!need_rays && !p.out_hit && (MULTI ? !last_sample : p.defer_output != 0)) return;
    if constexpr (!MULTI) {
      // The common case of such a block — whole, one sample per pixel, 8-bit frame only, dword-aligned rows —
      // leaves at once: its rows are NT x 24 contiguous bytes of black, 12 per lane with NT = 4 (one store
      // instruction for the wave), and nothing else is owed (no record, no occupancy bit).
      if (!need_rays && simple) {
        uint8_t* o = p.out_rgb + (frame_px + (long long)(row0 + ly) * p.nx + col0) * 3;
        if constexpr (NT == 4) {
          uint32_t* w = reinterpret_cast<uint32_t*>(o + lx * 12);
          w[0] = 0u; w[1] = 0u; w[2] = 0u;
        } else {
          for (int i = lx; i < 6 * NT; i += 8) reinterpret_cast<uint32_t*>(o)[i] = 0u;
        }
        return;
      }
    }
    if (need_rays) {
    float dx[NT], dy[NT], dz[NT], t[NT];
#pragma unroll
    for (int q = 0; q < NT; ++q) {
      uint32_t pixel_key;
      primary_ray<MODE>(p, cam, fr.seed, p.x0 + col0 + 8 * q + lx, y, p.sample_index, pixel_key, dx[q], dy[q],
                        dz[q]);
    }
    if (p.n_spheres > kVisCullFrom)
      primary_hit_culled<NT>(p.geom, table, p.n_spheres, lane, cone, first, region_kept, keep_lo, keep_hi, cam.origin[0],
                             cam.origin[1], cam.origin[2], dx, dy, dz, t, idx);
    else   // few spheres: every group is scanned (pipelined scalar loads: cheaper here than group-by-group fetches)
      primary_hit_n<NT>(p.geom, table, p.n_spheres, cam.origin[0], cam.origin[1], cam.origin[2], dx, dy, dz, t, idx);
    if constexpr (CUBES) {   // the cubes of the object list (:159-160), wave-uniform scalar loads
      cgeom_t cg = as_const(p.cubes);
      float ix[NT], iy[NT], iz[NT];
#pragma unroll
      for (int q = 0; q < NT; ++q) { ix[q] = frcp(dx[q]); iy[q] = frcp(dy[q]); iz[q] = frcp(dz[q]); }
      for (int c = 0; c < p.n_cubes; ++c) {
        const float4 lo = cg[2 * c], hi = cg[2 * c + 1];
#pragma unroll
        for (int q = 0; q < NT; ++q) {
          float tc;
          int face;
          if (cube_t(lo, hi, cam.origin[0], cam.origin[1], cam.origin[2], dx[q], dy[q], dz[q], ix[q], iy[q], iz[q],
                     tc, face) && tc < t[q]) {
            t[q] = tc; idx[q] = p.n_spheres + c;
          }
        }
      }
    }
    }   // need_rays
    if constexpr (!MULTI) {
      if (simple) {
        // hand-over (see below) and black pixels of a whole block: a tile with a hit gets its record, header and
        // occupancy bit; a tile the stream kernel will not store gets 8 x 24 bytes of black, 48 lanes x one dword
        int16_t* rec = p.vis + (size_t)frame * p.vis_stride + ((size_t)(row0 >> 3) * p.tile_pitch + (col0 >> 3)) * kVisTile;
        const uint32_t id0 = uint32_t(frame) * uint32_t(p.tile_pitch * p.tiles8_y) + uint32_t(row0 >> 3) * uint32_t(p.tile_pitch) +
                             uint32_t(col0 >> 3);
        const int r = (lane * 43) >> 8, w = lane - 6 * r;          // lane l < 48: dword w of tile row r
        uint8_t* o = p.out_rgb + (frame_px + (long long)(row0 + r) * p.nx + col0) * 3 + w * 4;
#pragma unroll
        for (int q = 0; q < NT; ++q) {
          const bool occupied = __ballot(idx[q] >= 0) != 0ull;
          if (occupied) {
            rec[q * kVisTile + lane] = int16_t(idx[q]);
            if (lane == 0) {
              *reinterpret_cast<uint2*>(rec + q * kVisTile + 64) =
                  make_uint2(uint32_t((col0 >> 3) + q) | (uint32_t(row0 >> 3) << 16), uint32_t(frame));
              const uint32_t id = id0 + uint32_t(q);
              const uint32_t slot = id / kTileCounters, h = id % kTileCounters;
              atomicOr(p.occ + size_t(h) * size_t(p.occ_words) + (slot >> 5), 1u << (slot & 31u));
            }
          }
          if (!(occupied && p.tile_flush) && lane < 48) *reinterpret_cast<uint32_t*>(o + q * 24) = 0u;
        }
        return;
      }
    }
#pragma unroll
    for (int q = 0; q < NT; ++q) {
      if (!live[q]) idx[q] = -1;
      if (live[q] && p.out_hit && last_sample) p.out_hit[frame_px + wpix0 + 8 * q] = idx[q];
      if constexpr (!MULTI)
        if (live[q] && idx[q] < 0 && !p.defer_output)   // :237-238 miss -> black (resolves the running mean over samples too)
          finish_pixel(p, frame_px + wpix0 + 8 * q, accum_px + wpix0 + 8 * q, 0.f, 0.f, 0.f, &rgbx[q]);
    }
  }
  // tile-major hand-over: the 64 indices of a tile are 128 contiguous bytes; lanes outside the
  // window write -1, so the consumer needs no bounds checks ...
  // ... and a header names the tile, so a record is self-describing for whoever loads it
  int16_t* rec = p.vis + (size_t)frame * p.vis_stride + ((size_t)(row0 >> 3) * p.tile_pitch + (col0 >> 3)) * kVisTile;
  uint8_t* out = p.out_rgb + frame_px * 3;
  const bool write_rgb = p.out_rgb && last_sample && !p.defer_output;
  const bool dwords = (p.nx & 3) == 0 && (reinterpret_cast<uintptr_t>(out) & 3) == 0;
  const uint32_t id0 = uint32_t(frame) * uint32_t(p.tile_pitch * p.tiles8_y) + uint32_t(row0 >> 3) * uint32_t(p.tile_pitch) +
                       uint32_t(col0 >> 3);               // tile id of the wave's first tile; the others follow
  // MULTI (2..64 samples, all shaded by ONE stream-kernel launch): this launch traces sample p.sample_index of every
  // pixel.  A sample's hits go into the sample's own tile records; which samples of a tile hit anything is collected
  // in `present` (64 bits per tile, zeroed per batch), and the launch of the LAST sample, which sees all the others'
  // bits, completes sample 0's record — the one the stream kernel's tile supply loads — with a header that lists
  // them, sets the occupancy bit, and writes the pixels of tiles that no sample hit.
  unsigned long long before[NT];
  if constexpr (MULTI) {
#pragma unroll
    for (int q = 0; q < NT; ++q) before[q] = 0ull;
    if (last_sample) {
      const unsigned long long* pr = p.present + id0;      // (earlier launches' atomics: visible across the kernel boundary)
#pragma unroll
      for (int q = 0; q < NT; ++q) if (cols[q] > 0) before[q] = pr[q];
    }
  }
#pragma unroll
  for (int q = 0; q < NT; ++q) {
    if (cols[q] <= 0) break;                                  // wave-uniform: this tile and the ones after it do not exist
    // (a tile without a hit gets no record and no occupancy bit: the stream kernel never looks at it)
    const bool hit_now = __ballot(idx[q] >= 0) != 0ull;
    bool occupied = hit_now;
    if constexpr (MULTI) {
      int16_t* rec_s = rec + (size_t)p.sample_index * (size_t)p.vis_sample_stride;
      if (hit_now) rec_s[q * kVisTile + lane] = int16_t(idx[q]);
      if (!last_sample) {
        if (hit_now && lane == 0) atomicOr(p.present + id0 + q, 1ull << p.sample_index);   // no return value
        continue;
      }
      const unsigned long long all = before[q] | (hit_now ? 1ull << p.sample_index : 0ull);
      occupied = all != 0ull;
      if (occupied && !(all & 1ull)) rec[q * kVisTile + lane] = int16_t(-1);   // sample 0 hit nothing: its record says so
      if (occupied && lane == 0)
        *reinterpret_cast<uint4*>(rec + q * kVisTile + 64) =
            make_uint4(uint32_t((col0 >> 3) + q) | (uint32_t(row0 >> 3) << 16), uint32_t(frame), uint32_t(all), uint32_t(all >> 32));
    } else {
      if (occupied) rec[q * kVisTile + lane] = int16_t(idx[q]);
      if (occupied && lane == 0)
        *reinterpret_cast<uint2*>(rec + q * kVisTile + 64) =
            make_uint2(uint32_t((col0 >> 3) + q) | (uint32_t(row0 >> 3) << 16), uint32_t(frame));
    }
    if (occupied && lane == 0) {   // tile id = slot * kTileCounters + counter (see the stream kernel's tile supply)
      const uint32_t id = id0 + uint32_t(q);
      const uint32_t slot = id / kTileCounters, h = id % kTileCounters;
      atomicOr(p.occ + size_t(h) * size_t(p.occ_words) + (slot >> 5), 1u << (slot & 31u));   // no return value: fire and forget
    }
    // (tile_flush: the stream kernel stores an occupied tile whole, its black pixels included)
    if (write_rgb && !(occupied && p.tile_flush))
      store_tile8(out, p.nx, row0, col0 + 8 * q, rows, cols[q], lane, rgbx[q], dwords);
  }
}


// The kernel: a workgroup is 4 waves side by side; a wave takes kVisRows blocks one below the other — a region of
// 32 x 32 pixels with the defaults.  Before anything else the wave culls the scene against the cone of the WHOLE
// region: most regions of a mostly-sky frame hold nothing, and such a region of a plain 8-bit frame leaves as 32
// rows of black from one cone and one cull turn (a quarter of what its four blocks would spend to find out the same).
// Six blocks (48 rows) per region: against four, +4.9 % on config 2, +4 % on config 5, +2.7 % on config 3, +3.6 % on the
// glass and cube scenes, configs 1 and 4 and the 64-sphere target unchanged within 0.6 %.  Five: config 3 -6 %; seven:
// config 1 -2 %; eight: config 1 -29 % (A/B in one call each; twelve measures like six on config 2, -1.4 % on config 1).
#ifndef RT_VIS_ROWS
#define RT_VIS_ROWS 6
#endif
// Workgroups of TWO waves (two regions side by side): a workgroup's wave slots come free when its last wave ends, and most
// regions are empty and over in ~100 instructions while a region with spheres in it runs six blocks of rays — with four
// waves per workgroup the pass held 19 % of its wave slots.  Against 256 threads (all at six blocks): config 2 +1.1 %, the
// 64-sphere target +4.7 %, config 3 +2.3 %, config 4 +1.9 %; 64 threads -5.7 % on config 2 (dispatch), 512 / 1024: -8.5 / -13 %.
#ifndef RT_VIS_BLOCK
#define RT_VIS_BLOCK 128
#endif
constexpr int kVisRows = RT_VIS_ROWS;
template <int MODE, bool CUBES, bool MULTI = false, bool MANY = false>
__global__ __launch_bounds__(RT_VIS_BLOCK, RT_VIS_WAVES) void rt_visibility_kernel(const RenderParams p) {
  static_assert(!MANY || !CUBES, "the many-sphere form of the pass exists for scenes without cubes");
  constexpr int NT = kVisTiles;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int frame = blockIdx.z;
  const int col0 = (int(blockIdx.y) * (RT_VIS_BLOCK / 64) + wave) * (8 * NT), rowR = int(blockIdx.x) * (8 * kVisRows);   // window coordinates
  if (p.nx - col0 <= 0) return;                           // wave-uniform: beyond the right edge
  constexpr bool region_kept = MANY;
  bool region_any = false;
  uint32_t region_lo = 0u, region_hi = 0u;      // many spheres: lane c holds the ballot of spheres 64 c .. against the region's cone
  if constexpr (RT_VIS_SKIP && kVisRows > 1) {
    const long long frame_px = (long long)frame * p.frame_stride_px;
    const bool whole = rowR + 8 * kVisRows <= p.n_rows && p.nx - col0 >= 8 * NT;
    const bool nothing_owed = !p.out_hit && (MULTI ? p.sample_index + 1 != p.n_samples : p.defer_output != 0);
    // (MULTI: the launch of the last sample writes the black of tiles that NO sample hit — `present` knows)
    const bool black_owed = (MULTI ? p.sample_index + 1 == p.n_samples : p.n_samples == 1) && !p.out_f32 && !p.out_hit &&
                            !p.defer_output && p.out_rgb && whole && (p.nx & 3) == 0 && (frame_px & 3) == 0 &&
                            (reinterpret_cast<uintptr_t>(p.out_rgb) & 3) == 0;
    if constexpr (MANY) {
      {                         // many spheres: every chunk of 64 against the region's cone, the ballots kept for the blocks
        const FrameRec fr = as_const(p.frames)[frame];
        const int last = min(rowR + 8 * kVisRows, p.n_rows) - 1;
        const int ya = p.tile_stride == 1 ? p.y0 + p.tile_first * p.tile_rows + rowR : row_of(p, rowR);
        const int yb = (p.tile_stride == 1 ? p.y0 + p.tile_first * p.tile_rows + last : row_of(p, last)) + 1;
        const RayCone cone = bundle_cone(p, fr.cam, lane, p.x0 + col0, p.x0 + col0 + 8 * NT, ya, yb);
        unsigned long long reached = 0ull;
        float4 sphere = p.geom[lane];
        for (int c = 0; c * 64 < p.n_spheres; ++c) {
          const int i = c * 64 + lane, nxt = min(i + 64, p.n_spheres - 1);
          const float4 ahead = p.geom[nxt];                 // (the next chunk's spheres are on their way during this one's test)
          const unsigned long long m = __ballot(i < p.n_spheres && sphere_in_cone(sphere, cone, fr.cam.origin[0], fr.cam.origin[1], fr.cam.origin[2]));
          if (lane == c) { region_lo = uint32_t(m); region_hi = uint32_t(m >> 32); }
          reached |= m;
          sphere = ahead;
        }
        region_any = reached != 0ull;
      }
    }
    if ((nothing_owed || black_owed) && (p.n_spheres <= 64 || region_kept) && (!CUBES || (p.n_spheres > 0 && p.n_cubes <= 64))) {
      bool any = region_any;
      if (p.n_spheres > 0 && !region_kept) {
        const float4 sphere = p.geom[lane < p.n_spheres ? lane : p.n_spheres - 1];
        const FrameRec fr = as_const(p.frames)[frame];
        const int last = min(rowR + 8 * kVisRows, p.n_rows) - 1;
        const int ya = p.tile_stride == 1 ? p.y0 + p.tile_first * p.tile_rows + rowR : row_of(p, rowR);
        const int yb = (p.tile_stride == 1 ? p.y0 + p.tile_first * p.tile_rows + last : row_of(p, last)) + 1;
        const RayCone cone = bundle_cone(p, fr.cam, lane, p.x0 + col0, p.x0 + col0 + 8 * NT, ya, yb);
        any = __ballot(lane < p.n_spheres && sphere_in_cone(sphere, cone, fr.cam.origin[0], fr.cam.origin[1], fr.cam.origin[2])) != 0ull;
        if constexpr (CUBES) {
          const int c = lane < p.n_cubes ? lane : p.n_cubes - 1;
          any = any || __ballot(lane < p.n_cubes && box_in_double_cone(p.cubes[2 * c], p.cubes[2 * c + 1], cone, fr.cam.origin[0],
                                                                       fr.cam.origin[1], fr.cam.origin[2])) != 0ull;
        }
      }
      if constexpr (MULTI) {
        if (!any && black_owed) {   // did an earlier sample hit a tile of the region?  (lane l: tile l % NT of block l / NT)
          const uint32_t id = uint32_t(frame) * uint32_t(p.tile_pitch * p.tiles8_y) +
                              uint32_t((rowR >> 3) + lane / NT) * uint32_t(p.tile_pitch) + uint32_t((col0 >> 3) + lane % NT);
          any = __ballot(lane < NT * kVisRows && p.present[id] != 0ull) != 0ull;
        }
      }
      if (!any) {
        if (black_owed) {
          const int lx = lane & 7, ly = lane >> 3;
          uint8_t* o = p.out_rgb + (frame_px + (long long)(rowR + ly) * p.nx + col0) * 3;
#pragma unroll
          for (int b = 0; b < kVisRows; ++b) {
            uint8_t* ob = o + (long long)(8 * b) * p.nx * 3;
            if constexpr (NT == 4) {
              uint32_t* w = reinterpret_cast<uint32_t*>(ob + lx * 12);
              w[0] = 0u; w[1] = 0u; w[2] = 0u;
            } else {
              for (int i = lx; i < 6 * NT; i += 8) reinterpret_cast<uint32_t*>(ob)[i] = 0u;
            }
          }
        }
        return;
      }
    }
  }
#pragma nounroll
  for (int b = 0; b < kVisRows; ++b) {
    const int row0 = rowR + 8 * b;
    if (row0 >= p.n_rows) break;
    // (the argument block is read afresh per block, as in the stream kernel: hoisted out of this loop its fields
    // and everything derived from them would not fit the scalar registers)
    int lane_b = int(threadIdx.x) & 63;
    asm volatile("" : "+v"(lane_b));
    visibility_block<MODE, CUBES, MULTI, MANY>(params_here(), frame, row0, col0, lane_b, region_lo, region_hi);
  }
}

// ------------------------------------------------------------------------------------------
// STREAM KERNEL (second half of the default path): persistent WAVES, each with its own work lists.
//
// The launch holds 4 workgroups of 4 waves per CU (4 waves per SIMD).  A workgroup only shares the
// LDS copy of the sphere list; after the one barrier that publishes it, every wave runs on its
// own.  Work arrives as 8x8 tiles of the frames of the batch, claimed from global counters until
// none is left.  Per wave, in LDS, one work queue of records {P, D, T, acc, pixel,
// level|frame|object, pixel key} of rays that hit something:
//   intake : one coalesced load of a tile's record (the visibility pass's hit indices + header);
//            the tile's frame is wave-uniform, so its camera and seed come by scalar load, and
//            every hit pixel rebuilds its primary ray and hit point exactly as the visibility
//            pass measured them and becomes a level-0 record (ballot + prefix count).
//   pass   : the 64 records on top of the queue; per lane compute_lighting with its shadow tests,
//            the throughput update, the reflection ray and its nearest-hit scan; a new hit goes
//            back onto the queue, anything else finishes the pixel (3 byte stores into the frame
//            the visibility pass has already cleared).
// The queue lives across tiles and frames, so a wave runs partially filled only in its final
// drain — not wherever pixels miss or die at different depths (the pixel kernel below, kept for
// A/B evidence).  No barrier and no LDS atomic sits on this path: waves drift apart, so one
// wave's memory-latency-bound intake overlaps another's VALU-bound shading on the same SIMD.
//
// Work distribution.  A launch is short (a 16-frame 1080p batch: ~50 passes per wave) and the cost
// of a tile ranges from nothing (sky) to 6 passes (64 pixels of facing mirrors), so the unit handed
// out must be small and a wave's share must not be spatially coherent:
//   * tile ids are dealt round-robin to kTileCounters counters on separate cache lines — counter h
//     hands out tiles h, h + K, h + 2K, ... ("slots").  One address serves returning atomics at only
//     ~23 ns each on this part; K addresses work in parallel, and a claim of consecutive slots is K
//     tiles apart in the frame: decorrelated.
//   * guided self-scheduling: a claim is 1/(2 x waves per counter) of what its counter had left,
//     between RT_CLAIM_MIN and kClaimMax slots, so private backlogs shrink towards the end of the launch.
//   * only OCCUPIED tiles are loaded: the visibility pass sets one bit per tile in which some pixel
//     hit something (`occ`, laid out per counter so that a claim's slots are consecutive bits); when
//     a claim lands, the two words holding its bits are fetched by scalar loads, and a refill later
//     they say which of its tiles exist for this kernel.  81 % of config 2's tiles are sky: taking
//     each of them through the ring cost a quarter of the kernel's instruction issue.
//   * nothing on the supply side waits for memory between two passes: the NEXT claim's atomic is in
//     flight while the current claim is worked through, and the index data of the next kRing tiles
//     is brought in by LDS-direct loads (global_load_lds_dword: 2 tiles = 256 bytes per instruction,
//     no registers held across the lighting loops) issued before a pass and read after it.  A
//     wave's first tile is its own global wave number (no atomic at the start).
//   * a wave whose counter runs dry re-homes: one coherent load of all K counters, the next one
//     that still has tiles becomes its counter, and claiming goes on there through the same
//     pipelined, guided-size path (one-tile synchronous claims at the end cost 8-9 % at 64 spheres).
// ------------------------------------------------------------------------------------------
constexpr int kBounceCap = 128;                       // <= 63 records when a tile is taken in (<= 64 more); a pass takes 64 and pushes <= 64
constexpr int kQueueFields = 15;     // P(3) D(3) T(3) acc(3) pixel(1) level|frame|sphere(1) pixel key(1)
constexpr int kQueueFieldsSlots = 14; // tile slots: the pixel is a slot and a position in it, part of the meta word
constexpr int kQueueFieldsRefr = 12; // refraction: no acc (nodes add into the pixel sums); pixel, meta and key are fields 9..11
constexpr int kQueueFieldsRefrSlots = 11; // ... with tile slots (the pixel is a slot and a position in it): meta and key are fields 9, 10
constexpr int queue_fields(bool refr, int out = 0) {
  return refr ? (out ? kQueueFieldsRefrSlots : kQueueFieldsRefr) : out ? kQueueFieldsSlots : kQueueFields;
}
// Tile slots (SLOTS variant of the stream kernel): the pixels of a tile leave together.  Finished pixels are
// staged in LDS, one RGBX word each, in one of kTileSlots records per wave; when the last hit pixel of a tile
// has finished, the wave stores the tile as the visibility pass stores an empty one (store_tile8: whole dwords,
// 8 x 24 contiguous bytes).  Scattered 3-byte stores of single pixels cost a fabric write request each
// (profiles/r1m_c2_*: 3.7e8 requests for 3.9e8 hit pixels per 1024-frame launch, 20 x the bytes they carry).
#ifndef RT_TILE_SLOTS
#define RT_TILE_SLOTS 8
#endif
constexpr int kTileSlots = RT_TILE_SLOTS;
static_assert(kTileSlots >= 2 && kTileSlots <= 8, "slot | pixel takes 9 bits of a record's meta word");
// All samples of a pixel in one launch pair (MULTI, 2..64 samples, render_pixel :292-297): a slot sums its
// tile's samples in LDS as three floats per pixel and takes them ONE AFTER ANOTHER, in sample order — the sum
// is the one the per-sample launches form in `accum`, to the last bit — while other slots keep the passes full.
// It holds the tile record of its next sample (prefetched) besides.
#ifndef RT_TILE_SLOTS_MULTI_SMALL
#define RT_TILE_SLOTS_MULTI_SMALL 3   // OUT = 3: few enough slots (and one ring entry) for FOUR workgroups per CU on scenes of up to 36 spheres
#endif
#ifndef RT_TILE_SLOTS_MULTI
#define RT_TILE_SLOTS_MULTI 6
#endif
constexpr int kTileSlotsMulti = RT_TILE_SLOTS_MULTI;
static_assert(kTileSlotsMulti >= 2 && kTileSlotsMulti <= 8, "slot | pixel takes 9 bits of a record's meta word");
constexpr int kSlotHdr = 4;                                 // words: tx | ty << 16, frame, samples still to take (64 bits)
// Refractive scenes (REFR with tile slots): a hit has two children, so a pixel is a TREE of records that end in different
// passes.  All of them stay in the wave that took the tile in (queue and spill stack are per wave), so the tile's slot
// holds three SUMS per pixel that every node of the tree adds its share to, and a count of the tile's records still
// alive (16 bits per slot) instead of a count of unfinished pixels.  The sums are 32-bit FIXED-POINT (24 fractional
// bits, LDS integer atomics, saturating): integer addition is associative, so a pixel's bytes depend on its tree alone
// — not on the order in which the passes happened to reach its nodes, which varies with the work distribution from run
// to run and from tiling to tiling (float atomics into a slab in device memory, the earlier form, gave neither
// repeatable bytes nor acceptable write traffic: 22 x the frame's bytes on glass cubes).
#ifndef RT_TILE_SLOTS_REFR
#define RT_TILE_SLOTS_REFR 2     // (older records are shaded first, so few tiles are in flight: at four workgroups per CU two slots measured
                                 // +2 % over three on the glass spheres and cubes, -1 % on 40 spheres; four slots cost the fourth workgroup: -17 %)
#endif
constexpr int kTileSlotsRefr = RT_TILE_SLOTS_REFR;
static_assert(kTileSlotsRefr >= 2 && kTileSlotsRefr <= 5, "five 12-bit record counts in one scalar register pair");
constexpr float kSumScale = 16777216.0f;                    // 2^24: a sum saturates at 256 (x samples <= 64: the mean still clamps to 1)
constexpr int tile_slots(int out, bool refr = false) {
  return refr ? kTileSlotsRefr : out >= 3 ? RT_TILE_SLOTS_MULTI_SMALL : out == 2 ? kTileSlotsMulti : kTileSlots;
}
constexpr int slot_bytes(int out, bool refr = false) {     // per wave: stage (RGBX words, or 3 sums per pixel) + header (+ next record)
  return out == 0 ? 0
       : refr ? kTileSlotsRefr * (3 * 64 * 4 + kSlotHdr * 4 + (out >= 2 ? 72 * 2 : 0))
       : out == 1 ? kTileSlots * (64 * 4 + kSlotHdr * 4) : tile_slots(out) * (3 * 64 * 4 + kSlotHdr * 4 + 72 * 2);
}
// meta word of a SLOTS record: level (7 bits) | slot * 64 + pixel (9 bits) | object (12 bits, from bit 20)
__device__ __forceinline__ uint32_t pack_meta_slot(int level, int slotpix, int object) {
  return uint32_t(level) | (uint32_t(slotpix) << 7) | (uint32_t(object) << 20);
}
__device__ __forceinline__ int meta_slotpix(uint32_t m) { return int((m >> 7) & 511u); }
#ifndef RT_RING
#define RT_RING 2     // 2, 4 and 8 measure the same since only occupied tiles are loaded
#endif
constexpr int kRing = RT_RING;             // tiles whose index data is in LDS or on its way there (power of two)
#ifndef RT_RING_REFR
#define RT_RING_REFR 2     // (one entry: -4 % on the glass scenes)
#endif
constexpr int ring_of(int out, int refr = 0) { return (refr && out) ? RT_RING_REFR : out >= 3 ? 1 : kRing; }   // (one entry costs the opaque one-sample path 9 %; OUT = 3 and the refractive slot kernels buy a workgroup per CU with it)
#ifndef RT_CLAIM
#define RT_CLAIM 32    // slots per claim at most: one occupancy word's worth
#endif
constexpr int kClaimMax = RT_CLAIM;
#ifndef RT_CLAIM_MIN
#define RT_CLAIM_MIN 16 // slots per claim at least: ~3 occupied tiles at config 2's 19 % (with 1, the last claims of a launch are mostly empty: a round trip each)
#endif
#ifndef RT_GUIDE
#define RT_GUIDE 2      // a claim is 1/(RT_GUIDE x waves per counter) of what its counter has left
#endif

// Refraction: a lane may push two records, so the queue is longer: max_depth + 2 x 64 (see the pass sizing in the kernel).
// Two kinds of refractive kernels (the REFR template argument): 1 = max_depth <= kRefrShallowDepth, 160 records — with two
// tile slots 8.9 KB of LDS per wave, FOUR workgroups per CU; 2 = any depth up to RT_MAX_DEPTH, 192 records, three workgroups
// per CU (measured on the glass scenes: three workgroups are 15 - 20 % slower than four).
constexpr int kRefrShallowDepth = 32;
constexpr int kBounceCapRefrShallow = kRefrShallowDepth + 2 * 64;
constexpr int kBounceCapRefr = RT_MAX_DEPTH + 2 * 64;
constexpr int bounce_cap(int refr) { return refr == 2 ? kBounceCapRefr : refr == 1 ? kBounceCapRefrShallow : 128; }
// Refraction: the fields a record carries (no acc: nodes add into the pixel sums), in the order they are set aside
constexpr int kSpillFields = 12;                     // (the most: a block holds whole records of 11 or 12 fields)
static_assert(kSpillFields == kQueueFieldsRefr && kQueueFieldsRefrSlots <= kSpillFields, "a spill block holds whole records");
constexpr int kSpillBlocksMax = 32;                   // blocks of 64 records a wave may have set aside (3 KB each)
static_assert(kBounceCapRefr + kSpillBlocksMax * 64 < 4096, "a slot's count of live records takes 12 bits");
template <int CAP>
struct Queue {
  float* f;        // kQueueFields arrays of CAP words (field-major: conflict-free lane access)
  __device__ __forceinline__ float& at(int field, int i) const { return f[field * CAP + i]; }
};
constexpr int wave_lds_bytes(int refr, int out = 0) {
  return queue_fields(refr != 0, out) * bounce_cap(refr) * 4 + ring_of(out, refr) * kVisTile * 2 + slot_bytes(out, refr != 0);
}

// LDS accesses of one wave are executed in program order; this only stops the compiler from moving
// list reads and writes of different lanes across each other
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ int lanes_below(unsigned long long mask) {   // set bits of mask below this lane
  return int(__builtin_amdgcn_mbcnt_hi(uint32_t(mask >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(mask), 0u)));
}

// Refraction: the bottom 64 records of a wave's queue go to its spill block `dst`, the rest moves down
// (lane l touches positions l, 64 + l and 128 + l only: no ordering between lanes is needed)
template <int CAP, int FIELDS>
__device__ __forceinline__ void queue_set_aside(float* qf, float* dst, int lane, int n) {
  Queue<CAP> q{qf};
  const bool row2 = 128 + lane < n;
#pragma unroll
  for (int f = 0; f < FIELDS; ++f) {
    dst[f * 64] = q.at(f, lane);
    const float a = q.at(f, 64 + lane);                    // (64 + lane < 128 <= n)
    const float b = row2 ? q.at(f, 128 + lane) : 0.f;
    q.at(f, lane) = a;
    if (row2) q.at(f, 64 + lane) = b;
  }
}
template <int CAP, int FIELDS>
__device__ __forceinline__ void queue_take_back(float* qf, const float* src, int at) {
  Queue<CAP> q{qf};
#pragma unroll
  for (int f = 0; f < FIELDS; ++f) q.at(f, at) = src[f * 64];
}

// OUT: how finished pixels leave — 0: one by one (float / multi-sample / refraction paths), 1: whole tiles from
// RGBX staging in per-wave tile slots, 2: the same with all samples of a pixel summed in the slot (MULTI), 3: MULTI
// with three slots and one ring entry per wave — 40.4 KB of LDS per workgroup besides the spheres, so that scenes of up
// to 36 spheres keep FOUR workgroups per CU (OUT = 2: six slots, three workgroups, 133 VGPRs), 4: the same as TWO
// workgroups of eight waves, which share one copy of the spheres: scenes of up to 72 spheres (the 64-sphere target)
// GRID: 0 = few spheres (bit tables), 1 = many spheres (grid or lists for reflection rays, lists for shadow rays), 2 = the
// same as workgroups of EIGHT waves around one copy of the sphere list: from 513 spheres on a copy per four waves leaves
// LDS for three workgroups per CU, one per eight waves for two of twice the size (config 4: 308 -> 334 Grays/s).
// (Before that, 2 was the four-wave form compiled for three waves per SIMD — 156 VGPRs, nothing spilled: 271 -> 281.)
// 3 = SIXTEEN waves around one copy (from ~1100 spheres on, where two eight-wave workgroups no longer fit a CU).
template <int MODE, int REFR, bool CUBES, int GRID = 0, int OUT = 0>
__global__ __launch_bounds__((GRID == 3 ? 1024 : (GRID == 2 || OUT == 4) ? 512 : kBlock), (REFR && OUT) ? RT_REFR_WAVES : (OUT == 2) ? 3 : RT_WAVES_PER_SIMD) void rt_stream_kernel(const RenderParams) {
  constexpr int kThreads = GRID == 3 ? 1024 : (GRID == 2 || OUT == 4) ? 512 : kBlock;
  // (the argument block is read through params_here(), phase by phase: never as a whole)
  constexpr bool SLOTS = OUT != 0, MULTI = OUT >= 2;
  constexpr bool SUMS = MULTI || (REFR && SLOTS);     // a slot holds three sums per pixel (else one RGBX word)
  constexpr int kRing = ring_of(OUT, REFR);
  static_assert(GRID == 0 || (!REFR && !CUBES), "the grid variant exists for sphere-only opaque scenes");
  static_assert(!REFR || OUT <= 2, "refractive scenes: scattered sums (0), tile slots (1), all samples in one launch (2)");
  constexpr int kCap = bounce_cap(REFR);
  static_assert(kBounceCap == bounce_cap(0), "opaque scenes: a pass takes 64 records and pushes at most 64");
  constexpr int kWaveLdsBytes = wave_lds_bytes(REFR, OUT);
  constexpr int kFields = queue_fields(REFR, OUT);
  constexpr int kSlots = tile_slots(OUT, REFR != 0);
  // what a slot still waits for, one count per slot in a scalar register pair: hit pixels of the tile (at most 64: a byte),
  // or — REFR — records of its pixels' trees that are alive in the queue or the spill stack (at most kCap + 32 x 64 = 2240: 12 bits)
  constexpr int kCntBits = REFR ? 12 : 8;
  static_assert(kSlots * kCntBits <= 64, "the slots' counts share one scalar register pair");
  extern __shared__ __align__(16) unsigned char smem[];
  float4* sg = reinterpret_cast<float4*>(smem);                                // spheres, then cubes (2 entries each)
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // The lane number, from a source the compiler cannot trace: every phase of the main loop takes a fresh one, so
  // that nothing derived from it (LDS addresses of the wave's queue, ring and slots, pixel offsets in a tile) is
  // carried across lighting() in registers it does not have: such values went to scratch at the top of the kernel
  // and came back every pass, and the wait for them also waited for the tile loads that are meant to fly.
  auto lane_here = [&]() { int l = int(threadIdx.x) & 63; asm volatile("" : "+v"(l)); return l; };
  int total_tiles;
  unsigned char* mine;                                       // wave-uniform
  {
    const RenderParams& p = params_here();
    const int geom_bytes = (p.n_spheres + 2 * p.n_cubes) * 16;
    mine = smem + geom_bytes + wave * kWaveLdsBytes;
    for (int i = tid; i < p.n_spheres; i += kThreads) sg[i] = p.geom[i];
    if constexpr (CUBES)
      for (int i = tid; i < 2 * p.n_cubes; i += kThreads) sg[p.n_spheres + i] = p.cubes[i];
    const int tiles_per_frame = p.tile_pitch * p.tiles8_y;   // tile ids, padding ids (no record, no occupancy bit) included
    total_tiles = tiles_per_frame * p.n_frames;
  }
  Queue<kCap> q{reinterpret_cast<float*>(mine)};
  int16_t* ring = reinterpret_cast<int16_t*>(mine + kFields * kCap * 4);   // [kRing][kVisTile] tile records
  // SLOTS: per wave kTileSlots tiles in progress — finished pixels as RGBX words and a header in LDS; the number
  // of hit pixels each still waits for in one byte of a scalar register pair (a tile has at most 64)
  uint32_t* slot_stage = reinterpret_cast<uint32_t*>(mine + kFields * kCap * 4 + kRing * kVisTile * 2);   // [kSlots][64] RGBX
  float* slot_sum = reinterpret_cast<float*>(slot_stage);                                                   // MULTI: [kSlots][3][64]
  uint32_t* slot_fix = slot_stage;                                                                          // REFR: the same as fixed-point sums
  uint32_t* slot_hdr = slot_stage + kSlots * 64 * (SUMS ? 3 : 1);                                           // [kSlots][kSlotHdr]
  int16_t* slot_next = reinterpret_cast<int16_t*>(slot_hdr + kSlots * kSlotHdr);                            // MULTI: [kSlots][kVisTile]
  uint32_t slot_free = (1u << kSlots) - 1u;                                                                 // wave-uniform
  unsigned long long slot_left = 0ull;                                                                      // field k (kCntBits bits): slot k
  uint32_t slot_ready = 0u;       // MULTI: slots whose current sample is finished and that have another to take in
  constexpr int kFPixel = REFR ? 9 : 12;                                        // field numbers (the pixel: without slots only)
  constexpr int kFMeta = REFR ? (SLOTS ? 9 : 10) : SLOTS ? 12 : 13, kFKey = kFMeta + 1;
  __syncthreads();   // the only workgroup-wide synchronisation

  const int gwave = __builtin_amdgcn_readfirstlane(int(blockIdx.x) * (kThreads / 64) + wave);
  const int n_waves = int(gridDim.x) * (kThreads / 64);
  uint32_t n_trace = 0, n_shade = 0;
  [[maybe_unused]] unsigned long long st_fetch = 0, st_pop = 0, st_light = 0, st_near = 0;     // (read in RT_STAMPS builds only)
  [[maybe_unused]] const unsigned long long st_begin = RT_NOW();
  int n_bounce = 0;               // wave-uniform queue length
  bool done = false;              // no tile is left for this wave

  // ---- tile supply (all wave-uniform).  Slot s of counter h is tile s * K + h; counter h has
  // statically handed its first ceil((n_waves - h) / K) slots to the waves with number h mod K.
  auto first_of = [&](int h) { return (n_waves - h + kTileCounters - 1) / kTileCounters; };
  auto slots_of = [&](int h) { return (total_tiles - h + kTileCounters - 1) / kTileCounters; };
  const int waves_per_counter2 = RT_GUIDE * max(1, n_waves / kTileCounters);
  auto claim_size = [&](int h, int slot_seen) {   // from the counter position this wave saw last
    return max(RT_CLAIM_MIN, min(kClaimMax, (slots_of(h) - slot_seen) / waves_per_counter2));
  };
  int hcur = gwave % kTileCounters;
  uint32_t dry = 0u;               // counters this wave has seen exhausted
  // (tile records are self-describing and lie in tile-id order, frame after frame, so the supply side
  // deals in tile ids only: no decoding into frame / row / column here)
  // current claim: occupancy bits (visibility pass: "some pixel of the tile hit something") of its
  // slots not yet started, and the tile id of bit 0.  Empty tiles (81 % of config 2's) are never loaded.
  uint32_t cl_mask = 0u;
  int cl_base = 0;
  if (gwave < total_tiles) {   // (a padding id or an empty tile has no bit)
    const RenderParams& p = params_here();
    const int slot0 = gwave / kTileCounters;
    cl_mask = (as_const(p.occ)[size_t(hcur) * size_t(p.occ_words) + (slot0 >> 5)] >> (slot0 & 31)) & 1u;
    cl_base = gwave;
  }
  bool homeless = false;           // the counter this wave drew from is exhausted: it must find another
  int pf_n = 0, pf_c = 0;          // the claim being requested (lane 0 receives the counter value)
  bool nx_valid = false;           // the claim after the current one: slots, first tile id, and the two
  int nx_n = 0, nx_id = 0, nx_shift = 0;   // occupancy words that hold its bits from bit nx_shift on (scalar
  uint32_t nx_w0 = 0u, nx_w1 = 0u;         // loads issued when the claim lands, used a refill later)
  int seen = first_of(hcur);       // position of the home counter when this wave last looked
  // ---- ring of tiles whose index data is in LDS or on its way (FIFO: head = next to consume)
  uint32_t head = 0, tail = 0;     // tile counts
  bool supply_done = false;
  [[maybe_unused]] unsigned long long st_topup = 0, st_wait = 0;
  [[maybe_unused]] unsigned st_passes = 0, st_lanes = 0, st_starved = 0;

  // ---- REFR: records set aside in global memory (see the pass sizing below), a stack of 64-record blocks
  int sp_blocks = 0;
  auto sp_block = [&](int b) {   // (derived from the kernel arguments where it is needed: the main loop is short of scalar registers)
    const RenderParams& p = params_here();
    return p.spill + (size_t(gwave) * size_t(p.spill_blocks) + size_t(b)) * size_t(kSpillFields * 64) + lane_here();   // (room for 12 fields)
  };

  // ---- a tile's hit pixels (one sample of them) become level-0 records on the queue.  The tile's frame is
  // wave-uniform: its camera and seed come by scalar load; every hit pixel rebuilds its primary ray and hit
  // point exactly as the visibility pass measured them, with all lanes of the tile at work.
  auto take_in = [&](int idx, unsigned long long mask, uint32_t m0, uint32_t m1, int sample, int k) {
    int lane = lane_here();
    if constexpr (REFR && SLOTS) {
      // Older records first: what is left in the queue (fewer than 64 records, of tiles taken in earlier) moves up above
      // the new tile's hits, which go to the bottom.  Passes take from the top, so the trees of a tile are finished
      // before those of the next one are begun, a tile's slot comes free as early as it can, and few slots suffice
      // (new hits on top would starve the records below them for as long as tiles keep coming).  One lane per record.
      const int m = __popcll(mask);
      if (n_bounce > 0 && m > 0) {
        float rec[kFields];
        if (lane < n_bounce) {
#pragma unroll
          for (int f = 0; f < kFields; ++f) rec[f] = q.at(f, lane);
        }
        wave_lds_fence();
        if (lane < n_bounce) {
#pragma unroll
          for (int f = 0; f < kFields; ++f) q.at(f, m + lane) = rec[f];
        }
        wave_lds_fence();
        lane = lane_here();
      }
    }
    const RenderParams& p = params_here();
    const int col = int((m0 & 0xFFFFu) * 8u) + (lane & 7), row = int((m0 >> 16) * 8u) + (lane >> 3);
    const int frame = int(m1);
    const FrameRec fr = as_const(p.frames)[frame];
    const int slotpix = k * 64 + lane;
    if constexpr (SLOTS) slot_left |= (unsigned long long)__popcll(mask) << (kCntBits * k);      // (the field is 0 now)
    if (idx >= 0) {
      const rt_camera& cam = fr.cam;
      const int y = p.tile_stride == 1 ? p.y0 + p.tile_first * p.tile_rows + row : row_of(p, row);
      uint32_t pixel_key;
      float dx, dy, dz, px, py, pz;
      primary_ray<MODE>(p, cam, fr.seed, p.x0 + col, y, MULTI ? sample : p.sample_index, pixel_key, dx, dy, dz);
      if (CUBES && idx >= p.n_spheres) {
        float t = 0.f;
        int face = 0;
        const int c = idx - p.n_spheres;
        cube_t(sg[p.n_spheres + 2 * c], sg[p.n_spheres + 2 * c + 1], cam.origin[0], cam.origin[1], cam.origin[2],
               dx, dy, dz, frcp(dx), frcp(dy), frcp(dz), t, face);
        hit_point<CUBES>(sg, p.n_spheres, idx, face, cam.origin[0], cam.origin[1], cam.origin[2], dx, dy, dz, t,
                         px, py, pz);                                                                        // :168
      } else {
        const float inv_a = frcp(fmaf(dz, dz, fmaf(dy, dy, dx * dx)));
        float dl;
        const float t = sphere_t_exact(sg[idx], cam.origin[0], cam.origin[1], cam.origin[2], dx, dy, dz, inv_a, dl);
        px = fmaf(dx, t, cam.origin[0]); py = fmaf(dy, t, cam.origin[1]); pz = fmaf(dz, t, cam.origin[2]);  // :165
      }
      const int e = ((REFR && SLOTS) ? 0 : n_bounce) + lanes_below(mask);
      q.at(0, e) = px; q.at(1, e) = py; q.at(2, e) = pz;
      q.at(3, e) = dx; q.at(4, e) = dy; q.at(5, e) = dz;
      q.at(6, e) = 1.f; q.at(7, e) = 1.f; q.at(8, e) = 1.f;          // T = 1
      if constexpr (!REFR) { q.at(9, e) = 0.f; q.at(10, e) = 0.f; q.at(11, e) = 0.f; }   // acc = 0
      if constexpr (SLOTS) {
        q.at(kFMeta, e) = __uint_as_float(pack_meta_slot(0, slotpix, idx));
      } else {
        q.at(kFPixel, e) = __uint_as_float(uint32_t(row) * uint32_t(p.nx) + uint32_t(col));
        q.at(kFMeta, e) = __uint_as_float(pack_meta(0, frame, idx));
      }
      q.at(kFKey, e) = __uint_as_float(pixel_key);
    }
    n_bounce += __popcll(mask);
  };
  // MULTI: the record of sample `sample` of the tile in slot k is requested into the slot (36 lanes x 4 bytes,
  // LDS-direct): it lands while the slot's current sample is shaded
  auto request_next = [&](int k, uint32_t m0, uint32_t m1, int sample) {
    const int lane = lane_here();
    const RenderParams& p = params_here();
    const size_t id = size_t(m1) * size_t(p.tile_pitch * p.tiles8_y) + size_t(m0 >> 16) * size_t(p.tile_pitch) + size_t(m0 & 0xFFFFu);
    if (lane < kVisTile / 2 && sample < p.n_samples)   // (the second test only keeps a damaged header from becoming a stray load)
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(p.vis + size_t(sample) * size_t(p.vis_sample_stride) + id * kVisTile + lane * 2),
          (__attribute__((address_space(3))) void*)(slot_next + k * kVisTile), 4, 0, 0);
  };

  // Slots whose count has reached zero: the tile leaves whole (render :318, :325 -> store_tile8), or — MULTI — its next
  // sample comes in at the top of the main loop.
  auto flush_slots = [&](uint32_t complete) {
    const int fl = lane_here();
    const RenderParams& p = params_here();
    // (the lane's share of a tile store — row, dword, source lanes — is worked out here, from a lane number
    // the compiler cannot trace: hoisted out of the main loop it would sit in six registers across lighting(),
    // which has none to spare, and go to scratch and back every pass)
    do {
      const int k = __ffs(int(complete)) - 1;
      complete &= complete - 1u;
      uint32_t rgbx;
      if constexpr (MULTI) {
        // a sample of the tile is finished: the next one comes in (top of the loop), or, after the last,
        // the mean over samples is quantised (render_pixel :297, render :318, :325)
        if ((__builtin_amdgcn_readfirstlane(slot_hdr[kSlotHdr * k + 2]) | __builtin_amdgcn_readfirstlane(slot_hdr[kSlotHdr * k + 3])) != 0u) {
          slot_ready |= 1u << k;
          continue;
        }
      }
      if constexpr (REFR) {          // fixed-point sums over the nodes of every sample's tree
        const uint32_t* sum = slot_fix + k * 192 + fl;
        const float unit = p.inv_samples * (1.0f / kSumScale);
        rgbx = pack_rgbx(float(sum[0]) * unit, float(sum[64]) * unit, float(sum[128]) * unit);
      } else if constexpr (MULTI) {
        const float* sum = slot_sum + k * 192 + fl;
        rgbx = pack_rgbx(sum[0] * p.inv_samples, sum[64] * p.inv_samples, sum[128] * p.inv_samples);
      } else {
        rgbx = slot_stage[k * 64 + fl];
      }
      slot_free |= 1u << k;
      const uint32_t m0 = __builtin_amdgcn_readfirstlane(slot_hdr[kSlotHdr * k]);
      const uint32_t frame = __builtin_amdgcn_readfirstlane(slot_hdr[kSlotHdr * k + 1]);
      const int col0 = int((m0 & 0xFFFFu) * 8u), row0 = int((m0 >> 16) * 8u);
      uint8_t* out = p.out_rgb + (long long)frame * p.frame_stride_px * 3;
      const bool dwords = (p.nx & 3) == 0 && (reinterpret_cast<uintptr_t>(out) & 3) == 0;
      store_tile8(out, p.nx, row0, col0, min(8, p.n_rows - row0), min(8, p.nx - col0), fl, rgbx, dwords);
    } while (complete != 0u);
    wave_lds_fence();   // the slots' words are read before an intake reuses them
  };

  for (;;) {
    if constexpr (REFR) {
      // fewer than a pass's worth of records in LDS and a block set aside: it comes back, on top of the queue
      // (positions below 128 <= H, like a tile's fresh hits), before any new tile is taken in
      if ((n_bounce < 64 && sp_blocks > 0)) {
        --sp_blocks;
        queue_take_back<kCap, kFields>(q.f, sp_block(sp_blocks), n_bounce + lane_here());
        n_bounce += 64;
        wave_lds_fence();
        continue;
      }
    }
    // ------------------------------ tile intake ------------------------------------------
    if constexpr (MULTI) {
      // slots whose sample is finished take in their next one first (they hold LDS that new tiles wait for)
      while (n_bounce < 64 && slot_ready != 0u) {
        const int lane = lane_here();
        const int k = __ffs(int(slot_ready)) - 1;
        slot_ready &= ~(1u << k);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the record requested when the previous sample came in
        const int idx = slot_next[k * kVisTile + lane];
        const unsigned long long mask = __ballot(idx >= 0);
        const uint32_t m0 = __builtin_amdgcn_readfirstlane(slot_hdr[kSlotHdr * k]);
        const uint32_t m1 = __builtin_amdgcn_readfirstlane(slot_hdr[kSlotHdr * k + 1]);
        unsigned long long todo = ((unsigned long long)uint32_t(__builtin_amdgcn_readfirstlane(slot_hdr[kSlotHdr * k + 3])) << 32) |
                                  (unsigned long long)uint32_t(__builtin_amdgcn_readfirstlane(slot_hdr[kSlotHdr * k + 2]));
        const int sample = __ffsll((long long)todo) - 1;
        todo &= todo - 1ull;
        wave_lds_fence();                                  // the record is in registers before the next one may land on it
        if (lane == 0) { slot_hdr[kSlotHdr * k + 2] = uint32_t(todo); slot_hdr[kSlotHdr * k + 3] = uint32_t(todo >> 32); }
        if (todo != 0ull) request_next(k, m0, m1, __ffsll((long long)todo) - 1);
        take_in(idx, mask, m0, m1, sample, k);
        wave_lds_fence();
      }
    }
    // (SLOTS: a tile needs a free slot; with none left the wave shades what it has — a pass frees slots)
    if (n_bounce < 64 && !done && (!SLOTS || slot_free != 0u)) {
      const unsigned long long s0 = RT_NOW();
      const int lane = lane_here();
      // Take tiles from the ring until a full pass is available, then refill the ring.  Those loads
      // (and the next claim's atomic) fly during the coming pass, so the waits below are normally
      // free; only when the ring runs empty inside one intake (a run of kRing tiles with few hits)
      // is a load's latency exposed.
      bool consuming = true;
      for (;;) {
        if (head == tail || !consuming) {
          // ---- refill: one tile per turn while the ring has room.  When the current claim will not
          // outlast this refill the next one is requested FIRST, so its round trip overlaps the
          // tile loads issued below; it is read back at the end of the refill (the compiler would
          // wait for it there anyway: it does not leave a returning atomic's register alone across
          // the pass).
          const unsigned long long t0 = RT_NOW();
          const RenderParams& p = params_here();
          for (;;) {
            bool asked = false;
            if (!nx_valid && !homeless && __popc(cl_mask) < kRing) {
              pf_n = claim_size(hcur, seen);
              if (lane == 0) pf_c = atomicAdd(p.chunk_counter + hcur * 32, pf_n);
              asked = true;
            }
            bool stalled = false;              // the ring has room but the next claim has not landed
            while (tail - head < uint32_t(kRing) && !supply_done) {
              if (cl_mask == 0u) {
                if (nx_valid) {
                  const unsigned long long bits = (((unsigned long long)nx_w1 << 32) | nx_w0) >> nx_shift;
                  cl_mask = uint32_t(bits) & (nx_n >= 32 ? 0xFFFFFFFFu : ((1u << nx_n) - 1u));
                  cl_base = nx_id;
                  nx_valid = false;
                  continue;
                }
                if (homeless) {
                  // look (coherently) at all counters and move in with the first one, at or after the
                  // old home, that still has tiles; claims there go on through the same pipeline
                  int next = total_tiles, pos = 0;
                  if (lane < kTileCounters) {
                    pos = __hip_atomic_load(p.chunk_counter + lane * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) +
                          first_of(lane);
                    next = pos * kTileCounters + lane;
                  }
                  const uint32_t avail = uint32_t(__ballot(next < total_tiles)) & ~dry;
                  if (avail == 0u) { supply_done = true; break; }
                  const uint32_t from = avail >> hcur;
                  hcur = from ? hcur + __ffs(int(from)) - 1 : __ffs(int(avail)) - 1;
                  seen = __builtin_amdgcn_readlane(pos, hcur);
                  homeless = false;
                }
                stalled = true;                // ask (again) at the top of the refill loop
                break;
              }
              // next occupied slot of the claim (slots beyond the last tile have no bit set)
              const int cl_id = cl_base + (__ffs(int(cl_mask)) - 1) * kTileCounters;
              cl_mask &= cl_mask - 1u;
              // ---- issue the tile: 36 lanes x 4 bytes (indices + header) straight into ring[slot]
              const uint32_t slot = tail & uint32_t(kRing - 1);
              if (lane < kVisTile / 2)
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)(p.vis + (size_t)cl_id * kVisTile + lane * 2),
                    (__attribute__((address_space(3))) void*)(ring + slot * kVisTile), 4, 0, 0);
              ++tail;
            }
            if (asked) {                       // read the claim back
              const int slot = first_of(hcur) + __builtin_amdgcn_readfirstlane(pf_c);
              seen = slot + pf_n;
              if (slot * kTileCounters + hcur < total_tiles) {
                nx_valid = true; nx_n = pf_n; nx_id = slot * kTileCounters + hcur; nx_shift = slot & 31;
                cocc_t w = as_const(p.occ) + size_t(hcur) * size_t(p.occ_words) + (slot >> 5);
                nx_w0 = w[0]; nx_w1 = w[1];
              }
              else { homeless = true; dry |= 1u << hcur; }
            }
            if (!stalled) break;               // else: go round again with the claim that has just landed
          }
          st_topup += RT_NOW() - t0;
          if (!consuming) break;
          if (head == tail) { done = true; break; }   // nothing left anywhere
        }
        // ---- consume one tile
        const unsigned long long w0 = RT_NOW();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // LDS-direct loads are not tracked by the compiler
        st_wait += RT_NOW() - w0;
        const uint32_t slot = head & uint32_t(kRing - 1);
        ++head;
        const int idx = ring[slot * kVisTile + lane];
        const unsigned long long mask = __ballot(idx >= 0);
        if (mask != 0ull || MULTI) {                         // else: empty tile (sky)
          const uint2 m = *reinterpret_cast<const uint2*>(ring + slot * kVisTile + 64);   // the record's header
          const uint32_t m0 = __builtin_amdgcn_readfirstlane(m.x), m1 = __builtin_amdgcn_readfirstlane(m.y);
          int k = 0, sample = -1;
          if constexpr (SLOTS) {
            // the tile's slot: every pixel black until its record finishes (the misses stay so)
            k = __ffs(int(slot_free)) - 1;
            slot_free &= ~(1u << k);
            if constexpr (MULTI) {
              // (this is sample 0's record; its header also says which samples of the tile hit anything)
              const uint2 sm = *reinterpret_cast<const uint2*>(ring + slot * kVisTile + 68);
              // (readfirstlane returns a signed int: without the casts bit 31 would smear over the upper word)
              unsigned long long todo = ((unsigned long long)uint32_t(__builtin_amdgcn_readfirstlane(sm.y)) << 32) |
                                        (unsigned long long)uint32_t(__builtin_amdgcn_readfirstlane(sm.x));
              slot_sum[k * 192 + lane] = 0.f; slot_sum[k * 192 + 64 + lane] = 0.f; slot_sum[k * 192 + 128 + lane] = 0.f;
              if (todo & 1ull) sample = 0;
              todo &= ~1ull;
              if (lane == 0) {
                slot_hdr[kSlotHdr * k] = m0; slot_hdr[kSlotHdr * k + 1] = m1;
                slot_hdr[kSlotHdr * k + 2] = uint32_t(todo); slot_hdr[kSlotHdr * k + 3] = uint32_t(todo >> 32);
              }
              if (todo != 0ull) request_next(k, m0, m1, __ffsll((long long)todo) - 1);
              if (sample < 0) slot_ready |= 1u << k;           // nothing in sample 0: the next one can come in at once
            } else {
              if constexpr (SUMS) { slot_fix[k * 192 + lane] = 0u; slot_fix[k * 192 + 64 + lane] = 0u; slot_fix[k * 192 + 128 + lane] = 0u; }
              else slot_stage[k * 64 + lane] = 0u;
              if (lane == 0) { slot_hdr[kSlotHdr * k] = m0; slot_hdr[kSlotHdr * k + 1] = m1; }
            }
          }
          if (!MULTI || sample >= 0) take_in(idx, mask, m0, m1, sample, k);
        }
        if (n_bounce >= 64 || (SLOTS && slot_free == 0u)) consuming = false;   // one more turn: refill, then shade
      }
      wave_lds_fence();
      st_fetch += RT_NOW() - s0;
      // (straight on into the pass: the intake ends only with a pass's worth of records, without a free slot, or with
      // nothing left anywhere — going round the loop first came to the same decisions through ~130 instructions of
      // scalar-register spill traffic at the loop head)
    }
    if (n_bounce == 0) {
      // (an intake may have left nothing to shade but a slot whose first sample was empty: its next sample comes in at
      // the top of the loop; likewise a block that a refractive wave has set aside)
      if (MULTI && slot_ready != 0u) continue;
      if (REFR && sp_blocks > 0) continue;
      break;
    }

    // ------------------------------ shading pass -----------------------------------------
    {
      const unsigned long long s0 = RT_NOW();
      const int lane = lane_here();
      int nb = min(64, n_bounce);                                   // the top records of the queue: lanes [0, nb)
      if constexpr (REFR) {
        // A lane may push TWO records, so the queue can grow; it holds kCap records.  Records are taken
        // from its top (deepest first) and a pass is sized from the levels found there: a record at
        // max_depth pushes nothing, any other is counted as two pushes.  The pass takes the longest
        // run of top records that either leaves no more than H = kCap - max_depth records behind, or
        // contains at most one pushing record.  The second case is a depth-first walk below that
        // record (one more record per level at most), which keeps every record's position below
        // H + its level: the queue never holds more than H + max_depth = kCap records.  (Fresh level-0
        // records are put at positions below 128 <= H by the intake, which runs only below 64 records.)
        //   That rule alone degenerates on scenes whose trees really branch (glass cubes: every hit has two
        // children): the queue fills with pushing records and the passes shrink to one or two lanes.  So
        // when the run is cut short for want of room, the BOTTOM 64 records (the shallowest) are set aside first,
        // in a per-wave stack of blocks in global memory (12 coalesced 256-byte stores), and the rest
        // moves down; a block comes back when fewer than 64 records are left (top of the loop).  Lane l
        // touches positions l, 64 + l and 128 + l only, so the move needs no ordering between lanes.
        // Positions only decrease, so the bound above still holds; when the stack is full (spill_blocks)
        // the rule above alone keeps the queue within kCap.
        static_assert(kCap >= (REFR == 1 ? kRefrShallowDepth : RT_MAX_DEPTH) + 2 * 64, "a tile's fresh hits must fit below the high-water mark");
        static_assert(kCap <= 3 * 64, "the move below handles three rows of 64 records");
        const RenderParams& p = params_here();
        const int high = kCap - p.max_depth;
        if (nb > 0) {
          const bool mine = lane < nb;                               // lane l looks at the l-th record from the top
          const uint32_t mt = mine ? __float_as_uint(q.at(kFMeta, n_bounce - 1 - lane)) : 0u;
          const bool pushing = mine && meta_level(mt) < p.max_depth;
          const unsigned long long pm = __ballot(pushing);
          const int upto = lanes_below(pm) + (pushing ? 1 : 0);      // pushing records among the top l + 1
          auto run = [&](int n) {                                    // length of the top run of a queue of n records
            const bool fits = mine && (n - (lane + 1) + 2 * upto <= high || upto <= 1);
            return 64 - __clzll((long long)__ballot(fits));          // (bit 0 is always set)
          };
          const int want = nb;
          nb = run(n_bounce);
          if ((nb + 8 <= want && n_bounce >= 128 && sp_blocks < p.spill_blocks)) {   // cut short for want of room
            queue_set_aside<kCap, kFields>(q.f, sp_block(sp_blocks), lane, n_bounce);
            ++sp_blocks;
            n_bounce -= 64;                                          // (still >= 64: the top `want` = 64 records are the same ones)
            wave_lds_fence();
            nb = run(n_bounce);
          }
        }
      }
      const int base = n_bounce - nb;
      const bool have = lane < nb;
      // Lane l works on record slot e = base + l.  Only what the lighting loops need comes into
      // registers now (P, D, meta, key); T, acc and the pixel stay in the slot until after them.
      int e = base + lane;
      float px = 0.f, py = 0.f, pz = 0.f, dx = 0.f, dy = 0.f, dz = 0.f;
      uint32_t meta = 0, pixel_key = 0;
      if (lane < nb) {
        px = q.at(0, e); py = q.at(1, e); pz = q.at(2, e);
        dx = q.at(3, e); dy = q.at(4, e); dz = q.at(5, e);
        meta = __float_as_uint(q.at(kFMeta, e));
        pixel_key = __float_as_uint(q.at(kFKey, e));
      }
      wave_lds_fence();
      const unsigned long long s1 = RT_NOW();
      unsigned long long s2 = s1;
      bool again = false;
      float tr = 1.f, tg = 1.f, tb = 1.f, ar = 0.f, ag = 0.f, ab = 0.f;
      uint32_t wpix = 0;
      auto late_state = [&]() {
        tr = q.at(6, e); tg = q.at(7, e); tb = q.at(8, e);
        if constexpr (!REFR) { ar = q.at(9, e); ag = q.at(10, e); ab = q.at(11, e); }
        if constexpr (!SLOTS) wpix = __float_as_uint(q.at(kFPixel, e));
        pixel_key = __float_as_uint(q.at(kFKey, e));
      };
      // The material of the object a lane shades is fetched BEHIND lighting(), like T and acc (it is needed
      // only to weight the sums the shadow rays leave): loaded ahead of the shadow loops it held five registers
      // across them, which have none to spare — it went to scratch and came back, a load of the same latency.
      struct LateFromRecord {
        const RenderParams& p;
        decltype(late_state)& state;
        const Queue<kCap>& q;
        int& e;
        uint32_t& meta;
        int base, lane_now;
        __device__ __forceinline__ void request(int) {}
        __device__ __forceinline__ void load(float4& m, float& spec) {
          e = base + lane_now;                                   // (a fresh lane number: see lane_here)
          meta = __float_as_uint(q.at(kFMeta, e));
          const int idx = meta_object(meta);
          m = p.mat[idx];
          spec = p.spec[idx];
          state();
        }
      };
      if constexpr (REFR) {
        // ---- refraction in scope (raytracer.py:213-230, :252-254): a hit has up to two children.  With
        //   C = ((m (1-r) + r C_refl) (1-f) + f C_refr) * l
        // a node adds T m (1-r)(1-f) l to its PIXEL (float atomics into `accum`: the branches of a
        // pixel end in different passes), and hands T r (1-f) l to the reflected, T f l to the
        // refracted ray.  Records carry no acc here; pixels are resolved by rt_resolve_kernel.
        bool push_r = false, push_f = false;
        float rpx = 0.f, rpy = 0.f, rpz = 0.f, rdx = 0.f, rdy = 0.f, rdz = 0.f, rtr = 0.f, rtg = 0.f, rtb = 0.f;
        float fpx = 0.f, fpy = 0.f, fpz = 0.f, fdx = 0.f, fdy = 0.f, fdz = 0.f, ftr = 0.f, ftg = 0.f, ftb = 0.f;
        uint32_t meta_r = 0u, meta_f = 0u;
        if (have) {
          ++n_shade;
          float nx, ny, nz, sx, sy, sz, lr, lg, lb;
          float4 m;
          float2 rf;
          {
            const RenderParams& p = params_here();                   // the lights and tables of lighting(): dead after it
            const int idx = meta_object(meta);
            surface_normal<CUBES>(sg, p.n_spheres, idx, px, py, pz, nx, ny, nz);   // :139-140 / :142-150
            sx = fmaf(nx, 0.001f, px); sy = fmaf(ny, 0.001f, py); sz = fmaf(nz, 0.001f, pz);
            const LightSums lw = lighting<MODE, (CUBES ? kRefrCubeShadowBatch : kRefrShadowBatch), true, CUBES>(p.L, sg, p.geom, p.refr, as_const(p.glass_bits), p.occl + size_t(idx) * size_t(p.occl_words), nullptr, p.n_spheres, p.n_cubes, px, py, pz, nx, ny, nz, dx, dy, dz,
                                    sx, sy, sz, mix(pixel_key, uint32_t(meta_level(meta) + 1)));
            // (material, record state and the meta word come in BEHIND the shadow loops, as in the opaque path:
            // the 13-ray batch leaves no registers for them)
            float spec;
            LateFromRecord late{p, late_state, q, e, meta, base, lane_here()};
            late.load(m, spec);
            rf = p.refr[meta_object(meta)];
            light_colour(p.L, lw, m.x, m.y, m.z, spec, lr, lg, lb);
          }
          const RenderParams& p = params_here();
          const int level = meta_level(meta);
          [[maybe_unused]] const int rframe = SLOTS ? 0 : meta_frame(meta);
          // the bits of the meta word that the children keep: the pixel's slot and position, or its frame
          const uint32_t meta_home = SLOTS ? (meta & 0x0000FF80u) : (uint32_t(rframe) << 7);
          s2 = RT_NOW();
          const float refl = m.w, refr = rf.x;
          const float own = (1.0f - refl) * (1.0f - refr);                   // :249, :254 (1 - 0 = 1 exactly)
          if constexpr (SLOTS) {
            // the node's share joins its pixel's sums in the tile's slot: fixed point, saturating (a wrapped add is
            // followed by a store of the largest value; shares are never negative — the host takes this path only for
            // scenes whose colours, intensities and coefficients make it so —, so a sum that has once saturated stays
            // there and the pixel clamps to white, as the reference's would).  Two records of one pixel in one pass
            // (the two children of a node, usually) add to one address: the LDS serialises them, and integer
            // addition does not care in which order.
            uint32_t* sum = slot_fix + (meta_slotpix(meta) >> 6) * 192 + (meta_slotpix(meta) & 63);
            const uint32_t qr = uint32_t(fmaf(tr * (m.x * own) * lr, kSumScale, 0.5f));
            const uint32_t qg = uint32_t(fmaf(tg * (m.y * own) * lg, kSumScale, 0.5f));
            const uint32_t qb = uint32_t(fmaf(tb * (m.z * own) * lb, kSumScale, 0.5f));
            const uint32_t wr = atomicAdd(sum, qr), wg = atomicAdd(sum + 64, qg), wb = atomicAdd(sum + 128, qb);
            if (wr + qr < wr) atomicMax(sum, 0xFFFFFFFFu);
            if (wg + qg < wg) atomicMax(sum + 64, 0xFFFFFFFFu);
            if (wb + qb < wb) atomicMax(sum + 128, 0xFFFFFFFFu);
          } else {
            float* a = p.accum + ((long long)rframe * p.accum_stride_px + wpix) * 3;
            unsafeAtomicAdd(a + 0, tr * (m.x * own) * lr);
            unsafeAtomicAdd(a + 1, tg * (m.y * own) * lg);
            unsafeAtomicAdd(a + 2, tb * (m.z * own) * lb);
          }
          const bool deeper = level < p.max_depth;                           // :233 `depth > D` -> black
          if (refl > 0.0f && deeper) {                                       // :245-248
            const float dn2 = 2.0f * fmaf(dz, nz, fmaf(dy, ny, dx * nx));
            rdx = fmaf(-dn2, nx, dx); rdy = fmaf(-dn2, ny, dy); rdz = fmaf(-dn2, nz, dz);
            float t;
            int idx2, face;
            ++n_trace;
            if (!CUBES && p.refl_dim > 0) { nearest_table(p, sg, meta_object(meta), sx, sy, sz, rdx, rdy, rdz, t, idx2); face = 0; }
            else nearest_object<CUBES>(sg, p.geom, p.n_spheres, p.n_cubes, sx, sy, sz, rdx, rdy, rdz, t, idx2, face);
            if (idx2 >= 0) {
              push_r = true;
              hit_point<CUBES>(sg, p.n_spheres, idx2, face, sx, sy, sz, rdx, rdy, rdz, t, rpx, rpy, rpz);
              const float k = refl * (1.0f - refr);
              rtr = tr * k * lr; rtg = tg * k * lg; rtb = tb * k * lb;
              meta_r = meta_home | uint32_t(level + 1) | (uint32_t(idx2) << 20);
            }
          }
          if (refr > 0.0f && deeper) {                                       // :252-253 -> compute_refraction
            float n1 = 1.0f, n2 = rf.y;                                      // :214-215
            float cos_i = -fmaf(nz, dz, fmaf(ny, dy, nx * dx));              // :217
            float qx = nx, qy = ny, qz = nz;
            if (cos_i < 0.0f) {                                              // :218-221
              cos_i = -cos_i; qx = -qx; qy = -qy; qz = -qz;
              const float tmp = n1; n1 = n2; n2 = tmp;
            }
            const float n = n1 / n2;                                         // :223
            const float k = 1.0f - n * n * (1.0f - cos_i * cos_i);           // :224
            if (k >= 0.0f) {                                                 // :225 (else total internal reflection: black)
              const float w = n * cos_i - fsqrt(k);
              fdx = fmaf(qx, w, dx * n); fdy = fmaf(qy, w, dy * n); fdz = fmaf(qz, w, dz * n);   // :228
              const float inv = frsq(fmaf(fdz, fdz, fmaf(fdy, fdy, fdx * fdx)));
              fdx *= inv; fdy *= inv; fdz *= inv;
              const float ox = fmaf(qx, -0.001f, px), oy = fmaf(qy, -0.001f, py), oz = fmaf(qz, -0.001f, pz);  // :229
              float t;
              int idx2, face;
              ++n_trace;
              // (the refracted ray starts 0.001 INSIDE the object: still within reach of the table's bound)
              if (!CUBES && p.refl_dim > 0) { nearest_table(p, sg, meta_object(meta), ox, oy, oz, fdx, fdy, fdz, t, idx2); face = 0; }
              else nearest_object<CUBES>(sg, p.geom, p.n_spheres, p.n_cubes, ox, oy, oz, fdx, fdy, fdz, t, idx2, face);
              if (idx2 >= 0) {
                push_f = true;
                hit_point<CUBES>(sg, p.n_spheres, idx2, face, ox, oy, oz, fdx, fdy, fdz, t, fpx, fpy, fpz);
                ftr = tr * refr * lr; ftg = tg * refr * lg; ftb = tb * refr * lb;
                meta_f = meta_home | uint32_t(level + 1) | (uint32_t(idx2) << 20);
              }
            }
          }
        }
        wave_lds_fence();   // every lane has its whole record in registers before slots are rewritten
        const unsigned long long mask_r = __ballot(push_r), mask_f = __ballot(push_f);
        // (slot < kCap always holds by the pass sizing above; the test only keeps a logic error from
        // becoming a stray LDS write)
        if (push_r && base + lanes_below(mask_r) < kCap) {
          const int slot = base + lanes_below(mask_r);
          q.at(0, slot) = rpx; q.at(1, slot) = rpy; q.at(2, slot) = rpz;
          q.at(3, slot) = rdx; q.at(4, slot) = rdy; q.at(5, slot) = rdz;
          q.at(6, slot) = rtr; q.at(7, slot) = rtg; q.at(8, slot) = rtb;
          if constexpr (!SLOTS) q.at(kFPixel, slot) = __uint_as_float(wpix);
          q.at(kFMeta, slot) = __uint_as_float(meta_r);
          q.at(kFKey, slot) = __uint_as_float(pixel_key);
        }
        if (push_f && base + __popcll(mask_r) + lanes_below(mask_f) < kCap) {
          const int slot = base + __popcll(mask_r) + lanes_below(mask_f);
          q.at(0, slot) = fpx; q.at(1, slot) = fpy; q.at(2, slot) = fpz;
          q.at(3, slot) = fdx; q.at(4, slot) = fdy; q.at(5, slot) = fdz;
          q.at(6, slot) = ftr; q.at(7, slot) = ftg; q.at(8, slot) = ftb;
          if constexpr (!SLOTS) q.at(kFPixel, slot) = __uint_as_float(wpix);
          q.at(kFMeta, slot) = __uint_as_float(meta_f);
          q.at(kFKey, slot) = __uint_as_float(pixel_key);
        }
        n_bounce = min(kCap, base + __popcll(mask_r) + __popcll(mask_f));
        wave_lds_fence();
        if constexpr (SLOTS) {
          // records of each slot's trees: this pass's have gone, their children have come; a slot whose count reaches
          // zero has every node of every pixel in its sums (one turn per slot that had records in the pass)
          uint32_t complete = 0u;
          const int fslot = meta_slotpix(meta) >> 6;
          unsigned long long todo = __ballot(have);
          while (todo != 0ull) {
            const int k = __builtin_amdgcn_readlane(fslot, __ffsll((long long)todo) - 1);
            const unsigned long long same = __ballot(have && fslot == k);
            todo &= ~same;
            const long long delta = (long long)(__popcll(mask_r & same) + __popcll(mask_f & same)) - (long long)__popcll(same);
            slot_left += (unsigned long long)(delta << (kCntBits * k));      // (the field stays within 0 .. 4095: no borrow)
            if (((slot_left >> (kCntBits * k)) & ((1ull << kCntBits) - 1ull)) == 0ull) complete |= 1u << k;
          }
          if (complete != 0u) flush_slots(complete);
        }
        const unsigned long long s3 = RT_NOW();
        st_pop += s1 - s0; st_light += s2 - s1; st_near += s3 - s2;
#if RT_STAMPS
        st_passes += 1; st_lanes += unsigned(nb); st_starved += (nb < 64 && !done) ? 1u : 0u;
#endif
        continue;
      }
      if (have) {
        ++n_shade;
        float ox, oy, oz;
        bool cont;
        {
          const RenderParams& p = params_here();                     // the lights and tables of lighting(): dead after it
          LateFromRecord late{p, late_state, q, e, meta, base, lane_here()};
          const uint32_t level_key = mix(pixel_key, uint32_t(meta_level(meta) + 1));
          cont = shade_hit<MODE, CUBES, (GRID != 0)>(p, sg, meta_object(meta), px, py, pz, dx, dy, dz, level_key,
                                        tr, tg, tb, ar, ag, ab, ox, oy, oz, late);
        }
        const RenderParams& p = params_here();
        const int level = meta_level(meta);                          // (the meta word as re-read behind lighting())
        cont = cont && (level < p.max_depth);                        // :233 `depth > D` -> black
        s2 = RT_NOW();
        if (cont) {
          float t;
          int idx2, face;
          ++n_trace;
          if constexpr (GRID != 0) {
            if (p.list_dim > 0) nearest_list(p, sg, meta_object(meta), ox, oy, oz, dx, dy, dz, t, idx2);
            else nearest_hit(sg, p.geom, p.n_spheres, ox, oy, oz, dx, dy, dz, t, idx2);   // (RT_OPT_REFLECT_TABLE off, or lists beyond 256 MB)
            face = 0;
          }
          else if (!CUBES && p.refl_dim > 0) { nearest_table(p, sg, meta_object(meta), ox, oy, oz, dx, dy, dz, t, idx2); face = 0; }
          else nearest_object<CUBES>(sg, p.geom, p.n_spheres, p.n_cubes, ox, oy, oz, dx, dy, dz, t, idx2, face);
          if (idx2 >= 0) {
            again = true;
            hit_point<CUBES>(sg, p.n_spheres, idx2, face, ox, oy, oz, dx, dy, dz, t, px, py, pz);
            // (the frame / slot and pixel bits of the meta word stay; level and object are replaced)
            meta = (meta & (SLOTS ? 0x0000FF80u : 0x000FFF80u)) | uint32_t(level + 1) | (uint32_t(idx2) << 20);
          }
        }
        if (!again) {
          if constexpr (MULTI) {
            // color = color + trace_ray(...) (:296): this sample joins the pixel's sum in its tile's slot (no other
            // record of the pixel is in flight: a slot takes its samples in one after another)
            const int sp = meta_slotpix(meta);
            float* sum = slot_sum + (sp >> 6) * 192 + (sp & 63);
            sum[0] += ar; sum[64] += ag; sum[128] += ab;
          } else if constexpr (SLOTS) {
            // the pixel waits in its tile's slot; the tile leaves when its last pixel is in (below)
            slot_stage[meta_slotpix(meta)] = pack_rgbx(ar, ag, ab);
          } else {
            const int rframe = meta_frame(meta);
            finish_pixel(p, (long long)rframe * p.frame_stride_px + wpix,
                         (long long)rframe * p.accum_stride_px + wpix, ar, ag, ab, nullptr);
          }
        }
      }
      wave_lds_fence();   // every lane has its whole record in registers before slots are rewritten
      const unsigned long long mask = __ballot(again);
      if (again) {
        const int slot = base + lanes_below(mask);
        q.at(0, slot) = px; q.at(1, slot) = py; q.at(2, slot) = pz;
        q.at(3, slot) = dx; q.at(4, slot) = dy; q.at(5, slot) = dz;
        q.at(6, slot) = tr; q.at(7, slot) = tg; q.at(8, slot) = tb;
        q.at(9, slot) = ar; q.at(10, slot) = ag; q.at(11, slot) = ab;
        if constexpr (!SLOTS) q.at(kFPixel, slot) = __uint_as_float(wpix);
        q.at(kFMeta, slot) = __uint_as_float(meta);
        q.at(kFKey, slot) = __uint_as_float(pixel_key);
      }
      n_bounce = base + __popcll(mask);
      wave_lds_fence();
      if constexpr (SLOTS) {
        // pixels that finished in this pass, counted per slot: one turn per slot that has some (one to three)
        uint32_t complete = 0u;
        {
          const int fslot = meta_slotpix(meta) >> 6;
          unsigned long long fin = __ballot(have && !again);
          while (fin != 0ull) {
            const int k = __builtin_amdgcn_readlane(fslot, __ffsll((long long)fin) - 1);
            const unsigned long long same = __ballot(have && !again && fslot == k);
            fin &= ~same;
            slot_left -= (unsigned long long)__popcll(same) << (kCntBits * k);
            if (((slot_left >> (kCntBits * k)) & ((1ull << kCntBits) - 1ull)) == 0ull) complete |= 1u << k;
          }
        }
        // tiles whose last hit pixel finished in this pass leave as whole tiles (render :318, :325)
        if (complete != 0u) flush_slots(complete);
      }
      const unsigned long long s3 = RT_NOW();
      st_pop += s1 - s0; st_light += s2 - s1; st_near += s3 - s2;
#if RT_STAMPS
      st_passes += 1; st_lanes += unsigned(nb); st_starved += (nb < 64 && !done) ? 1u : 0u;
#endif
    }
  }

  // ---- ray statistics (per wave; primaries are the visibility pass's) -----------------------------
  const RenderParams& p = params_here();
  const int lane = lane_here();
  if (p.counters) {
    unsigned long long tot_t = n_trace, tot_s = n_shade;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      tot_t += __shfl_xor(tot_t, o);
      tot_s += __shfl_xor(tot_s, o);
    }
    if (lane == 0) {
      unsigned long long* slot = p.counters + size_t(gwave % kCounterSlots) * kCounterWords;
      if (tot_t) atomicAdd(slot + 0, tot_t);
      if (tot_s) {
        atomicAdd(slot + 1, tot_s);
        atomicAdd(slot + 3, tot_s * (unsigned long long)(1 + p.L.ns));
      }
#if RT_STAMPS
      // per-wave phase sums (diagnostic build only; read by tools/stamps.py)
      atomicMax(slot + 8, RT_NOW() - st_begin); atomicAdd(slot + 9, st_fetch); atomicAdd(slot + 10, st_pop);
      atomicAdd(slot + 11, st_light); atomicAdd(slot + 12, st_near); atomicAdd(slot + 13, st_topup);
      atomicAdd(slot + 7, st_wait);
      atomicAdd(slot + 14, RT_NOW() - st_begin); atomicAdd(slot + 15, 1ull);
      // [4] passes, [5] records shaded in them, [6] passes run short for want of a free tile slot (or of a room in the REFR queue)
      atomicAdd(slot + 4, (unsigned long long)st_passes); atomicAdd(slot + 5, (unsigned long long)st_lanes);
      atomicAdd(slot + 6, (unsigned long long)st_starved);
#endif
    }
  }
}

// ------------------------------------------------------------------------------------------
// PIXEL KERNEL (rt_set_option(RT_OPT_KERNEL, RT_KERNEL_PIXEL)): one thread per pixel, the whole
// path in one loop, no compaction.  Same arithmetic as the stream kernel; kept as the baseline
// for the divergence / active-lane counters (profiles/).
// ------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(kBlock, RT_WAVES_PER_SIMD) void rt_pixel_kernel(const RenderParams p) {
  extern __shared__ __align__(16) unsigned char smem[];
  float4* sg = reinterpret_cast<float4*>(smem);
  const int geom_bytes = ((p.n_spheres * 16 + 15) / 16) * 16;
  uint32_t* stage = reinterpret_cast<uint32_t*>(smem + geom_bytes);            // [8][32]
  unsigned int* s_cnt = reinterpret_cast<unsigned int*>(stage + kTileW * kTileH);

  const int tid = threadIdx.x;
  for (int i = tid; i < p.n_spheres; i += kBlock) sg[i] = p.geom[i];
  if (tid < 2) s_cnt[tid] = 0u;
  __syncthreads();

  const int tiles_x = (p.nx + kTileW - 1) / kTileW, tiles_y = (p.n_rows + kTileH - 1) / kTileH;
  const int frame = blockIdx.x / (tiles_x * tiles_y);
  const int tile = blockIdx.x - frame * (tiles_x * tiles_y);
  const int tile_y = tile / tiles_x, tile_x = tile - tile_y * tiles_x;
  const int wave = tid >> 6, lane = tid & 63;
  const int lx = wave * 8 + (lane & 7), ly = lane >> 3;
  const int col0 = tile_x * kTileW, row0 = tile_y * kTileH;
  const int cols = min(kTileW, p.nx - col0), rows = min(kTileH, p.n_rows - row0);
  const rt_camera& cam = p.frames[frame].cam;
  const uint32_t seed32 = p.frames[frame].seed;
  const long long frame_px = (long long)frame * p.frame_stride_px;
  const long long accum_px = (long long)frame * p.accum_stride_px;

  uint32_t n_trace = 0, n_shade = 0;
  if (lx < cols && ly < rows) {
    const int x = p.x0 + col0 + lx, y = row_of(p, row0 + ly);
    uint32_t pixel_key;
    float dx, dy, dz;
    primary_ray<MODE>(p, cam, seed32, x, y, p.sample_index, pixel_key, dx, dy, dz);
    float ox = cam.origin[0], oy = cam.origin[1], oz = cam.origin[2];
    float ar = 0.f, ag = 0.f, ab = 0.f, tr = 1.f, tg = 1.f, tb = 1.f;
    const long long gpix = frame_px + (long long)(row0 + ly) * p.nx + (col0 + lx);
    for (int level = 0; level <= p.max_depth; ++level) {               // :233
      float t;
      int idx;
      ++n_trace;
      if (level == 0)
        primary_hit(p.geom, p.camtab + size_t(frame) * cam_groups_stride(p.n_spheres), p.n_spheres, ox, oy, oz, dx, dy,
                    dz, t, idx);
      else
        nearest_hit(sg, p.geom, p.n_spheres, ox, oy, oz, dx, dy, dz, t, idx);
      if (level == 0 && p.out_hit && p.sample_index + 1 == p.n_samples) p.out_hit[gpix] = idx;
      if (idx < 0) break;                                              // :237-238
      ++n_shade;
      const float px = fmaf(dx, t, ox), py = fmaf(dy, t, oy), pz = fmaf(dz, t, oz);
      MaterialInRegisters late{p};
      if (!shade_hit<MODE>(p, sg, idx, px, py, pz, dx, dy, dz, mix(pixel_key, uint32_t(level + 1)),
                           tr, tg, tb, ar, ag, ab, ox, oy, oz, late))
        break;
    }
    finish_pixel(p, gpix, gpix - frame_px + accum_px, ar, ag, ab, &stage[ly * kTileW + lx]);
  }
  if (n_trace) atomicAdd(&s_cnt[0], n_trace);
  if (n_shade) atomicAdd(&s_cnt[1], n_shade);
  __syncthreads();
  if (p.out_rgb && p.sample_index + 1 == p.n_samples)
    store_tile_rgb<kTileW>(p.out_rgb + frame_px * 3, stage, p.nx, row0, col0, rows, cols, tid);
  if (tid == 0 && p.counters) flush_counters(p, s_cnt, blockIdx.x, (unsigned long long)rows * cols);
}

// out[y] = gathered[rank of y][local row of y]  (rt_scatter_rows)
__global__ __launch_bounds__(256) void rt_scatter_rows_kernel(const uint8_t* __restrict__ gathered,
                                                              long long rank_stride, int n_ranks,
                                                              int width, int height, int tile_rows,
                                                              uint8_t* __restrict__ out, int vec4,
                                                              long long frame_stride_in, long long frame_stride_out) {
  gathered += (long long)blockIdx.z * frame_stride_in;      // batch of frames: one per grid z
  out += (long long)blockIdx.z * frame_stride_out;
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

// ------------------------------------------------------------------------------------------
// Multi-GPU exchange: only the tiles of a rank's slab that hold something travel.
//
// Sender (rt_compact_tiles): the visibility pass has already marked the 8x8 tiles of the batch in which some
// pixel hit something (`occ`, one bit per tile id); every other tile is black.  One lane looks at one tile id;
// a wave reserves room for its occupied tiles with one atomic and copies each as an entry
//     { uint32 tile id, 8 rows x 24 bytes of packed RGB }  = kTileEntryBytes
// (48 lanes x one dword where rows are dword-aligned).  The number of entries stays on the device: no rank's
// host waits for its render.  SCAN: no occupancy bits at hand (the batch took several launches, or samples were
// separate launches): a tile is kept if some byte of it is not zero.
// ------------------------------------------------------------------------------------------
constexpr int kTileEntryBytes = RT_TILE_ENTRY_BYTES;
static_assert(kTileEntryBytes == 4 + 8 * 24, "entry = tile id + 8 rows of 8 RGB pixels");
struct TileGeom {          // the slab of one rank: n_frames windows of n_rows x nx pixels, frame_stride_px apart
  int nx, n_rows, n_frames, tiles8_x, tiles8_y, tile_pitch;
  long long frame_stride_px;
};
template <bool SCAN>
__global__ __launch_bounds__(256) void rt_compact_tiles_kernel(const uint8_t* __restrict__ slab, TileGeom g,
                                                               const uint32_t* __restrict__ occ, int occ_words,
                                                               uint8_t* __restrict__ entries, unsigned int* __restrict__ count,
                                                               unsigned int capacity) {
  const int lane = threadIdx.x & 63;
  const long long total = (long long)g.tile_pitch * g.tiles8_y * g.n_frames;
  const long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int per_frame = g.tile_pitch * g.tiles8_y;
  bool keep = false;
  int frame = 0, ty = 0, tx = 0;
  if (id < total) {
    frame = int(id / per_frame);
    const int rem = int(id - (long long)frame * per_frame);
    ty = rem / g.tile_pitch; tx = rem - ty * g.tile_pitch;
    if (tx < g.tiles8_x) {                                   // (the odd pitch leaves padding ids without a tile)
      if constexpr (SCAN) {
        const int rows = min(8, g.n_rows - ty * 8), nbytes = min(8, g.nx - tx * 8) * 3;
        const uint8_t* src = slab + ((long long)frame * g.frame_stride_px + (long long)(ty * 8) * g.nx + tx * 8) * 3;
        unsigned int any = 0u;
        for (int r = 0; r < rows; ++r)
          for (int b = 0; b < nbytes; ++b) any |= src[(long long)r * g.nx * 3 + b];
        keep = any != 0u;
      } else {
        const uint32_t slot = uint32_t(id) / kTileCounters, h = uint32_t(id) % kTileCounters;
        keep = (occ[size_t(h) * size_t(occ_words) + (slot >> 5)] >> (slot & 31u)) & 1u;
      }
    }
  }
  const unsigned long long kept = __ballot(keep);
  if (kept == 0ull) return;
  unsigned int base = 0u;
  if (lane == 0) base = atomicAdd(count, (unsigned int)__popcll(kept));
  base = __builtin_amdgcn_readfirstlane(base);
  const bool dwords = (g.nx & 3) == 0 && (g.frame_stride_px & 3) == 0 && (reinterpret_cast<uintptr_t>(slab) & 3) == 0;
  unsigned long long todo = kept;
  int k = 0;
  while (todo != 0ull) {                                     // the wave copies its kept tiles one after another
    const int src_lane = __ffsll((long long)todo) - 1;
    todo &= todo - 1ull;
    const unsigned int slot = base + unsigned(k++);
    if (slot >= capacity) continue;                          // (the count still says how many there were: the host checks)
    const int f = __builtin_amdgcn_readlane(frame, src_lane), y0 = __builtin_amdgcn_readlane(ty, src_lane) * 8,
              x0 = __builtin_amdgcn_readlane(tx, src_lane) * 8;
    const unsigned int tid = unsigned(__builtin_amdgcn_readlane(int(uint32_t(id)), src_lane));
    uint8_t* e = entries + size_t(slot) * kTileEntryBytes;
    if (lane == 0) *reinterpret_cast<uint32_t*>(e) = tid;
    const uint8_t* src = slab + ((long long)f * g.frame_stride_px + (long long)y0 * g.nx + x0) * 3;
    const int rows = min(8, g.n_rows - y0), cols = min(8, g.nx - x0);
    if (dwords && cols == 8) {
      const int r = (lane * 43) >> 8, w = lane - 6 * r;     // lane l < 48: dword w of tile row r
      if (lane < 48)
        reinterpret_cast<uint32_t*>(e + 4)[lane] =
            r < rows ? *reinterpret_cast<const uint32_t*>(src + (long long)r * g.nx * 3 + w * 4) : 0u;
    } else {
      for (int b = lane; b < 192; b += 64) {
        const int r = b / 24, c = b - 24 * r;
        e[4 + b] = (r < rows && c < cols * 3) ? src[(long long)r * g.nx * 3 + c] : uint8_t(0);
      }
    }
  }
}

// Root, step 1: where is the entry of a tile?  table[rank][tile id] = entry number + 1 (0 = none: the tile is black)
__global__ __launch_bounds__(256) void rt_index_tiles_kernel(const uint8_t* __restrict__ entries, const unsigned int* __restrict__ counts,
                                                             int capacity, long long ids_per_rank, uint32_t* __restrict__ table) {
  const int rank = blockIdx.y;
  const unsigned int n = min(counts[rank], (unsigned int)capacity);
  for (unsigned int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
    const uint32_t id = *reinterpret_cast<const uint32_t*>(entries + (size_t(rank) * capacity + k) * kTileEntryBytes);
    if ((long long)id < ids_per_rank) table[(long long)rank * ids_per_rank + id] = k + 1u;
  }
}

// Root, step 2: every row of every frame, written once — black where no entry covers it.  Image row y belongs to
// row tile t = y / tile_rows of rank t mod n_ranks and is that rank's local row (t / n_ranks) tile_rows + y mod
// tile_rows (rt_render_rows); the 24 bytes of an 8-pixel segment come from row (local row mod 8) of the entry of
// the rank's tile (local row / 8, x / 8).  A workgroup writes a band of 8 image rows of one frame, a thread one
// dword of a row at a time (bytes when rows are not dword-aligned): consecutive lanes, consecutive addresses.
__global__ __launch_bounds__(256) void rt_place_tiles_kernel(const uint8_t* __restrict__ entries, const uint32_t* __restrict__ table,
                                                             int capacity, long long ids_per_rank, int n_ranks, int width, int height,
                                                             int tile_rows, uint8_t* __restrict__ out, int vec4) {
  const int frame = blockIdx.y;
  const int tiles8_x = (width + 7) / 8, pitch = tiles8_x | 1;
  const int n_tiles = (height + tile_rows - 1) / tile_rows;
  const int row_bytes = width * 3;
  for (int y = int(blockIdx.x) * 8; y < min(height, int(blockIdx.x) * 8 + 8); ++y) {
    const int t = y / tile_rows, rank = t % n_ranks;
    const int lr = (t / n_ranks) * tile_rows + (y - t * tile_rows);
    // rows of this rank's window (its tile ids count ITS tile rows): row tiles rank, rank + n_ranks, ... of the frame
    const int mine = n_tiles > rank ? (n_tiles - rank + n_ranks - 1) / n_ranks : 0;
    const int rows_r = mine * tile_rows - (((n_tiles - 1) % n_ranks == rank) ? n_tiles * tile_rows - height : 0);
    const int tiles8_y = (rows_r + 7) / 8;
    const uint32_t* trow = table + (long long)rank * ids_per_rank + (long long)frame * pitch * tiles8_y + (long long)(lr >> 3) * pitch;
    const uint8_t* erow = entries + size_t(rank) * capacity * kTileEntryBytes + 4 + (lr & 7) * 24;
    uint8_t* dst = out + ((long long)frame * height + y) * row_bytes;
    if (vec4) {
      for (int w = threadIdx.x; w < row_bytes / 4; w += 256) {
        const int tx = w / 6, dw = w - 6 * tx;
        const uint32_t slot = trow[tx];
        reinterpret_cast<uint32_t*>(dst)[w] =
            slot ? *reinterpret_cast<const uint32_t*>(erow + size_t(slot - 1u) * kTileEntryBytes + dw * 4) : 0u;
      }
    } else {
      for (int b = threadIdx.x; b < row_bytes; b += 256) {
        const int tx = b / 24, c = b - 24 * tx;
        const uint32_t slot = trow[tx];
        dst[b] = slot ? erow[size_t(slot - 1u) * kTileEntryBytes + c] : uint8_t(0);
      }
    }
  }
}

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
  int refl_dim = 0, refl_pairs = 0;
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
  uint32_t* d_occl = nullptr;    // occluder table: [objects][ceil(n_spheres / 32)] (rt_upload_scene, rt_set_cubes)
  float4* d_cubes = nullptr;     // [n_cubes][2] min, max (rt_set_cubes); objects n_spheres.. of d_mat / d_spec / d_refr
  int n_cubes = 0;
  unsigned long long* d_counters = nullptr;
  FrameRec* d_frames = nullptr;  // [kMaxBatch] cameras and seeds of the batch in flight
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

constexpr int kGridFrom = 256;   // scenes with at least this many spheres: shadow candidates and direction-table entries as LISTS (the list kernels)
// Direction table for nearest_table(): bit j of entry [k][bucket] says that SOME ray which starts within rho_k =
// r_k + 0.001 (+ margin) of K and whose direction lies in the bucket can hit sphere j.  Such a ray reaches a point
// X of sphere j from an origin O:  X - O = (J - K) + e  with |e| <= rho_k + r_j, so its direction is within
// theta_kj = asin((rho_k + r_j) / |J - K|) of u = (J - K) / |J - K|  (any direction if the balls touch).  A bucket is
// a square patch of a cube face; the angle to a fixed axis is largest at a corner of the patch, so the patch lies in
// the cone (axis through its centre, half-angle h to the farthest corner), and j is listed iff
// angle(axis, u) <= theta_kj + h + 1e-3 (the last term: the kernel's FP32 bucket arithmetic, ~1e-6).  k itself is
// never listed: a ray that leaves a sphere outwards cannot come back to it.
constexpr int kReflTabFrom = RT_REFL_TABLE_FROM;   // scenes with at least this many spheres (and fewer than the grid's threshold)
constexpr size_t kReflTabMaxBytes = size_t(1) << 20;
int reflect_table_dim(int n) {
  if (n < kReflTabFrom || n >= kGridFrom) return 0;
  const size_t pairs = size_t((n + 63) / 64);
  for (int B : {16, 8}) if (size_t(n) * size_t(6 * B * B) * pairs * 8 <= kReflTabMaxBytes) return B;
  return 4;
}
struct DirPatch { double x, y, z, ch, sh, h; };   // bounding cone of a bucket: unit axis, cos / sin of its half-angle h (margin included)
std::vector<DirPatch> direction_patches(int B) {
  std::vector<DirPatch> patch(size_t(6 * B * B), DirPatch{0, 0, 0, 0, 0, 0});
  auto dir = [](int f, double a, double b, double* v) {   // the kernel's convention: face 2 m + (major < 0), a = next axis, b = the one after
    const int m = f >> 1;
    v[m] = (f & 1) ? -1.0 : 1.0; v[(m + 1) % 3] = a; v[(m + 2) % 3] = b;
    const double l = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    v[0] /= l; v[1] /= l; v[2] /= l;
  };
  for (int f = 0; f < 6; ++f)
    for (int iy = 0; iy < B; ++iy)
      for (int ix = 0; ix < B; ++ix) {
        const double a0 = -1.0 + 2.0 * ix / B, a1 = -1.0 + 2.0 * (ix + 1) / B, b0 = -1.0 + 2.0 * iy / B, b1 = -1.0 + 2.0 * (iy + 1) / B;
        double c[3], v[3], cmin = 1.0;
        dir(f, 0.5 * (a0 + a1), 0.5 * (b0 + b1), c);
        for (int k = 0; k < 4; ++k) {
          dir(f, (k & 1) ? a1 : a0, (k & 2) ? b1 : b0, v);
          cmin = std::min(cmin, c[0] * v[0] + c[1] * v[1] + c[2] * v[2]);
        }
        const double h = std::acos(std::max(-1.0, std::min(1.0, cmin))) + 1e-3;
        patch[size_t((f * B + iy) * B + ix)] = DirPatch{c[0], c[1], c[2], std::cos(h), std::sin(h), h};
      }
  return patch;
}
// The buckets of object k, sphere by sphere: emit(bucket, j) for every bucket that lists j, j ascending.  Buckets are
// looked at through a coarse level (4 x 4 patches per face, each the union of (B/4)^2 buckets) when B allows it:
// most spheres reach into one or two coarse patches only.
template <typename Emit>
void reflect_buckets_of(const float* geom4, int n, int k, int B, const std::vector<DirPatch>& fine,
                        const std::vector<DirPatch>& coarse, Emit&& emit) {
  const double kPi = 3.14159265358979323846;
  const int nb = 6 * B * B, Bc = coarse.empty() ? 0 : 4, sub = Bc ? B / Bc : 0;
  const double kx = geom4[4 * k], ky = geom4[4 * k + 1], kz = geom4[4 * k + 2];
  const double rho = std::sqrt(double(geom4[4 * k + 3])) * 1.001 + 2e-3;
  for (int j = 0; j < n; ++j) {
    if (j == k) continue;
    const double vx = geom4[4 * j] - kx, vy = geom4[4 * j + 1] - ky, vz = geom4[4 * j + 2] - kz;
    const double d = std::sqrt(vx * vx + vy * vy + vz * vz);
    const double reach = rho + std::sqrt(double(geom4[4 * j + 3])) * 1.001 + 1e-3;
    const bool all = !(d > reach * 1.0001);
    const double sn = all ? 1.0 : reach / d, theta = all ? kPi : std::asin(sn), cs = std::sqrt(std::max(0.0, 1.0 - sn * sn));
    const double ux = all ? 0.0 : vx / d, uy = all ? 0.0 : vy / d, uz = all ? 0.0 : vz / d;
    // angle(axis, u) <= theta + h  <=>  axis.u >= cos(theta + h)   (theta + h < pi)
    auto reaches = [&](const DirPatch& P) {
      return all || theta + P.h >= kPi || P.x * ux + P.y * uy + P.z * uz >= cs * P.ch - sn * P.sh;
    };
    if (!Bc) {
      for (int b = 0; b < nb; ++b) if (reaches(fine[size_t(b)])) emit(b, j);
      continue;
    }
    for (int f = 0; f < 6; ++f)
      for (int cy = 0; cy < Bc; ++cy)
        for (int cx = 0; cx < Bc; ++cx) {
          if (!reaches(coarse[size_t((f * Bc + cy) * Bc + cx)])) continue;
          for (int iy = cy * sub; iy < (cy + 1) * sub; ++iy)
            for (int ix = cx * sub; ix < (cx + 1) * sub; ++ix) {
              const int b = (f * B + iy) * B + ix;
              if (reaches(fine[size_t(b)])) emit(b, j);
            }
        }
  }
}
void build_reflect_table(const float* geom4, int n, int B, std::vector<uint2>& tab) {
  const int pairs = (n + 63) / 64, nb = 6 * B * B;
  tab.assign(size_t(n) * size_t(nb) * size_t(pairs), make_uint2(0u, 0u));
  const std::vector<DirPatch> fine = direction_patches(B);
  const std::vector<DirPatch> coarse = (B > 4 && B % 4 == 0) ? direction_patches(4) : std::vector<DirPatch>();
  for (int k = 0; k < n; ++k)
    reflect_buckets_of(geom4, n, k, B, fine, coarse, [&](int b, int j) {
      uint2& e = tab[(size_t(k) * size_t(nb) + size_t(b)) * size_t(pairs) + size_t(j >> 6)];
      if (j & 32) e.y |= 1u << (j & 31); else e.x |= 1u << (j & 31);
    });
}
// The table as lists (nearest_list): start[k * nb + b] .. start[k * nb + b + 1] are the 4-index chunks of entry
// [k][b] in `items` (ascending sphere indices, the last chunk padded with 0xFFFF).  Objects are independent, so the
// build runs on several threads.  Returns false — no table, reflection rays scan every sphere — if the lists would exceed
// kReflListMaxChunks (a scene where most spheres reach most directions of most objects: huge or interlocked spheres).
constexpr int kReflListDim = RT_REFL_LIST_DIM;
constexpr size_t kReflListMaxChunks = size_t(32) << 20;   // 256 MB of indices
bool build_reflect_lists(const float* geom4, int n, int B, std::vector<uint32_t>& start, std::vector<uint2>& items) {
  const int nb = 6 * B * B;
  const std::vector<DirPatch> fine = direction_patches(B);
  const std::vector<DirPatch> coarse = (B > 4 && B % 4 == 0) ? direction_patches(4) : std::vector<DirPatch>();
  std::vector<std::vector<uint32_t>> counts{static_cast<size_t>(n)};      // per object: chunks of each bucket
  std::vector<std::vector<uint2>> chunks{static_cast<size_t>(n)};        // per object: its buckets' chunks, bucket after bucket
  std::atomic<int> next(0);
  std::atomic<size_t> total(0);
  auto work = [&]() {
    std::vector<std::vector<uint16_t>> rows{static_cast<size_t>(nb)};
    for (;;) {
      const int k = next.fetch_add(1);
      if (k >= n || total.load() > kReflListMaxChunks) return;
      for (auto& r : rows) r.clear();
      reflect_buckets_of(geom4, n, k, B, fine, coarse, [&](int b, int j) { rows[size_t(b)].push_back(uint16_t(j)); });
      std::vector<uint32_t>& cnt = counts[size_t(k)];
      std::vector<uint2>& out = chunks[size_t(k)];
      cnt.assign(size_t(nb), 0u);
      for (int b = 0; b < nb; ++b) {
        const std::vector<uint16_t>& r = rows[size_t(b)];
        cnt[size_t(b)] = uint32_t((r.size() + 3) / 4);
        for (size_t i = 0; i < r.size(); i += 4) {
          uint32_t v[4];
          for (size_t q = 0; q < 4; ++q) v[q] = i + q < r.size() ? r[i + q] : 0xFFFFu;
          out.push_back(make_uint2(v[0] | (v[1] << 16), v[2] | (v[3] << 16)));
        }
      }
      total.fetch_add(out.size());
    }
  };
  const unsigned hw = std::thread::hardware_concurrency();
  const int n_threads = int(std::max(1u, std::min(16u, std::min(hw ? hw : 1u, unsigned(n / 64 + 1)))));
  std::vector<std::thread> pool;
  for (int t = 1; t < n_threads; ++t) pool.emplace_back(work);
  work();
  for (auto& t : pool) t.join();
  if (total.load() > kReflListMaxChunks) return false;
  start.assign(size_t(n) * size_t(nb) + 1, 0u);
  items.clear();
  items.reserve(total.load() + 1);
  size_t at = 0;
  for (int k = 0; k < n; ++k) {
    for (int b = 0; b < nb; ++b) {
      start[size_t(k) * size_t(nb) + size_t(b)] = uint32_t(at);
      at += counts[size_t(k)][size_t(b)];
    }
    items.insert(items.end(), chunks[size_t(k)].begin(), chunks[size_t(k)].end());
  }
  start.back() = uint32_t(at);
  items.push_back(make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu));   // (never empty: hipMalloc of zero bytes)
  return true;
}

// Occluder table for lighting()'s shadow scan: bit c of row k says that sphere c can be hit by SOME
// shadow ray that leaves sphere k (raytracer.py:184, :199: origins lie 0.001 off its surface) towards
// the point light or any point of the area light.  All such rays start inside the ball (K, rho) and
// pass through the ball (A, R) of the light, so they stay inside the cone that contains the first
// ball and has half-angle asin((rho + R) / |A - K|) about K -> A; a sphere matters only if it reaches
// into that cone.  FP64, with margins far above the kernel's FP32 error: a superset, and the kernel
// still applies its per-point cull and the exact test to every candidate, so results do not change.
void build_occluder_table(const float* geom4, int n, const DevLights& L, std::vector<uint32_t>& tab,
                          bool selective = true) {
  const int words = (n + 31) / 32;
  tab.assign(size_t(std::max(n, 1)) * size_t(std::max(words, 1)), 0u);
  struct Ball { double x, y, z, r; };
  const Ball lights[2] = {{L.ppos[0], L.ppos[1], L.ppos[2], 1e-4},
                          {L.apos[0], L.apos[1], L.apos[2], double(L.arad) * 1.001 + 1e-4}};
  for (int k = 0; k < n; ++k) {
    const double kx = geom4[4 * k], ky = geom4[4 * k + 1], kz = geom4[4 * k + 2];
    const double rho = std::sqrt(double(geom4[4 * k + 3])) * 1.001 + 2e-3;
    uint32_t* row = tab.data() + size_t(k) * size_t(words);
    bool all = !selective;        // RT_OPT_OCCLUDER_TABLE off (verification runs): every sphere is a candidate
    for (const Ball& A : lights) {
      if (all) break;
      const double ax = A.x - kx, ay = A.y - ky, az = A.z - kz;
      const double d = std::sqrt(ax * ax + ay * ay + az * az);
      if (!(d > (rho + A.r) * 1.01)) { all = true; break; }    // the light is (nearly) inside the ball: rays go anywhere
      const double ux = ax / d, uy = ay / d, uz = az / d;
      const double sn = (rho + A.r) / d, cs = std::sqrt(1.0 - sn * sn);
      const double back = rho / sn;                            // apex of the cone that contains the ball (K, rho)
      const double px = kx - ux * back, py = ky - uy * back, pz = kz - uz * back;
      for (int c = 0; c < n; ++c) {
        const double cx = geom4[4 * c] - px, cy = geom4[4 * c + 1] - py, cz = geom4[4 * c + 2] - pz;
        const double d2 = cx * cx + cy * cy + cz * cz;
        const double along = cx * ux + cy * uy + cz * uz;
        const double perp = std::sqrt(std::max(0.0, d2 - along * along));
        const double rc = std::sqrt(double(geom4[4 * c + 3]));
        if (perp * cs - along * sn <= rc * 1.001 + 1e-3 * (1.0 + std::sqrt(d2))) row[c >> 5] |= 1u << (c & 31);
      }
    }
    if (all)
      for (int c = 0; c < n; ++c) row[c >> 5] |= 1u << (c & 31);
  }
}

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
  p.occl = ctx->d_occl; p.occl_words = (ctx->n_spheres + 31) / 32;
  p.occl_start = ctx->d_occl_start; p.occl_items = ctx->d_occl_items;
  p.glass_bits = ctx->d_glass_bits;
  p.list_start = ctx->d_list_start; p.list_items = ctx->d_list_items;
  p.list_dim = (refr || cubes || !stream_kernel) ? 0 : ctx->list_dim;
  p.refl_tab = ctx->d_refl_tab; p.refl_pairs = ctx->refl_pairs;
  p.refl_dim = (cubes || !stream_kernel || listed) ? 0 : ctx->refl_dim;
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

template <typename K>
hipError_t allow_big_lds(K kernel, size_t bytes) {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                             hipFuncAttributeMaxDynamicSharedMemorySize, int(bytes));
}

}  // namespace

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
  if ((e = hipMalloc(&ctx->d_occl, sizeof(uint32_t) * RT_MAX_SPHERES * (RT_MAX_SPHERES / 32))) != hipSuccess)
    return bail(e, "hipMalloc");
  if ((e = hipMalloc(&ctx->d_occl_start, sizeof(uint32_t) * (RT_MAX_SPHERES + 1))) != hipSuccess) return bail(e, "hipMalloc");
  if ((e = hipMalloc(&ctx->d_glass_bits, sizeof(uint32_t) * (RT_MAX_SPHERES / 32))) != hipSuccess) return bail(e, "hipMalloc");
  if ((e = hipMemset(ctx->d_glass_bits, 0, sizeof(uint32_t) * (RT_MAX_SPHERES / 32))) != hipSuccess) return bail(e, "hipMemset");
  if ((e = hipMalloc(&ctx->d_frames, sizeof(FrameRec) * kMaxBatch)) != hipSuccess) return bail(e, "hipMalloc");
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
  L.arad = 0.5f * (std::sqrt(L.au[0] * L.au[0] + L.au[1] * L.au[1] + L.au[2] * L.au[2]) +
                   std::sqrt(L.av[0] * L.av[0] + L.av[1] * L.av[1] + L.av[2] * L.av[2]));
  if (n_spheres > 0) {
    std::vector<uint32_t> occl;
    build_occluder_table(geom4, n_spheres, L, occl, ctx->occluder_table);
    RT_HIP(ctx, hipMemcpy(ctx->d_occl, occl.data(), sizeof(uint32_t) * occl.size(), hipMemcpyHostToDevice));
    if (n_spheres >= kGridFrom) {   // the GRID kernels read the rows as lists (lighting<..., LIST>)
      const int words = (n_spheres + 31) / 32;
      std::vector<uint32_t> start(size_t(n_spheres) + 1, 0u);
      std::vector<uint2> items;
      for (int k = 0; k < n_spheres; ++k) {
        start[size_t(k)] = uint32_t(items.size());
        uint32_t v[4];
        int fill = 0;
        for (int c = 0; c < n_spheres; ++c)
          if ((occl[size_t(k) * size_t(words) + size_t(c >> 5)] >> (c & 31)) & 1u) {
            v[fill++] = uint32_t(c);
            if (fill == 4) { items.push_back(make_uint2(v[0] | (v[1] << 16), v[2] | (v[3] << 16))); fill = 0; }
          }
        if (fill) {
          for (int q = fill; q < 4; ++q) v[q] = 0xFFFFu;
          items.push_back(make_uint2(v[0] | (v[1] << 16), v[2] | (v[3] << 16)));
        }
      }
      start[size_t(n_spheres)] = uint32_t(items.size());
      items.push_back(make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu));
      const size_t ib = items.size() * sizeof(uint2);
      if (ib > ctx->occl_items_bytes) {
        if (ctx->d_occl_items) (void)hipFree(ctx->d_occl_items);
        ctx->d_occl_items = nullptr; ctx->occl_items_bytes = 0;
        RT_HIP(ctx, hipMalloc(&ctx->d_occl_items, ib));
        ctx->occl_items_bytes = ib;
      }
      RT_HIP(ctx, hipMemcpy(ctx->d_occl_start, start.data(), start.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
      RT_HIP(ctx, hipMemcpy(ctx->d_occl_items, items.data(), ib, hipMemcpyHostToDevice));
    }
  }
  ctx->list_dim = 0;
  if (ctx->reflect_table && n_spheres >= kGridFrom) {   // many spheres: the direction table as lists
    std::vector<uint32_t> start;
    std::vector<uint2> items;
    if (build_reflect_lists(geom4, n_spheres, kReflListDim, start, items)) {
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
      ctx->list_dim = kReflListDim;
    }
  }
  {
    const int B = ctx->reflect_table ? reflect_table_dim(n_spheres) : 0;
    ctx->refl_dim = 0;
    if (B > 0) {
      std::vector<uint2> tab;
      build_reflect_table(geom4, n_spheres, B, tab);
      const size_t bytes = tab.size() * sizeof(uint2);
      if (bytes > ctx->refl_tab_bytes) {
        if (ctx->d_refl_tab) (void)hipFree(ctx->d_refl_tab);
        ctx->d_refl_tab = nullptr; ctx->refl_tab_bytes = 0;
        RT_HIP(ctx, hipMalloc(&ctx->d_refl_tab, bytes));
        ctx->refl_tab_bytes = bytes;
      }
      RT_HIP(ctx, hipMemcpy(ctx->d_refl_tab, tab.data(), bytes, hipMemcpyHostToDevice));
      ctx->refl_dim = B; ctx->refl_pairs = (n_spheres + 63) / 64;
    }
  }
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
    // occluder table rows of the cubes: every sphere may shadow a point of a box
    const int words = (ctx->n_spheres + 31) / 32;
    if (words > 0) {
      std::vector<uint32_t> rows(size_t(n_cubes) * size_t(words), 0u);
      for (int i = 0; i < n_cubes; ++i)
        for (int c = 0; c < ctx->n_spheres; ++c) rows[size_t(i) * size_t(words) + size_t(c >> 5)] |= 1u << (c & 31);
      RT_HIP(ctx, hipMemcpy(ctx->d_occl + at * size_t(words), rows.data(), sizeof(uint32_t) * rows.size(),
                            hipMemcpyHostToDevice));
    }
  }
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
  L.arad = 0.5f * (std::sqrt(L.au[0] * L.au[0] + L.au[1] * L.au[1] + L.au[2] * L.au[2]) +
                   std::sqrt(L.av[0] * L.av[0] + L.av[1] * L.av[1] + L.av[2] * L.av[2]));
  std::vector<uint32_t> tab;
  build_occluder_table(geom4, n_spheres, L, tab);
  const size_t words = size_t(n_spheres) * size_t((n_spheres + 31) / 32);
  if (words) memcpy(out_words, tab.data(), words * sizeof(uint32_t));
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
  RT_HIP(ctx, hipSetDevice(ctx->device));
  RT_HIP(ctx, hipMemsetAsync(count_dev, 0, sizeof(uint32_t), ctx->stream));
  if (total == 0) return RT_OK;
  const dim3 grid(unsigned((total + 255) / 256));
  const uint32_t* occ = reinterpret_cast<const uint32_t*>(ctx->d_chunk) + kTileCounters * 32;
  if (ctx->last.occ_valid)
    hipLaunchKernelGGL(rt_compact_tiles_kernel<false>, grid, dim3(256), 0, ctx->stream, slab, g, occ, ctx->last.occ_words,
                       entries, count_dev, (unsigned int)capacity_tiles);
  else
    hipLaunchKernelGGL(rt_compact_tiles_kernel<true>, grid, dim3(256), 0, ctx->stream, slab, g, (const uint32_t*)nullptr, 0,
                       entries, count_dev, (unsigned int)capacity_tiles);
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
  const int vec4 = (row_bytes % 4 == 0) && (reinterpret_cast<uintptr_t>(out_rgb) % 4 == 0) && (reinterpret_cast<uintptr_t>(entries) % 4 == 0);
  hipLaunchKernelGGL(rt_place_tiles_kernel, dim3(unsigned((height + 7) / 8), unsigned(n_frames)),
                     dim3(256), 0, ctx->stream, entries, (const uint32_t*)table, capacity_tiles, ids_per_rank, n_ranks, width, height,
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
