// rt_core.hpp — part of the ONE translation unit rt355.hip (included there, inside its anonymous namespace; not a
// header for anyone else).  Device core of rt355: kernel parameters, keyed RNG, arithmetic helpers, the intersection tests of raytracer.py:93-172
// (spheres, cubes, the table-driven scans of reflection rays) and the per-batch primary-ray tables.
#pragma once

// ------------------------------------------------------------------------------------------
// kernel parameters (passed by value: the kernarg segment is read with scalar loads)
// ------------------------------------------------------------------------------------------
struct DevLights {
  float ppos[3], pcol[3], pint;
  float apos[3], au[3], av[3], acol[3], aint;
  int ns;
  float arad;   // max(|au + av|, |au - av|) / 2: every area-light sample lies within this distance of apos
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
  int occl_dim;          // > 0: rows per (object, bucket of the surface normal): row (k * 6 dim^2 + direction_bucket(N, dim)); 0: row k
  int occl_list_dim;     // the same for the list rows (occl_start)
  // the same table as lists, for scenes with many spheres (the GRID kernels; 32 .. 64 mostly empty words per row otherwise):
  // row k is the run of 4-index chunks occl_start[k] .. occl_start[k + 1] of occl_items (uint16, ascending, padded with 0xFFFF)
  const uint32_t* occl_start;
  const uint2* occl_items;
  // direction table for reflection rays (refl_dim = 0: none): [object][6 refl_dim^2 direction buckets][refl_pairs] 64-bit
  // masks of the spheres a ray that leaves the object in a direction of the bucket can hit (nearest_table)
  const uint2* refl_tab;
  int refl_dim, refl_pairs;
  int refl_cpairs;       // ... followed by this many 64-bit masks of the BOXES such a ray's line can cross (scenes with cubes)
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
#define RT_VIS_WAVES 4     // its __launch_bounds__ occupancy target (waves per SIMD): 108 VGPRs and no spill; at 5 (96 VGPRs, 60 bytes of scratch
                           // since the region culls its blocks jointly) config 2 / target / config 3 / config 1 ran 0.9 / 1.2 / 1.7 / 1.9 % slower
#endif
#ifndef RT_VIS_WAVES_MANY
#define RT_VIS_WAVES_MANY 5   // the many-sphere variant (config 4: 4 -> -2.9 %)
#endif
constexpr int kVisTiles = RT_VIS_TILES;
#ifndef RT_VIS_REGION_BLOCKS
#define RT_VIS_REGION_BLOCKS 1   // the region culls the scene against the cones of all its blocks at once (scenes of up to 64 spheres)
#endif
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
#ifndef RT_LIGHT_ALL
#define RT_LIGHT_ALL 1     // a copy of lighting() for lights that ask for exactly one batch of rays (no per-ray guard)
#endif
#ifndef RT_SHADOW_BATCH
#define RT_SHADOW_BATCH 13
#endif
constexpr int kShadowBatch = RT_SHADOW_BATCH;   // shadow rays traced together: 1 point-light ray + 12 area-light rays (raytracer.py:80) by default
constexpr int kReferenceShadowRays = 13;
#ifndef RT_REFR_BATCH
#define RT_REFR_BATCH 13     // (7 while the kernels were short of registers; see kRefrShadowBatch)
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
#define RT_REFR_CUBE_BATCH 7     // (5 until both verdicts of a ray shared one register: 266 -> 287 Grays/s on cube_scene(glass); 9: 254, 13: 182)
#endif
constexpr int kCubeShadowBatch = RT_CUBE_BATCH;            // scenes with boxes
constexpr int kRefrCubeShadowBatch = RT_REFR_CUBE_BATCH;   // ... and refractive materials
constexpr int kRefrShadowBatch = RT_REFR_BATCH;   // refractive scenes: all 13 rays in one batch since the per-lane ray counters left the vector registers (473 Grays/s on refractive_scene(); 7 + 6: 456, 10 + 3: 401; scenes with boxes stay at 7: 9 -> -10 %, 10 -> -15 %)

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

// (the same with a row0 of the lane's own: a vector division where the rows are not contiguous)
__device__ __forceinline__ int tile_row_base_lane(const RenderParams& p, int row0) {
  if (p.tile_stride == 1) return p.y0 + p.tile_first * p.tile_rows + row0;
  const int j = row0 / p.tile_rows;
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
// Row of the occluder table (bit masks: `dim` = p.occl_dim; lists: p.occl_list_dim) for a hit on object `idx` whose
// unit surface normal is N: rows are kept per bucket of the normal (build_occluder_buckets), or per object (dim = 0).
__device__ __forceinline__ size_t occluder_row(int dim, int idx, float nx, float ny, float nz) {
  return dim > 0 ? size_t(idx) * size_t(6 * dim * dim) + size_t(direction_bucket(nx, ny, nz, dim)) : size_t(idx);
}
template <bool CUBES = false>
__device__ __forceinline__ void nearest_table(const RenderParams& p, const float4* __restrict__ sg, int obj, float ox,
                                              float oy, float oz, float dx, float dy, float dz, float& best_t,
                                              int& best_i, int& best_face) {
  const float inv_a = frcp(fmaf(dz, dz, fmaf(dy, dy, dx * dx)));
  best_t = INFINITY;
  best_i = -1;
  best_face = 0;
  const int B = p.refl_dim;
  const int bucket = direction_bucket(dx, dy, dz, B);
  const uint2* row = p.refl_tab + (size_t(obj) * size_t(6 * B * B) + size_t(bucket)) * size_t(p.refl_pairs + (CUBES ? p.refl_cpairs : 0));
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
  if constexpr (CUBES) {
    // the boxes the entry lists, after the spheres and in index order (nearest_object's order: same winner on ties)
    const float ix = frcp(dx), iy = frcp(dy), iz = frcp(dz);
    const int n = p.n_spheres;
    for (int w = 0; w < p.refl_cpairs; ++w) {
      const uint2 mask = row[p.refl_pairs + w];
      uint32_t lo = mask.x, hi = mask.y;
      while (__ballot((lo | hi) != 0u) != 0ull) {
        if ((lo | hi) != 0u) {
          const bool in_lo = lo != 0u;
          uint32_t cur = in_lo ? lo : hi;
          const int c = 64 * w + (in_lo ? 0 : 32) + __ffs(int(cur)) - 1;
          cur &= cur - 1u;
          if (in_lo) lo = cur; else hi = cur;
          float t;
          int face;
          if (cube_t(sg[n + 2 * c], sg[n + 2 * c + 1], ox, oy, oz, dx, dy, dz, ix, iy, iz, t, face) && t < best_t) {
            best_t = t; best_i = n + c; best_face = face;
          }
        }
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
                                                   int (&best_i)[NP], uint32_t kept_chunks = 0u) {
  float inv_a[NP];
#pragma unroll
  for (int q = 0; q < NP; ++q) {
    best_t[q] = INFINITY;
    best_i[q] = -1;
    inv_a[q] = frcp(fmaf(dz[q], dz[q], fmaf(dy[q], dy[q], dx[q] * dx[q])));
  }
  camtab_t tab = as_const(camtab);
  cgeom_t geom = as_const(gg);
  // (`kept`: only the chunks that hold a kept sphere — bit c of kept_chunks —, in increasing order as the plain walk)
  for (int base = kept ? (kept_chunks ? (__ffs(int(kept_chunks)) - 1) * 64 : n) : 0; base < n;
       base = kept ? ((kept_chunks &= kept_chunks - 1u) ? (__ffs(int(kept_chunks)) - 1) * 64 : n) : base + 64) {
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
