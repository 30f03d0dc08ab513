/* rt355 — C ABI of the MI355X-native sphere ray tracer (librt355.so).
 *
 * This is the drop-in boundary for the per-pixel render path of aram-ap/RayTracer's
 * raytracer.py.  The reference has no FFI: its boundary is the Python function
 * render(width, height, samples) (raytracer.py:310-325) reading the module-global `scene`
 * (raytracer.py:78-87).  The Python host in raytracer_amd/ keeps that entry point and calls
 * the functions below through ctypes; INTEGRATION.md shows the binding a maintainer of the
 * reference would add.
 *
 * Conventions
 *   - plain C types only; every function returns 0 on success and a negative RT_E* code on
 *     failure, with a message retrievable by rt_last_error(); no C++ exception crosses the ABI.
 *   - the caller owns every buffer.  `out_*` pointers are DEVICE pointers on the context's GPU
 *     (e.g. torch.Tensor.data_ptr()); scene/camera/light inputs are HOST pointers.
 *   - launches are asynchronous on the context's stream (rt_set_stream); rt_synchronize() or
 *     the caller's own stream sync completes them.  One host thread per context.
 *   - the library never falls back to a CPU path: without a gfx950 GPU rt_create fails.
 */
#ifndef RT355_H
#define RT355_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_MAX_SPHERES 2048      /* 16 B of LDS per sphere in the shading kernels: 32 KiB      */
#define RT_MAX_AREA_SAMPLES 64   /* AreaLight.samples (raytracer.py:69); the reference uses 12 */
#define RT_MAX_DEPTH 64          /* upper bound accepted for max_depth                         */
#define RT_FRAMES_PER_LAUNCH 1024 /* frames of a batch that share one kernel launch pair       */
#define RT_NOALLOC_SCRATCH_BYTES (24 << 20) /* rt_render_noalloc: device scratch per GPU (frames up to 4K) */

#define RT_OK 0
#define RT_EINVAL (-1)   /* bad argument (null pointer, size out of range, misaligned)          */
#define RT_EDEVICE (-2)  /* no usable GPU / HIP runtime error                                   */
#define RT_ESTATE (-3)   /* call order (e.g. render before rt_upload_scene)                     */

/* RNG replacement for random.random() (raytracer.py:180-181, 293-294).  The reference draws
 * from Python's global Mersenne Twister in call order, which no parallel renderer can
 * reproduce; both modes are pure functions of position. */
#define RT_RNG_CONSTANT 0 /* every draw is 0.5: pixel-centre rays, area light at its centre     */
#define RT_RNG_KEYED 1    /* hash of (seed, x, y, sample, level, draw) — see DESIGN.md §RNG      */

typedef struct rt_ctx rt_ctx;

/* Pinhole camera.  The reference hard-codes origin 0, direction normalize((u, v, -1))
 * (raytracer.py:293-295); this is that with right=(1,0,0), up=(0,1,0), forward=(0,0,-1). */
typedef struct rt_camera {
  float origin[3];
  float right[3];
  float up[3];
  float forward[3];
} rt_camera;

/* scene['global_light'] (Light, raytracer.py:62-66, :79) and scene['area_light']
 * (AreaLight, raytracer.py:68-75, :80). */
typedef struct rt_lights {
  float point_pos[3];
  float point_color[3];
  float point_intensity;
  float area_pos[3];
  float area_u[3];
  float area_v[3];
  float area_color[3];
  float area_intensity;
  int32_t area_samples;
} rt_lights;

/* Create / destroy a context bound to one GPU (one process per GPU; multi-GPU runs create one
 * context per rank).  Replaces nothing in the reference (it has no device). */
int rt_create(int device_id, rt_ctx** out);
void rt_destroy(rt_ctx* ctx);

/* Message for the last failure on this context (ctx may be NULL for rt_create failures). */
const char* rt_last_error(const rt_ctx* ctx);

/* Select the stream every later call launches on: the caller's hipStream_t passed as void*
 * (NULL is HIP's legacy default stream, which is what torch.cuda.current_stream().cuda_stream
 * is by default), or, with use_own_stream != 0, the context's own non-blocking stream (the
 * state after rt_create). */
int rt_set_stream(rt_ctx* ctx, void* hip_stream, int use_own_stream);
int rt_synchronize(rt_ctx* ctx);

/* Tuning / diagnostic options (no reference counterpart).
 *   RT_OPT_KERNEL: RT_KERNEL_STREAM (default) = visibility pass (primary rays) followed by the
 *   persistent-wave shading kernel that compacts hits into dense 64-lane passes; RT_KERNEL_PIXEL = one
 *   thread per pixel, whole path in one loop (kept as the baseline for divergence counters).  Both
 *   compute the same arithmetic per ray.
 *   RT_OPT_OCCLUDER_TABLE: 1 (default) = the shadow scan of a hit on object k only looks at the spheres
 *   that can shadow some point of k (see rt_occluder_table); 0 = at every sphere, as the reference
 *   does.  Images are byte-identical either way (tests); takes effect at the next rt_upload_scene.
 *   RT_OPT_QUEUE_SPILL: 1 (default) = on refractive scenes a wave of the stream kernel whose work queue
 *   cannot take a full pass's children sets its 64 shallowest records aside in device memory (<= 32 blocks of
 *   3 KB per wave, allocated on first use) instead of shrinking the pass; 0 = shrink only; 2..32 = at most
 *   that many blocks per wave (tests: a wave whose stack is full shrinks its passes again).  Same bytes either
 *   way (tests): the pixel sums are integers.
 *   RT_OPT_TILE_FLUSH: 1 (default) = the stream kernel stages the finished pixels of a tile in LDS and stores the
 *   tile whole (8 x 24 contiguous bytes, whole dwords) when its last pixel is in — the coalesced packed-RGB store
 *   of the frame (raytracer.py:318, :325); 0 = every finished pixel is stored by itself (three byte stores: one
 *   memory request per pixel).  Same bytes either way on opaque scenes (tests).  The tile slots also carry the
 *   one-launch path for 2 .. 64 samples per pixel and the fixed-point pixel sums of refractive scenes (see
 *   rt_set_refraction); with 0 those fall back to float sums in device memory.
 *   RT_OPT_REFLECT_TABLE: 1 (default) = reflection (and refraction) rays of scenes with 4 or more spheres and no
 *   cubes look only at the spheres a table built at scene upload lists for (object they leave, direction bucket) —
 *   bit masks up to 255 spheres (see rt_reflect_table), lists from 256 on; 0 = the plain scan of
 *   find_nearest_intersection (raytracer.py:152-172).  Same images (tests); takes effect at the next rt_upload_scene.
 *   (Option 3, a uniform grid for those rays, was removed in round 3: the lists replaced it.) */
#define RT_OPT_KERNEL 1
#define RT_OPT_OCCLUDER_TABLE 2
#define RT_OPT_QUEUE_SPILL 4
#define RT_OPT_TILE_FLUSH 5
#define RT_OPT_REFLECT_TABLE 6
#define RT_KERNEL_STREAM 0
#define RT_KERNEL_PIXEL 1
int rt_set_option(rt_ctx* ctx, int option, int value);

/* Upload scene['spheres'] as struct-of-arrays plus the two lights.
 * Replaces the reference's AoS Python objects Sphere/Material (raytracer.py:42-54) and the
 * `scene` dict (raytracer.py:78-87).
 *   geom4[n][4] = cx, cy, cz, radius^2      mat4[n][4] = R, G, B, reflection     spec[n] = specular
 * All host pointers; 0 <= n <= RT_MAX_SPHERES; 1 <= area_samples <= RT_MAX_AREA_SAMPLES. */
int rt_upload_scene(rt_ctx* ctx, const float* geom4, const float* mat4, const float* spec,
                    int n_spheres, const rt_lights* lights);

/* Material.refraction / Material.refractive_index (raytracer.py:43, :213-230, :252-254) of the spheres
 * of the uploaded scene, n_spheres values each (host pointers).  rt_upload_scene resets them to
 * Material's defaults (0, 1); call this after it.  While any refraction is > 0 the renderer follows
 * both children of a hit, a shadow ray is blocked only by an opaque NEAREST hit (:187, :202), and
 * RT_KERNEL_PIXEL is refused.  A pixel is then the sum over the nodes of a tree; 8-bit frames at up to 64 samples
 * per pixel sum them in 32-bit fixed point (24 fractional bits, saturating at 256) per tile in LDS: the bytes of a
 * frame do not depend on the run, the tiling or the batch.  Float output (rt_render_chunk), more than 64 samples,
 * RT_OPT_TILE_FLUSH off, and scenes in which a node's share can be negative (a negative colour component, light
 * colour, intensity, specular or reflection / refraction outside [0, 1]) are summed with float atomics in device
 * memory instead: run-to-run differences at FP32 rounding level. */
int rt_set_refraction(rt_ctx* ctx, const float* refraction, const float* refractive_index, int n_spheres);

/* Host-only helper (needs no GPU, no context): the occluder table rt_upload_scene builds for the shadow
 * scan of compute_lighting (raytracer.py:179-187, :198-202).  Row k (ceil(n/32) words) has bit c set if
 * sphere c can be hit by some shadow ray that leaves sphere k towards either light — a conservative
 * superset computed in FP64; the kernels test only these candidates (and still test them exactly).
 * No counterpart in the reference (which tests every sphere for every ray); exported so that the
 * superset property can be tested against brute force without a GPU.
 * out_words: n * ceil(n/32) uint32. */
int rt_occluder_table(const float* geom4, int n_spheres, const rt_lights* lights, uint32_t* out_words);

/* The table as rt_upload_scene uploads it for scenes of fewer than 256 spheres (lists of the same rows from 256 on): one
 * row per (sphere k, bucket of the unit surface normal at the shaded point) — the cube-map buckets of rt_reflect_table
 * below, applied to the normal.  Bit c of row [k][bucket] is set if sphere c can be hit by some shadow ray that leaves
 * a point of sphere k whose normal lies in the bucket; c = k itself only where such a ray can point back into the
 * sphere (a point that faces both lights cannot be shadowed by its own sphere).  Most rows are empty.  Host-only, FP64,
 * a conservative superset; exported for the brute-force test.  out_words: n * 6 dim^2 * ceil(n/32) uint32; 1 <= dim <= 16. */
int rt_occluder_table_buckets(const float* geom4, int n_spheres, const rt_lights* lights, int dim, uint32_t* out_words);

/* Host-only helper (needs no GPU, no context): the direction table rt_upload_scene builds for the reflection rays
 * of find_nearest_intersection (raytracer.py:152-172, called from trace_ray :246-248).  Directions are bucketed on
 * a cube map: face 2 m + (d[m] < 0) for the major axis m, cell (ix, iy) = floor((d[(m+1)%3] / |d[m]| + 1) dim / 2),
 * floor((d[(m+2)%3] / |d[m]| + 1) dim / 2), bucket = (face * dim + iy) * dim + ix.  Entry [k][bucket] is
 * ceil(n/64) pairs of uint32 (bits 0..31, 32..63 of each 64 spheres); bit j set = some ray that starts within
 * 0.001 of sphere k's surface with a direction in the bucket can hit sphere j (j != k) — a conservative superset
 * computed in FP64; the kernels test only those spheres (and test them exactly).  No counterpart in the reference
 * (which tests every sphere); exported so that the superset property can be tested against brute force without a
 * GPU.  out_words: n * 6 dim^2 * 2 ceil(n/64) uint32; 1 <= dim <= 16. */
int rt_reflect_table(const float* geom4, int n_spheres, int dim, uint32_t* out_words);

/* Cube objects (raytracer.py:56-60) in the list find_nearest_intersection scans (:156-160, :164-169):
 * intersect_cube (:106-137) and cube_normal (:142-150).  n_cubes boxes, host pointers:
 *   box6[n][6] = min_point xyz, max_point xyz     mat4[n][4] = R, G, B, reflection     spec[n] = specular
 *   refraction[n], refractive_index[n]: Material's, or NULL for the defaults 0 / 1.
 * The cubes become objects n_spheres, n_spheres + 1, ... of the scene (the hit index of out_hit); the
 * host mirror (raytracer_amd.scenes.pack_scene) maps these back to positions in the reference's list.
 * A sphere wins an exact tie in t against a cube here whatever their list order (the reference keeps
 * the earlier list member, :162); such ties have measure zero.  rt_upload_scene resets the scene to no
 * cubes; call this after it.  n_spheres + 2 n_cubes <= RT_MAX_SPHERES.  RT_KERNEL_PIXEL is refused
 * while cubes are present.
 * The reference's behaviour is reproduced as it is, not as a box should look: intersect_cube returns
 * t_max unchecked when t_min <= 0 (:137) and :162 accepts any non-zero t, so a cube is hit by every
 * ray whose LINE passes through it, also behind the origin (t < 0, which then beats every sphere).
 * An opaque cube therefore shadows itself completely and every point whose line to a light crosses
 * it in either direction. */
int rt_set_cubes(rt_ctx* ctx, const float* box6, const float* mat4, const float* spec, const float* refraction,
                 const float* refractive_index, int n_cubes);

/* render(width, height, samples) (raytracer.py:310-325): the whole frame.
 *   out_rgb: device uint8 [height][width][3], row 0 = top, tightly packed — the array the
 *            reference returns and hands to glTexImage2D (raytracer.py:346).
 *   out_hit: device int32 [height][width] primary hit sphere index of the last sample (-1 =
 *            miss), or NULL.
 *   max_depth D: trace_ray's literal `depth > 5` (raytracer.py:233) as a parameter; D = 5 is
 *            the reference. */
int rt_render(rt_ctx* ctx, const rt_camera* cam, int width, int height, int samples,
              int max_depth, int rng_mode, uint64_t seed, uint8_t* out_rgb, int32_t* out_hit);

/* rt_render that allocates nothing.  NO COUNTERPART IN THE REFERENCE: it exists for callers that cannot
 * allocate device memory — code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc.
 * Same arguments, same bytes in out_rgb / out_hit as rt_render, for ONE frame at samples == 1 of a scene without
 * refractive materials (anything else: RT_EINVAL).  rt_render keeps its tile records, work counters, occupancy
 * bits, frame record and primary-ray table in buffers it grows with hipMalloc on first use; this entry point keeps
 * them in a fixed array in device memory instead (RT_NOALLOC_SCRATCH_BYTES = 24 MiB per GPU the library has a
 * context on, held until the process ends), so it calls no hipMalloc / hipFree / allocating routine.  A frame
 * whose records do not fit — 2.25 bytes per pixel; 3840x2160 fits — is refused with RT_EINVAL and a message.
 * Every call on a GPU uses that GPU's one array: calls that overlap on one GPU (two contexts, or two streams)
 * race; calls on one stream are ordered.  The scene must have been uploaded before: rt_upload_scene, rt_set_refraction
 * and rt_set_cubes are NOT allocation-free (they synchronise, and grow the scene's tables — shadow-candidate lists,
 * direction table — with hipMalloc / hipFree when a scene needs more room than the last one). */
int rt_render_noalloc(rt_ctx* ctx, const rt_camera* cam, int width, int height, int samples,
                      int max_depth, int rng_mode, uint64_t seed, uint8_t* out_rgb, int32_t* out_hit);

/* Row-tile variant for multi-GPU sharding: renders row tiles t = tile_first, tile_first +
 * tile_stride, ... (tile t covers image rows [t*tile_rows, min((t+1)*tile_rows, height))) and
 * writes them packed, in that order, to out_rgb [n_local_rows][width][3] (and out_hit likewise).
 * The live equivalent of the reference's unused render_chunk tile split (raytracer.py:299-308).
 * *n_local_rows_out (nullable) receives the number of rows written. */
int rt_render_rows(rt_ctx* ctx, const rt_camera* cam, int width, int height, int samples,
                   int max_depth, int rng_mode, uint64_t seed, int tile_rows, int tile_first,
                   int tile_stride, uint8_t* out_rgb, int32_t* out_hit, int* n_local_rows_out);

/* rt_render_rows for a batch of frames (frame f: cams[f], seeds[f]) of one uploaded scene, up to RT_FRAMES_PER_LAUNCH
 * frames per kernel launch.  Frame f is written at out_rgb + f * stride * 3 bytes (out_hit + f *
 * stride ints), stride = out_frame_stride_px pixels, or n_local_rows * width when that is 0.  This
 * is what a rank of a multi-GPU animation / benchmark batch calls (row tiles of every frame into
 * equally sized slabs, one gather). */
int rt_render_frames_rows(rt_ctx* ctx, const rt_camera* cams, const uint64_t* seeds, int n_frames,
                          int width, int height, int samples, int max_depth, int rng_mode,
                          int tile_rows, int tile_first, int tile_stride, int64_t out_frame_stride_px,
                          uint8_t* out_rgb, int32_t* out_hit, int* n_local_rows_out);

/* render_chunk((x_start, x_end, y_start, y_end, width, height, scene, samples))
 * (raytracer.py:299-308): clamped, UNquantised colour of a sub-rectangle.
 *   out_rgbf: device float32 [y_end-y_start][x_end-x_start][3]  (the reference returns float64) */
int rt_render_chunk(rt_ctx* ctx, const rt_camera* cam, int x_start, int x_end, int y_start,
                    int y_end, int width, int height, int samples, int max_depth, int rng_mode,
                    uint64_t seed, float* out_rgbf);

/* Animation batch (BASELINE.json config 5): n_frames renders of one uploaded scene, frame f with
 * cams[f] and seeds[f], into out_rgb [n_frames][height][width][3].  No reference counterpart
 * beyond calling render() in a loop. */
int rt_render_frames(rt_ctx* ctx, const rt_camera* cams, const uint64_t* seeds, int n_frames,
                     int width, int height, int samples, int max_depth, int rng_mode,
                     uint8_t* out_rgb);

/* Place row tiles gathered from `n_ranks` ranks back into image order (on the root GPU, after
 * the RCCL gather).  gathered: device uint8 [n_ranks][rank_stride_bytes], rank r's slab holding
 * its packed rows as written by rt_render_rows(tile_first = r, tile_stride = n_ranks).
 * out_rgb: device uint8 [height][width][3]. */
int rt_scatter_rows(rt_ctx* ctx, const uint8_t* gathered, int64_t rank_stride_bytes, int n_ranks,
                    int width, int height, int tile_rows, uint8_t* out_rgb);
/* The same for a batch: frame f of rank r starts at gathered + r * rank_stride_bytes + f * frame_stride_bytes
 * (the layout a gather of rt_render_frames_rows slabs produces); out_rgb: [n_frames][height][width][3].
 * One launch for the whole batch. */
int rt_scatter_rows_frames(rt_ctx* ctx, const uint8_t* gathered, int64_t rank_stride_bytes,
                           int64_t frame_stride_bytes, int n_ranks, int n_frames, int width, int height,
                           int tile_rows, uint8_t* out_rgb);

/* Multi-GPU exchange without black pixels (no reference counterpart: the reference has no distributed code; its
 * unused render_chunk tile split, raytracer.py:299-308, is the model).  Most of a frame is background (81 % of
 * config 2's 8x8 tiles), the renderer knows which tiles are not (the visibility pass's occupancy bits), and a rank
 * renders several times what an xGMI link carries, so only occupied tiles travel.
 *
 * rt_compact_tiles (every rank, right after the render call that wrote `slab`, same stream): the occupied 8x8
 *   tiles of that batch are copied into `entries` (device; capacity_tiles entries of RT_TILE_ENTRY_BYTES: uint32
 *   tile id, then 8 rows x 24 bytes of packed RGB) and *count_dev (device uint32) receives their number, which may
 *   exceed capacity_tiles (then entries were dropped: size the buffer for every tile, ((w+7)/8|1) * ((rows+7)/8)
 *   per frame).  Tile id = (frame * tiles_y + tile_row) * pitch + tile_col, pitch = ((w+7)/8)|1, in the slab's own
 *   row numbering.  `slab` must be the out_rgb of the last render call on this context (RT_ESTATE otherwise).
 *   Nothing is read back: the host never waits for the render.
 * rt_place_tiles (root, after the gather): entries [n_ranks][capacity_tiles] as gathered (rank r's written by
 *   rt_render_frames_rows(tile_first = r, tile_stride = n_ranks) + rt_compact_tiles), counts_dev [n_ranks]
 *   (device) -> out_rgb [n_frames][height][width][3], every byte written once: black where no entry covers it.
 *   `table` is work space: table_words >= n_ranks * n_frames * (((width+7)/8)|1) * ((rows_max+7)/8) uint32,
 *   rows_max = the rows of rank 0. */
#define RT_TILE_ENTRY_BYTES 196
int rt_compact_tiles(rt_ctx* ctx, const uint8_t* slab, uint8_t* entries, uint32_t* count_dev, int capacity_tiles);
int rt_place_tiles(rt_ctx* ctx, const uint8_t* entries, const uint32_t* counts_dev, int n_ranks, int capacity_tiles,
                   int n_frames, int width, int height, int tile_rows, uint32_t* table, int64_t table_words,
                   uint8_t* out_rgb);

/* RCCL behind the C ABI (no reference counterpart: the reference has no distributed code; SURVEY.md §5 / §8(b)
 * sketch the gather of row tiles to GPU 0 as grouped ncclSend / ncclRecv on a communicator the library owns).
 * One process per GPU, one communicator per context.  The library binds librccl.so.1 at run time (dlopen: the copy the
 * process has already loaded — torch's — if there is one, else the loader's search path, else RT355_RCCL_LIB), so
 * librt355.so itself has no link-time dependency on RCCL and single-GPU users never load it.
 *   rt_comm_unique_id   one rank (rank 0) makes the 128-byte id (ncclGetUniqueId); the host hands it to the other
 *                       ranks by whatever means it has (raytracer_amd uses torch.distributed's store: plumbing).
 *   rt_comm_init        collective over the n_ranks processes: ncclCommInitRank on the context's GPU.
 *   rt_comm_destroy     collective; rt_destroy calls it for a context that still holds a communicator.
 *   rt_comm_all_gather_counts   counts_all_dev[r] = rank r's *count_dev (one 32-bit word each: the tile counts
 *                       rt_compact_tiles leaves on the device), ncclAllGather on `hip_stream`.
 *   rt_comm_gather      rank r's `send_bytes` bytes at `send` arrive on `root` at gathered + r * rank_stride_bytes.
 *                       One grouped call: the root posts an ncclRecv of recv_bytes[r] bytes per rank (host array
 *                       [n_ranks], read on the root only; it must name what each rank sends — both sides take the
 *                       numbers from the count exchange), every rank an ncclSend of its own bytes, the root included
 *                       (its share is a local copy inside RCCL) — n_ranks - 1 concurrent point-to-point transfers into
 *                       the root, one per xGMI link, each of its own length: no padding to the largest rank travels.
 *                       Ranks that have nothing to send post nothing.
 * All device pointers; `hip_stream` is the caller's hipStream_t as void* (NULL = the context's stream of
 * rt_set_stream): the calls are asynchronous on it and ordered with whatever else it holds. */
#define RT_COMM_ID_BYTES 128
int rt_comm_unique_id(uint8_t id[RT_COMM_ID_BYTES]);
int rt_comm_init(rt_ctx* ctx, const uint8_t id[RT_COMM_ID_BYTES], int n_ranks, int rank);
int rt_comm_destroy(rt_ctx* ctx);
/* n_ranks / rank of the context's communicator (0 / -1 without one) and the RCCL version code bound (0: none yet). */
int rt_comm_info(const rt_ctx* ctx, int* n_ranks, int* rank, int* rccl_version);
int rt_comm_all_gather_counts(rt_ctx* ctx, const uint32_t* count_dev, uint32_t* counts_all_dev, void* hip_stream);
int rt_comm_gather(rt_ctx* ctx, const void* send, int64_t send_bytes, void* gathered, const int64_t* recv_bytes,
                   int64_t rank_stride_bytes, int root, void* hip_stream);

/* Ray statistics accumulated by every render call since the last reset (device counters; this
 * call synchronises the stream).  out[0] = primary rays, out[1] = reflection rays,
 * out[2] = shadow rays, out[3] = shaded hits (compute_lighting calls).  A "ray" is one
 * find_nearest_intersection-equivalent scan (raytracer.py:152-172), SURVEY.md §8d. */
int rt_get_ray_counts(rt_ctx* ctx, uint64_t out[4]);
int rt_reset_ray_counts(rt_ctx* ctx);

/* Raw counter words summed over slots (diagnostics): [0..3] as rt_get_ray_counts' inputs, [7..15]
 * per-phase s_memtime sums of the stream kernel (tools/stamps.py), non-zero only in a -DRT_STAMPS=1
 * build; [4..6] culling statistics in a -DRT_CULL_STATS=1 build. */
int rt_debug_counters(rt_ctx* ctx, uint64_t out[16]);

/* Build/ABI information. */
const char* rt_version(void);

#ifdef __cplusplus
}
#endif
#endif /* RT355_H */
