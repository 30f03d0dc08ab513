// rt_kernels_misc.hpp — part of the ONE translation unit rt355.hip (included there, inside its anonymous namespace; not a
// header for anyone else).  rt_pixel_kernel (one thread per pixel: the A/B baseline) and the multi-GPU exchange kernels (row scatter, tile
// compaction and placement).
#pragma once

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
// One wave looks at n_chunks (<= kCompactChunks) x 64 tile ids OF ONE COUNTER h (ids slot * 32 + h, consecutive slots): their occupancy
// bits are consecutive bits of row h of `occ` — two scalar dwords per 64 ids, no per-lane look-up.  A workgroup of four waves
// reserves room for all its kept tiles with ONE atomic.  (Round 3.  Before: one lane per consecutive id — a 32-line gather
// per wave — and one atomic per wave; 130 000 atomics on one address took 0.66 ms per 256 1080p frames, five times what the
// copy takes.  Sixteen waves per workgroup: fewer atomics still, but the barrier then holds them for the slowest.)
#ifndef RT_COMPACT_CHUNKS
#define RT_COMPACT_CHUNKS 16
#endif
#ifndef RT_COMPACT_AT_ONCE
#define RT_COMPACT_AT_ONCE 4
#endif
constexpr int kCompactThreads = 256, kCompactChunks = RT_COMPACT_CHUNKS;
template <bool SCAN>
__global__ __launch_bounds__(kCompactThreads) void rt_compact_tiles_kernel(const uint8_t* __restrict__ slab, TileGeom g,
                                                               const uint32_t* __restrict__ occ, int occ_words,
                                                               uint8_t* __restrict__ entries, unsigned int* __restrict__ count,
                                                               unsigned int capacity, int n_chunks) {
  constexpr int kWaves = kCompactThreads / 64;
  __shared__ unsigned long long masks[kWaves][kCompactChunks];
  __shared__ unsigned int wave_count[kWaves + 1];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6));
  const uint32_t total = uint32_t(g.tile_pitch) * uint32_t(g.tiles8_y) * uint32_t(g.n_frames);   // (< 2^31: the host checks)
  const uint32_t per_frame = uint32_t(g.tile_pitch * g.tiles8_y);
  const uint32_t gw = uint32_t(blockIdx.x) * kWaves + uint32_t(wave);
  const uint32_t h = gw % kTileCounters, slot_base = (gw / kTileCounters) * (64u * uint32_t(n_chunks));
  // where tile id `id` lies (the odd pitch leaves padding ids without a tile: tx >= tiles8_x)
  auto where = [&](uint32_t id, int& frame, int& ty, int& tx) {
    frame = int(id / per_frame);
    const uint32_t rem = id - uint32_t(frame) * per_frame;
    ty = int(rem / uint32_t(g.tile_pitch)); tx = int(rem) - ty * g.tile_pitch;
  };
  unsigned int mine = 0u;
  for (int c = 0; c < n_chunks; ++c) {
    const uint32_t slot0 = slot_base + 64u * uint32_t(c);
    unsigned long long m = 0ull;
    if (slot0 * kTileCounters + h < total) {                 // (wave-uniform: the chunk's first id exists)
      if constexpr (SCAN) {
        const uint32_t id = (slot0 + uint32_t(lane)) * kTileCounters + h;
        bool keep = false;
        if (id < total) {
          int frame, ty, tx;
          where(id, frame, ty, tx);
          if (tx < g.tiles8_x) {
            const int rows = min(8, g.n_rows - ty * 8), nbytes = min(8, g.nx - tx * 8) * 3;
            const uint8_t* src = slab + ((long long)frame * g.frame_stride_px + (long long)(ty * 8) * g.nx + tx * 8) * 3;
            unsigned int any = 0u;
            for (int r = 0; r < rows; ++r)
              for (int b = 0; b < nbytes; ++b) any |= src[(long long)r * g.nx * 3 + b];
            keep = any != 0u;
          }
        }
        m = __ballot(keep);
      } else {
        const uint32_t* row = occ + size_t(h) * size_t(occ_words) + (slot0 >> 5);      // (wave-uniform: scalar loads)
        m = (unsigned long long)row[0] | ((unsigned long long)row[1] << 32);          // (occ_words has a spare word)
        // (bits of ids at or beyond `total` are never set: the visibility pass marks tiles that exist)
      }
    }
    if (lane == 0) masks[wave][c] = m;
    mine += (unsigned int)__popcll(m);
  }
  if (lane == 0) wave_count[wave] = mine;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned int sum = 0u;
    for (int w = 0; w < kWaves; ++w) { const unsigned int n = wave_count[w]; wave_count[w] = sum; sum += n; }
    wave_count[kWaves] = sum ? atomicAdd(count, sum) : 0u;
  }
  __syncthreads();
  if (mine == 0u) return;
  const unsigned int base = __builtin_amdgcn_readfirstlane(wave_count[kWaves] + wave_count[wave]);
  const bool dwords = (g.nx & 3) == 0 && (g.frame_stride_px & 3) == 0 && (reinterpret_cast<uintptr_t>(slab) & 3) == 0;
  const int r48 = (lane * 43) >> 8, w48 = lane - 6 * r48;    // lane l < 48: dword w of tile row r
  int k = 0;
  for (int c = 0; c < n_chunks; ++c) {
    unsigned long long todo = masks[wave][c];
    todo = ((unsigned long long)uint32_t(__builtin_amdgcn_readfirstlane(int(uint32_t(todo >> 32)))) << 32) |
           (unsigned long long)uint32_t(__builtin_amdgcn_readfirstlane(int(uint32_t(todo))));
    if (todo == 0ull) continue;
    const uint32_t id = (slot_base + 64u * uint32_t(c) + uint32_t(lane)) * kTileCounters + h;
    int frame, ty, tx;
    where(id, frame, ty, tx);
    // The wave copies the chunk's kept tiles FOUR at a time: four loads are in flight before the first store waits for its own.
    constexpr int kAtOnce = RT_COMPACT_AT_ONCE;
    while (todo != 0ull) {
      uint32_t val[kAtOnce];
      uint8_t* dst[kAtOnce];
      unsigned int ids[kAtOnce];
      bool fast[kAtOnce];
#pragma unroll
      for (int u = 0; u < kAtOnce; ++u) {
        fast[u] = false; val[u] = 0u; dst[u] = nullptr; ids[u] = 0u;
        if (todo == 0ull) continue;                            // (wave-uniform)
        const int src_lane = __ffsll((long long)todo) - 1;
        todo &= todo - 1ull;
        const unsigned int slot = base + unsigned(k++);
        if (slot >= capacity) continue;                        // (the count still says how many there were: the host checks)
        const int f = __builtin_amdgcn_readlane(frame, src_lane), y0 = __builtin_amdgcn_readlane(ty, src_lane) * 8,
                  x0 = __builtin_amdgcn_readlane(tx, src_lane) * 8;
        ids[u] = unsigned(__builtin_amdgcn_readlane(int(id), src_lane));
        uint8_t* e = entries + size_t(slot) * kTileEntryBytes;
        dst[u] = e;
        const uint8_t* src = slab + ((long long)f * g.frame_stride_px + (long long)y0 * g.nx + x0) * 3;
        const int rows = min(8, g.n_rows - y0), cols = min(8, g.nx - x0);
        if (dwords && cols == 8) {
          fast[u] = true;
          if (lane < 48 && r48 < rows) val[u] = *reinterpret_cast<const uint32_t*>(src + (long long)r48 * g.nx * 3 + w48 * 4);
        } else {
          if (lane == 0) *reinterpret_cast<uint32_t*>(e) = ids[u];
          for (int b = lane; b < 192; b += 64) {
            const int r = b / 24, cc = b - 24 * r;
            e[4 + b] = (r < rows && cc < cols * 3) ? src[(long long)r * g.nx * 3 + cc] : uint8_t(0);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < kAtOnce; ++u) {
        if (!fast[u]) continue;                                // (wave-uniform)
        if (lane == 0) *reinterpret_cast<uint32_t*>(dst[u]) = ids[u];
        if (lane < 48) reinterpret_cast<uint32_t*>(dst[u] + 4)[lane] = val[u];
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
__global__ __launch_bounds__(384) void rt_place_tiles_kernel(const uint8_t* __restrict__ entries, const uint32_t* __restrict__ table,
                                                             int capacity, long long ids_per_rank, int n_ranks, int width, int height,
                                                             int tile_rows, uint8_t* __restrict__ out, int vec4) {
  const int frame = blockIdx.y;
  const int tiles8_x = (width + 7) / 8, pitch = tiles8_x | 1;
  const int n_tiles = (height + tile_rows - 1) / tile_rows;
  const int row_bytes = width * 3;
  if (vec4 == 2 && (tile_rows & 7) == 0) {
    // The common case — row tiles of a multiple of 8 rows, rows of a multiple of 16 bytes, 16-byte aligned frames: the 8 rows
    // of the band lie in ONE row tile of one rank, so the band's index arithmetic (several integer divisions) and a lane's
    // table look-ups are done once and serve 8 rows.  A lane owns 16-byte columns of the band: four dwords from at most two
    // tiles (a tile row is 24 bytes, so the columns fall 0 / 16 / 8 bytes into a tile with a period of three), eight
    // independent loads and eight global_store_dwordx4; where both tiles are black — four fifths of config 2's — no load.
    const int y0 = int(blockIdx.x) * 8, rows = min(8, height - y0);
    const int t = y0 / tile_rows, rank = t % n_ranks;
    const int lr0 = (t / n_ranks) * tile_rows + (y0 - t * tile_rows);          // (a multiple of 8)
    const int mine = n_tiles > rank ? (n_tiles - rank + n_ranks - 1) / n_ranks : 0;
    const int rows_r = mine * tile_rows - (((n_tiles - 1) % n_ranks == rank) ? n_tiles * tile_rows - height : 0);
    const int tiles8_y = (rows_r + 7) / 8;
    const uint32_t* trow = table + (long long)rank * ids_per_rank + (long long)frame * pitch * tiles8_y + (long long)(lr0 >> 3) * pitch;
    const uint8_t* ebase = entries + size_t(rank) * capacity * kTileEntryBytes + 4;
    uint8_t* dst = out + ((long long)frame * height + y0) * row_bytes;
    for (int c = threadIdx.x; c < row_bytes / 16; c += blockDim.x) {
      const int d0 = 4 * c, tx0 = d0 / 6, tx1 = (d0 + 3) / 6;
      const uint32_t s0 = trow[tx0], s1 = tx1 == tx0 ? s0 : trow[tx1];
      const uint8_t* e[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int d = d0 + k, tx = d / 6, dw = d - 6 * tx;
        const uint32_t slot = tx == tx0 ? s0 : s1;
        e[k] = slot ? ebase + size_t(slot - 1u) * kTileEntryBytes + dw * 4 : nullptr;
      }
      uint4 v[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        v[r] = make_uint4(0u, 0u, 0u, 0u);
        if ((s0 | s1) != 0u && r < rows) {
          if (e[0]) v[r].x = *reinterpret_cast<const uint32_t*>(e[0] + r * 24);
          if (e[1]) v[r].y = *reinterpret_cast<const uint32_t*>(e[1] + r * 24);
          if (e[2]) v[r].z = *reinterpret_cast<const uint32_t*>(e[2] + r * 24);
          if (e[3]) v[r].w = *reinterpret_cast<const uint32_t*>(e[3] + r * 24);
        }
      }
#pragma unroll
      for (int r = 0; r < 8; ++r)
        if (r < rows) reinterpret_cast<uint4*>(dst + (long long)r * row_bytes)[c] = v[r];
    }
    return;
  }
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
    if (vec4 == 2) {
      // sixteen bytes per lane and store (rows of a multiple of 16 bytes, 16-byte aligned frames): four dwords from at most two
      // tiles — a tile row is 24 bytes, so chunks fall 0 / 16 / 8 bytes into a tile with a period of three — and one
      // global_store_dwordx4; a band of a mostly black frame is written at the rate of a clear
      for (int c = threadIdx.x; c < row_bytes / 16; c += 256) {
        uint32_t v[4];
        const int d0 = 4 * c, tx0 = d0 / 6, tx1 = (d0 + 3) / 6;
        const uint32_t s0 = trow[tx0], s1 = tx1 == tx0 ? s0 : trow[tx1];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int d = d0 + k, tx = d / 6, dw = d - 6 * tx;
          const uint32_t slot = tx == tx0 ? s0 : s1;
          v[k] = slot ? *reinterpret_cast<const uint32_t*>(erow + size_t(slot - 1u) * kTileEntryBytes + dw * 4) : 0u;
        }
        reinterpret_cast<uint4*>(dst)[c] = make_uint4(v[0], v[1], v[2], v[3]);
      }
    } else if (vec4) {
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
