// rt_stream.hpp — part of the ONE translation unit rt355.hip (included there, inside its anonymous namespace; not a
// header for anyone else).  rt_stream_kernel: shading and every further bounce on persistent waves with per-wave work queues and tile slots in LDS.
#pragma once

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
  // ray statistics, wave-uniform (scalar registers: as per-lane counters they held two vector registers through every phase
  // of the kernel, and the many-sphere variants spilled them to scratch and back once per pass)
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
        bool traced_r = false, traced_f = false;
        n_shade += uint32_t(nb);
        if (have) {
          float nx, ny, nz, sx, sy, sz, lr, lg, lb;
          float4 m;
          float2 rf;
          {
            const RenderParams& p = params_here();                   // the lights and tables of lighting(): dead after it
            const int idx = meta_object(meta);
            surface_normal<CUBES>(sg, p.n_spheres, idx, px, py, pz, nx, ny, nz);   // :139-140 / :142-150
            sx = fmaf(nx, 0.001f, px); sy = fmaf(ny, 0.001f, py); sz = fmaf(nz, 0.001f, pz);
            constexpr int kB = CUBES ? kRefrCubeShadowBatch : kRefrShadowBatch;
            const uint32_t row = uint32_t(occluder_row(p.occl_dim, idx, nx, ny, nz)) * uint32_t(p.occl_words);
            const uint32_t lkey = mix(pixel_key, uint32_t(meta_level(meta) + 1));
            LightSums lw;
            if (RT_LIGHT_ALL && 2 * kB >= kReferenceShadowRays && 1 + p.L.ns == kReferenceShadowRays)      // (wave-uniform) the reference's lights: one batch holds every ray
              lw = lighting<MODE, kB, true, CUBES, false, true>(p.L, sg, p.geom, p.refr, as_const(p.glass_bits), p.occl, row, nullptr, p.n_spheres, p.n_cubes,
                                                                px, py, pz, nx, ny, nz, dx, dy, dz, sx, sy, sz, lkey);
            else
              lw = lighting<MODE, kB, true, CUBES>(p.L, sg, p.geom, p.refr, as_const(p.glass_bits), p.occl, row, nullptr, p.n_spheres, p.n_cubes,
                                                   px, py, pz, nx, ny, nz, dx, dy, dz, sx, sy, sz, lkey);
            // (material, record state and the meta word come in BEHIND the shadow loops, as in the opaque path:
            // the 13-ray batch leaves no registers for them)
            float spec;
            LateFromRecord late{p, late_state, q, e, meta, base, lane_here()};
            late.load(m, spec);
            // (the hit point again, from the record: the refracted ray starts from it below, and held in registers through the
            // shadow loops two of its three words went to scratch)
            px = q.at(0, e); py = q.at(1, e); pz = q.at(2, e);
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
            traced_r = true;
            if (p.refl_dim > 0) nearest_table<CUBES>(p, sg, meta_object(meta), sx, sy, sz, rdx, rdy, rdz, t, idx2, face);
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
              traced_f = true;
              // (the refracted ray starts 0.001 INSIDE the object: still within reach of the table's bound)
              if (p.refl_dim > 0) nearest_table<CUBES>(p, sg, meta_object(meta), ox, oy, oz, fdx, fdy, fdz, t, idx2, face);
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
        n_trace += uint32_t(__popcll(__ballot(traced_r)) + __popcll(__ballot(traced_f)));
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
      n_shade += uint32_t(nb);
      bool traced = false;
      if (have) {
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
        traced = cont;
        if (cont) {
          float t;
          int idx2, face;
          if constexpr (GRID != 0) {
            if (p.list_dim > 0) nearest_list(p, sg, meta_object(meta), ox, oy, oz, dx, dy, dz, t, idx2);
            else nearest_hit(sg, p.geom, p.n_spheres, ox, oy, oz, dx, dy, dz, t, idx2);   // (RT_OPT_REFLECT_TABLE off, or lists beyond 256 MB)
            face = 0;
          }
          else if (p.refl_dim > 0) nearest_table<CUBES>(p, sg, meta_object(meta), ox, oy, oz, dx, dy, dz, t, idx2, face);
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
      n_trace += uint32_t(__popcll(__ballot(traced)));
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
    const unsigned long long tot_t = n_trace, tot_s = n_shade;      // (a wave shades fewer than 2^32 records per launch)
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
