// rt_visibility.hpp — part of the ONE translation unit rt355.hip (included there, inside its anonymous namespace; not a
// header for anyone else).  rt_visibility_kernel: primary rays of a batch of frames (render_pixel :289-295 + the first find_nearest_intersection).
#pragma once

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

// The same cone with all five directions evaluated by the lane itself: every lane may ask for a block of its own
// (the region's cull of all its blocks at once, rt_visibility_kernel).  Same expressions, same margin.
__device__ __forceinline__ RayCone bundle_cone_own(const RenderParams& p, const rt_camera& cam, int xa, int xb, int ya, int yb) {
  const float u0 = float(xa) * p.inv_w * 2.0f - 1.0f, u1 = float(xb) * p.inv_w * 2.0f - 1.0f;
  const float v0 = -(float(ya) * p.inv_h * 2.0f - 1.0f) * p.inv_aspect, v1 = -(float(yb) * p.inv_h * 2.0f - 1.0f) * p.inv_aspect;
  auto dir = [&](float ul, float vl, float& ex, float& ey, float& ez) {
    ex = fmaf(cam.up[0], vl, cam.right[0] * ul) + cam.forward[0];
    ey = fmaf(cam.up[1], vl, cam.right[1] * ul) + cam.forward[1];
    ez = fmaf(cam.up[2], vl, cam.right[2] * ul) + cam.forward[2];
    const float einv = frsq(fmaf(ez, ez, fmaf(ey, ey, ex * ex)));
    ex *= einv; ey *= einv; ez *= einv;
  };
  RayCone cone;
  dir(0.5f * (u0 + u1), 0.5f * (v0 + v1), cone.ax, cone.ay, cone.az);
  float cmin = 1.0f;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    float ex, ey, ez;
    dir((c & 1) ? u1 : u0, (c & 2) ? v1 : v0, ex, ey, ez);
    cmin = fminf(cmin, fmaf(ez, cone.az, fmaf(ey, cone.ay, ex * cone.ax)));
  }
  cmin = fmaxf(0.0f, cmin - 1e-5f);
  cone.cos2_t = cmin * cmin;
  cone.sin_t = fsqrt(fmaxf(0.0f, 1.0f - cone.cos2_t));
  return cone;
}

// One block of the visibility pass: kVisTiles 8x8 tiles side by side, rows row0 .. row0 + 7 of the window, one wave.
// (`pre`: the region has culled the scene against THIS block's cone already — `pre_first` is the ballot of spheres 0..63
// that reach it; scenes of up to 64 spheres without cubes)
template <int MODE, bool CUBES, bool MULTI, bool MANY = false>
__device__ __forceinline__ void visibility_block(const RenderParams& p, const int frame, const int row0, const int col0,
                                                 const int lane, const uint32_t region_lo = 0u, const uint32_t region_hi = 0u,
                                                 const bool pre = false, const unsigned long long pre_first = 0ull,
                                                 const uint32_t region_chunks = 0u, const uint32_t pre_klo = 0u, const uint32_t pre_khi = 0u) {
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
  if (cull && !region_kept && !pre) first_sphere = p.geom[lane < p.n_spheres ? lane : p.n_spheres - 1];
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
    if (cull && pre) {
      first = pre_first;           // (many spheres: not a ballot of sphere indices, only "something reaches the block" — the kept
      reached = first;             //  spheres come chunk by chunk in pre_klo / pre_khi, as the block's own cull would leave them)
      keep_lo = pre_klo; keep_hi = pre_khi;
    } else if (cull) {
      const int ya = tile_rows_uniform(p) ? tile_row_base(p, row0) : row_of(p, row0);
      const int yb = (tile_rows_uniform(p) ? tile_row_base(p, row0) + 7 : row_of(p, min(row0 + 7, p.n_rows - 1))) + 1;
      cone = bundle_cone(p, cam, lane, p.x0 + col0, p.x0 + col0 + 8 * NT, ya, yb);
      if (region_kept) {
        // (only the chunks in which the region kept a sphere — bit c of region_chunks —: walking all 16 .. 32 chunks to find
        // the two or three that hold something was most of the block's scalar work)
        for (uint32_t rc = region_chunks; rc != 0u; rc &= rc - 1u) {
          const int c = __ffs(int(rc)) - 1;
          const unsigned long long rm = mask_of_chunk(region_lo, region_hi, c);
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
    [[maybe_unused]] unsigned long long cube_mask = ~0ull;   // boxes 0..63 whose double cone the block's rays can meet (all, where not asked)
    if constexpr (CUBES) {   // (one cube per lane; beyond 64 cubes, or without a cone — no spheres at all —, every ray is built)
      cube_reach = true;
      if (cull && p.n_cubes <= 64) {
        const int c = lane < p.n_cubes ? lane : p.n_cubes - 1;
        cube_mask = __ballot(lane < p.n_cubes && box_in_double_cone(p.cubes[2 * c], p.cubes[2 * c + 1], cone, cam.origin[0],
                                                                    cam.origin[1], cam.origin[2]));
        cube_reach = cube_mask != 0ull;
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
                             cam.origin[1], cam.origin[2], dx, dy, dz, t, idx,
                             region_kept ? uint32_t(__ballot((keep_lo | keep_hi) != 0u)) : 0u);
    else   // few spheres: every group is scanned (pipelined scalar loads: cheaper here than group-by-group fetches)
      primary_hit_n<NT>(p.geom, table, p.n_spheres, cam.origin[0], cam.origin[1], cam.origin[2], dx, dy, dz, t, idx);
    if constexpr (CUBES) {   // the cubes of the object list (:159-160), wave-uniform scalar loads
      cgeom_t cg = as_const(p.cubes);
      float ix[NT], iy[NT], iz[NT];
#pragma unroll
      for (int q = 0; q < NT; ++q) { ix[q] = frcp(dx[q]); iy[q] = frcp(dy[q]); iz[q] = frcp(dz[q]); }
      // (only the boxes the block's cone — or its mirror image — can meet, in index order: a box it cannot meet is hit by no ray of
      // the block, so the nearest hit and the lowest index on a tie are those of the full scan)
      for (int c = 0; c < p.n_cubes; ++c) {
        if (c < 64 && ((cube_mask >> c) & 1ull) == 0ull) continue;
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
__global__ __launch_bounds__(RT_VIS_BLOCK, MANY ? RT_VIS_WAVES_MANY : RT_VIS_WAVES) void rt_visibility_kernel(const RenderParams p) {
  static_assert(!MANY || !CUBES, "the many-sphere form of the pass exists for scenes without cubes");
  constexpr int NT = kVisTiles;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int frame = blockIdx.z;
  // Which region row a workgroup takes.  Workgroups go to the 8 XCDs round-robin by linear id (x fastest), so with a number of
  // region rows that is a multiple of 4 an XCD would see the same one or two region rows of every column and frame — and
  // the rows through the spheres cost several times what the black ones do: 4 or 8 region rows (a rank's share at N = 7, 3,
  // 6) ran 2x slower than 3, 5 or 7.  There the row is rotated by column and frame, so every XCD gets every row.
  int region_row = int(blockIdx.x);
  if ((gridDim.x & 3u) == 0u) {
    region_row += int((blockIdx.y + blockIdx.z) & (gridDim.x >= 8u ? 7u : 3u));
    if (region_row >= int(gridDim.x)) region_row -= int(gridDim.x);
  }
  const int col0 = (int(blockIdx.y) * (RT_VIS_BLOCK / 64) + wave) * (8 * NT), rowR = region_row * (8 * kVisRows);   // window coordinates
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
    // Scenes of up to 64 spheres: one sphere per lane against the region's cone, one ballot.
    unsigned long long rmask = 0ull;
    bool have_rmask = false;
    float4 sphere = make_float4(0.f, 0.f, 0.f, 0.f);
    RayCone rcone{0.f, 0.f, 0.f, 0.f, 0.f};
    if (!region_kept && p.n_spheres > 0 && p.n_spheres <= 64) {
      sphere = p.geom[lane < p.n_spheres ? lane : p.n_spheres - 1];
      const FrameRec fr = as_const(p.frames)[frame];
      const int last = min(rowR + 8 * kVisRows, p.n_rows) - 1;
      const int ya = p.tile_stride == 1 ? p.y0 + p.tile_first * p.tile_rows + rowR : row_of(p, rowR);
      const int yb = (p.tile_stride == 1 ? p.y0 + p.tile_first * p.tile_rows + last : row_of(p, last)) + 1;
      rcone = bundle_cone(p, fr.cam, lane, p.x0 + col0, p.x0 + col0 + 8 * NT, ya, yb);
      rmask = __ballot(lane < p.n_spheres && sphere_in_cone(sphere, rcone, fr.cam.origin[0], fr.cam.origin[1], fr.cam.origin[2]));
      have_rmask = true;
    }
    if ((nothing_owed || black_owed) && (p.n_spheres <= 64 || region_kept) && (!CUBES || (p.n_spheres > 0 && p.n_cubes <= 64))) {
      bool any = region_any;
      if (have_rmask) {
        any = rmask != 0ull;
        if constexpr (CUBES) {
          const FrameRec fr = as_const(p.frames)[frame];
          const int c = lane < p.n_cubes ? lane : p.n_cubes - 1;
          any = any || __ballot(lane < p.n_cubes && box_in_double_cone(p.cubes[2 * c], p.cubes[2 * c + 1], rcone, fr.cam.origin[0],
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
    // The blocks of a region that something reaches, ALL AT ONCE: the k spheres of the region's ballot (a handful) times
    // the region's blocks fit one wave — lane l = (block l / k, the (l % k)-th sphere of the ballot) tests that sphere
    // against the cone of that block's rays, and one ballot holds the verdicts of every block.  A block then starts from
    // its bits (visibility_block: `pre`) instead of building its cone and culling again with a handful of lanes at work:
    // 78 % of the blocks of config 2's kept regions are empty, and each spent ~150 instructions to find that out.
    // The same for many spheres, where the region's kept spheres lie in the per-chunk ballots (lane c of region_lo / _hi): they
    // are counted, and if k x blocks fit the wave, lane (block, j) fetches the j-th of them and tests it against the block's cone.
    // A block's verdicts go back into the per-chunk form its scan reads (primary_hit_culled: `kept`).
    if constexpr (MANY && RT_VIS_REGION_BLOCKS) {
      const uint32_t rchunks = uint32_t(__ballot((region_lo | region_hi) != 0u));
      int k = 0;
      for (uint32_t rc = rchunks; rc != 0u; rc &= rc - 1u) k += __popcll(mask_of_chunk(region_lo, region_hi, __ffs(int(rc)) - 1));
      if (k > 0 && k * kVisRows <= 64 && tile_rows_uniform(p)) {
        const int magic = 65535 / k + 1;                      // lane / k for lane < 64 (exact: k <= 10)
        const int b = (lane * magic) >> 16, j = lane - b * k;
        int sphere = 0;
        uint32_t where = 0u;                                  // lane i < k: (chunk << 6 | bit) of the i-th kept sphere
        {
          int i = 0;
          for (uint32_t rc = rchunks; rc != 0u; rc &= rc - 1u) {
            const int c = __ffs(int(rc)) - 1;
            for (unsigned long long m = mask_of_chunk(region_lo, region_hi, c); m != 0ull; m &= m - 1ull, ++i) {
              const int bit = __ffsll((long long)m) - 1;
              if (j == i) sphere = c * 64 + bit;
              if (lane == i) where = uint32_t(c << 6 | bit);
            }
          }
        }
        const float4 mine = p.geom[sphere];
        const FrameRec fr = as_const(p.frames)[frame];
        const int ya = tile_row_base_lane(p, rowR + 8 * b);
        const RayCone cone = bundle_cone_own(p, fr.cam, p.x0 + col0, p.x0 + col0 + 8 * NT, ya, ya + 8);
        const unsigned long long block_bits =
            __ballot(lane < k * kVisRows && sphere_in_cone(mine, cone, fr.cam.origin[0], fr.cam.origin[1], fr.cam.origin[2]));
#pragma nounroll
        for (int bb = 0; bb < kVisRows; ++bb) {
          const int row0 = rowR + 8 * bb;
          if (row0 >= p.n_rows) break;
          const uint32_t sub = uint32_t(block_bits >> (bb * k)) & ((1u << k) - 1u);     // (k <= 10)
          int lane_b = int(threadIdx.x) & 63;
          asm volatile("" : "+v"(lane_b));
          uint32_t klo = 0u, khi = 0u;
          for (uint32_t sb = sub; sb != 0u; sb &= sb - 1u) {
            const uint32_t w = uint32_t(__builtin_amdgcn_readlane(int(where), __ffs(int(sb)) - 1));
            const int c = int(w >> 6), bit = int(w & 63u);
            if (lane_b == c) { if (bit < 32) klo |= 1u << bit; else khi |= 1u << (bit - 32); }
          }
          visibility_block<MODE, CUBES, MULTI, MANY>(params_here(), frame, row0, col0, lane_b, region_lo, region_hi, true,
                                                     sub ? 1ull : 0ull, 0u, klo, khi);
        }
        return;
      }
    }
    if constexpr (!CUBES && !MANY && RT_VIS_REGION_BLOCKS) {
      const int k = __popcll(rmask);
      if (have_rmask && k * kVisRows <= 64 && tile_rows_uniform(p)) {
        unsigned long long block_bits = 0ull;
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi(uint32_t(rmask >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(rmask), 0u));
        const bool in_mask = ((rmask >> lane) & 1ull) != 0ull;
        if (k > 0) {
          const int magic = 65535 / k + 1;                      // lane / k for lane < 64 (exact: k <= 10)
          const int b = (lane * magic) >> 16, j = lane - b * k;
          int src = 0;                                          // the lane that holds the j-th sphere of the ballot
          {
            unsigned long long m = rmask;
            for (int i = 0; i < k; ++i) {
              const int pos = __ffsll((long long)m) - 1;
              m &= m - 1ull;
              if (j == i) src = pos;
            }
          }
          auto from = [&](float v) { return __int_as_float(__builtin_amdgcn_ds_bpermute(src << 2, __float_as_int(v))); };
          const float4 mine = make_float4(from(sphere.x), from(sphere.y), from(sphere.z), from(sphere.w));
          const FrameRec fr = as_const(p.frames)[frame];
          const int ya = tile_row_base_lane(p, rowR + 8 * b);
          const RayCone cone = bundle_cone_own(p, fr.cam, p.x0 + col0, p.x0 + col0 + 8 * NT, ya, ya + 8);
          block_bits = __ballot(lane < k * kVisRows && sphere_in_cone(mine, cone, fr.cam.origin[0], fr.cam.origin[1], fr.cam.origin[2]));
        }
#pragma nounroll
        for (int b = 0; b < kVisRows; ++b) {
          const int row0 = rowR + 8 * b;
          if (row0 >= p.n_rows) break;
          const uint32_t sub = uint32_t(block_bits >> (b * k)) & ((1u << k) - 1u);     // (k <= 10)
          // the block's ballot over sphere INDICES again: sphere `lane` is in the region's ballot at rank r, and bit r of sub says
          const unsigned long long first = sub ? __ballot(in_mask && ((sub >> rank) & 1u) != 0u) : 0ull;
          int lane_b = int(threadIdx.x) & 63;
          asm volatile("" : "+v"(lane_b));
          visibility_block<MODE, CUBES, MULTI, MANY>(params_here(), frame, row0, col0, lane_b, 0u, 0u, true, first);
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
    visibility_block<MODE, CUBES, MULTI, MANY>(params_here(), frame, row0, col0, lane_b, region_lo, region_hi, false, 0ull,
                                               MANY ? uint32_t(__ballot((region_lo | region_hi) != 0u)) : 0u);
  }
}
