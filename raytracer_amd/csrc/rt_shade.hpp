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

// (ALL: the caller has checked that the lights ask for exactly B rays — the reference's 1 + 12 — so one batch holds them all and
// no ray needs the "beyond the light's sample count" guard: three scalar instructions and a select per ray)
template <int MODE, int B, bool REFR = false, bool CUBES = false, bool LIST = false, bool ALL = false>
__device__ __forceinline__ LightSums lighting(const DevLights& L, const float4* __restrict__ sg,
                                         const float4* __restrict__ gg, const float2* __restrict__ refr,
                                         const uint32_t* __restrict__ glass_bits,
                                         const uint32_t* __restrict__ occl_row, const uint2* __restrict__ occl_items, int n, int nc, float px, float py, float pz, float nx, float ny, float nz,
                                         float dx, float dy, float dz, float sx, float sy, float sz,
                                         uint32_t level_key) {
#if RT_CULL_STATS
  unsigned dbg_rel_ = 0, dbg_lanes_ = 0, dbg_iters_ = 0, dbg_turns_ = 0;
#endif
  const int nrays = ALL ? B : 1 + L.ns;
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
  constexpr bool SHARE = REFR || B < kReferenceShadowRays;     // (the reference's 13 rays in one batch: nothing to share, no registers held)
  [[maybe_unused]] uint32_t kept0 = 0u, kept1 = 0u, keptc = 0u;
  for (int base = 0; base < nrays; base += B) {
    float Lx[B], Ly[B], Lz[B], m[B];
    // REFR: a ray has TWO running verdicts — "no opaque object hit yet" in the sign bit of m[k], "no refractive one hit yet" in
    // its bit 0 (one register per ray, as in the opaque kernels: a second array of verdicts cost B registers, and with them
    // the single batch of 13 rays).  A candidate's verdict word is rotated onto the bit of its class (v_alignbit by a
    // per-lane amount) and every other bit is kept: m &= rot(verdict) | keep.
    constexpr uint32_t kNoHitYet = REFR ? 0x80000001u : 0xBF800000u;     // (0xBF800000 = -1.0f: the opaque kernels' "sign set")
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
      m[k] = __uint_as_float(kNoHitYet);   // sign bit set = "no sphere hit yet" in both accumulation forms
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
      [[maybe_unused]] const uint32_t rot = (glass >> 31) * 31u;                          // refractive: sign bit -> bit 0
      [[maybe_unused]] const uint32_t keep = 0x7FFFFFFFu ^ ((glass >> 31) * 0x80000001u);  // every bit but the class's own
#pragma unroll
      for (int k = 0; k < B; ++k) {
        const float nb = fmaf(cz, Lz[k], fmaf(cy, Ly[k], cx * Lx[k]));
        const float dh = fmaf(nb, nb, -chi);
        const float dlo = dh + thr2;
        const uint32_t verdict = __float_as_uint(dh) | __float_as_uint(nb);
        if constexpr (REFR) m[k] = __uint_as_float(__float_as_uint(m[k]) & (__builtin_amdgcn_alignbit(verdict, verdict, rot) | keep));
        else m[k] = __uint_as_float(__float_as_uint(m[k]) & verdict);
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
          if constexpr (REFR) m[k] = __uint_as_float(__float_as_uint(m[k]) & (__builtin_amdgcn_alignbit(verdict, verdict, rot) | keep));
          else m[k] = __uint_as_float(__float_as_uint(m[k]) & verdict);
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
                  m[k] = __uint_as_float(__float_as_uint(m[k]) & (glass_cube ? 0xFFFFFFFEu : 0x7FFFFFFFu));
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
      // (raytracer.py:187, :202).  The scan above has told, per ray, whether it hits SOME opaque object (sign of m[k] clear)
      // and whether it hits SOME refractive one (bit 0 clear) — the any-hit tests of the opaque path, over the same culled
      // candidates, no root taken.  No opaque hit: lit.  Opaque hits only: blocked.  Both (a ray through glass
      // towards something solid, or the other way round: rare): the nearest hit decides, found by the full scan
      // of find_nearest_intersection for that ray.  Cubes (their line rule, :137, :162) have entered their verdicts above.
      // (lit <=> no opaque hit or some refractive one <=> sign of w | ~(w << 31) set; whether any ray of the lane was hit
      // by objects of BOTH kinds — sign of w | (w << 31) clear — is asked once per batch.)
      // (m[k]: sign clear = an opaque object was hit, bit 0 clear = a refractive one was)
      uint32_t both_any = 0u;
#pragma unroll
      for (int k = 0; k < B; ++k) {
        const uint32_t w = __float_as_uint(m[k]);
        both_any |= ~(w | (w << 31));                            // sign set: both kinds
        asm volatile("" : "+v"(both_any));
      }
      if (__ballot(int(both_any) < 0) != 0ull) {
#pragma unroll
        for (int k = 0; k < B; ++k) {
          const uint32_t wk = __float_as_uint(m[k]);
          if (int(wk | (wk << 31)) >= 0 && base + k < nrays) {
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
            m[k] = __uint_as_float(blocked ? 0x00000001u : 0x80000001u);    // as if only the nearest kind had been hit
          }
        }
      }
#pragma unroll
      for (int k = 0; k < B; ++k) {
        const uint32_t w = __float_as_uint(m[k]);
        m[k] = __uint_as_float(w | ~(w << 31));                  // sign set: lit — no opaque hit, or some refractive one (what the weights look at)
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
  const uint32_t* row = LIST ? p.occl_start + occluder_row(p.occl_list_dim, idx, nx, ny, nz)
                             : p.occl + occluder_row(p.occl_dim, idx, nx, ny, nz) * size_t(p.occl_words);
  constexpr int kB = CUBES ? kCubeShadowBatch : kShadowBatch;
  LightSums w;
  if (RT_LIGHT_ALL && 1 + p.L.ns == kB)      // (wave-uniform) the reference's lights: all rays in one batch, none of them idle
    w = lighting<MODE, kB, false, CUBES, LIST, true>(p.L, sg, p.geom, nullptr, nullptr, row, p.occl_items, p.n_spheres, p.n_cubes,
                                                       px, py, pz, nx, ny, nz, dx, dy, dz, sx, sy, sz, level_key);
  else
    w = lighting<MODE, kB, false, CUBES, LIST>(p.L, sg, p.geom, nullptr, nullptr, row, p.occl_items, p.n_spheres, p.n_cubes,
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
