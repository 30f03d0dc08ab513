// rt_shade_batch.hpp — NOT a header: the body of one batch of shadow rays of lighting() (rt_shade.hpp), rays base .. base + B - 1,
// included there inside a scope that defines `base` — the loop variable of the general form, or a compile-time constant of the
// form for the reference's lights on kernels that take two batches (ALL && B < 13: every test of `base` and the ray the second
// batch does not have then fold away).  Written as a lambda called with constant bases, the same text cost the one-batch kernels
// 2 % (refractive_scene(), config 4) — hence the textual inclusion.  Everything it names is lighting()'s.
    float Lx[B], Ly[B], Lz[B], m[B];
    // REFR: a ray has TWO running verdicts — "no opaque object hit yet" in the sign bit of m[k], "no refractive one hit yet" in
    // its bit 0 (one register per ray, as in the opaque kernels: a second array of verdicts cost B registers, and with them
    // the single batch of 13 rays).  A candidate's verdict word is rotated onto the bit of its class (v_alignbit by a
    // per-lane amount) and every other bit is kept: m &= rot(verdict) | keep.
    constexpr uint32_t kNoHitYet = REFR ? 0x80000001u : 0xBF800000u;     // (0xBF800000 = -1.0f: the opaque kernels' "sign set")
#pragma unroll
    for (int k = 0; k < B; ++k) {
      if (ALL && B < kReferenceShadowRays && base + k >= kReferenceShadowRays) {              // (a ray the batch does not have: every later loop skips it)
        Lx[k] = 0.f; Ly[k] = 0.f; Lz[k] = 0.f; m[k] = __uint_as_float(kNoHitYet);
        continue;
      }
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
        if (ALL && B < kReferenceShadowRays && base + k >= kReferenceShadowRays) continue;      // (a ray the batch does not have: folds away)
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
          if (ALL && B < kReferenceShadowRays && base + k >= kReferenceShadowRays) continue;      // (a ray the batch does not have: folds away)
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
        uint32_t c = occl_base[occl_off];
        const uint32_t e = occl_base[occl_off + 1u];
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
        uint32_t cand = occl_base[occl_off + uint32_t(word)];
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
              if (ALL && B < kReferenceShadowRays && base + k >= kReferenceShadowRays) continue;      // (a ray the batch does not have: folds away)
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
        if (ALL && B < kReferenceShadowRays && base + k >= kReferenceShadowRays) continue;      // (a ray the batch does not have: folds away)
        const uint32_t w = __float_as_uint(m[k]);
        both_any |= ~(w | (w << 31));                            // sign set: both kinds
        asm volatile("" : "+v"(both_any));
      }
      if (__ballot(int(both_any) < 0) != 0ull) {
#pragma unroll
        for (int k = 0; k < B; ++k) {
          if (ALL && B < kReferenceShadowRays && base + k >= kReferenceShadowRays) continue;      // (a ray the batch does not have: folds away)
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
        if (ALL && B < kReferenceShadowRays && base + k >= kReferenceShadowRays) continue;      // (a ray the batch does not have: folds away)
        const uint32_t w = __float_as_uint(m[k]);
        m[k] = __uint_as_float(w | ~(w << 31));                  // sign set: lit — no opaque hit, or some refractive one (what the weights look at)
      }
    }
    // view . normal, view = -D (:257).  The list kernels form it here, behind the candidate loops: from the top of the function
    // it was one more register live through all of them, which they spilled and read back once per ray (config 4: -15 %);
    // the other kernels have the register, and for them the early form measures 0.8 % faster.
    const float vdn = LIST ? -(fmaf(dz, nz, fmaf(dy, ny, dx * nx))) : vdn_early;
#pragma unroll
    for (int k = 0; k < B; ++k) {
      if (ALL && B < kReferenceShadowRays && base + k >= kReferenceShadowRays) continue;      // (a ray the batch does not have: folds away)
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
