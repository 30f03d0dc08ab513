// rt_tables.hpp — part of the ONE translation unit rt355.hip (included there, inside its anonymous namespace; not a
// header for anyone else).  Host side, scene upload: the direction table of reflection rays (bit masks / lists) and the occluder table of shadow
// rays, built in FP64 as conservative supersets ("Tables").
#pragma once

constexpr int kGridFrom = 256;   // scenes with at least this many spheres: shadow candidates and direction-table entries as LISTS (the list kernels)
// Direction table for nearest_table(): bit j of entry [k][bucket] says that SOME ray which starts within rho_k =
// r_k + 0.001 (+ margin) of K and whose direction lies in the bucket can hit sphere j.  Such a ray reaches a point
// X of sphere j from an origin O:  X - O = (J - K) + e  with |e| <= rho_k + r_j, so its direction is within
// theta_kj = asin((rho_k + r_j) / |J - K|) of u = (J - K) / |J - K|  (any direction if the balls touch).  A bucket is
// a square patch of a cube face; the angle to a fixed axis is largest at a corner of the patch, so the patch lies in
// the cone (axis through its centre, half-angle h to the farthest corner), and j is listed iff
// angle(axis, u) <= theta_kj + h + 1e-3 (the last term: the kernel's FP32 bucket arithmetic, ~1e-6).  k itself is
// never listed: a ray that leaves a sphere outwards cannot come back to it.
// Boxes (rt_set_cubes) take part through their bounding spheres, as objects n .. n + n_boxes - 1: as SOURCES with rho = half
// diagonal (+ 0.001: the origin lies within that of the centre), as TARGETS with the line rule of intersect_cube — a box is
// hit by every ray whose LINE crosses it (cube_t: t < 0 counts), so a box is listed where the bucket's cone meets the cone
// towards it OR its mirror image; and a box lists ITSELF in every bucket (the line of a ray that leaves a face runs back
// through the box: the reference "hits" it at t < 0; a refracted ray inside a glass box meets its exit face).
constexpr int kReflTabFrom = RT_REFL_TABLE_FROM;   // scenes with at least this many spheres (and fewer than the grid's threshold)
constexpr size_t kReflTabMaxBytes = size_t(1) << 20;
int reflect_table_dim(int n, int n_boxes = 0) {
  if (n + n_boxes < kReflTabFrom || n >= kGridFrom) return 0;
  const size_t pairs = size_t((n + 63) / 64) + size_t((n_boxes + 63) / 64);
  for (int B : {16, 8}) if (size_t(n + n_boxes) * size_t(6 * B * B) * pairs * 8 <= kReflTabMaxBytes) return B;
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
// (`geom4`: the n spheres, then the bounding spheres of the boxes: centre, radius^2; objects from n_spheres on are boxes)
template <typename Emit>
void reflect_buckets_of(const float* geom4, int n, int k, int B, const std::vector<DirPatch>& fine,
                        const std::vector<DirPatch>& coarse, Emit&& emit, int n_spheres = -1) {
  if (n_spheres < 0) n_spheres = n;
  const double kPi = 3.14159265358979323846;
  const int nb = 6 * B * B, Bc = coarse.empty() ? 0 : 4, sub = Bc ? B / Bc : 0;
  const double kx = geom4[4 * k], ky = geom4[4 * k + 1], kz = geom4[4 * k + 2];
  const double rho = std::sqrt(double(geom4[4 * k + 3])) * 1.001 + 2e-3;
  for (int j = 0; j < n; ++j) {
    const bool box = j >= n_spheres;
    if (j == k) {
      if (box) for (int b = 0; b < nb; ++b) emit(b, j);       // a box lists itself everywhere (see above)
      continue;
    }
    const double vx = geom4[4 * j] - kx, vy = geom4[4 * j + 1] - ky, vz = geom4[4 * j + 2] - kz;
    const double d = std::sqrt(vx * vx + vy * vy + vz * vz);
    const double reach = rho + std::sqrt(double(geom4[4 * j + 3])) * 1.001 + 1e-3;
    const bool all = !(d > reach * 1.0001);
    const double sn = all ? 1.0 : reach / d, theta = all ? kPi : std::asin(sn), cs = std::sqrt(std::max(0.0, 1.0 - sn * sn));
    const double ux = all ? 0.0 : vx / d, uy = all ? 0.0 : vy / d, uz = all ? 0.0 : vz / d;
    // angle(axis, u) <= theta + h  <=>  axis.u >= cos(theta + h)   (theta + h < pi)
    auto reaches = [&](const DirPatch& P) {
      const double along = P.x * ux + P.y * uy + P.z * uz;
      return all || theta + P.h >= kPi || (box ? std::fabs(along) : along) >= cs * P.ch - sn * P.sh;   // (box: the cone or its mirror image)
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
// Entry [k][bucket] = ceil(n / 64) pairs of sphere bits, then ceil(n_boxes / 64) pairs of box bits.  `boxes`: n_boxes x {min
// xyz -, max xyz -} as rt_set_cubes stores them, or null.
void build_reflect_table(const float* geom4, int n, int B, std::vector<uint2>& tab, const float4* boxes = nullptr, int n_boxes = 0) {
  const int pairs_s = (n + 63) / 64, pairs = pairs_s + (n_boxes + 63) / 64, nb = 6 * B * B, n_obj = n + n_boxes;
  tab.assign(size_t(n_obj) * size_t(nb) * size_t(pairs), make_uint2(0u, 0u));
  const std::vector<DirPatch> fine = direction_patches(B);
  const std::vector<DirPatch> coarse = (B > 4 && B % 4 == 0) ? direction_patches(4) : std::vector<DirPatch>();
  std::vector<float> obj(geom4, geom4 + size_t(4) * size_t(n));
  for (int c = 0; c < n_boxes; ++c) {          // bounding spheres, rounded UP (float centre: half an ulp is inside the table's margins)
    const float4 lo = boxes[2 * c], hi = boxes[2 * c + 1];
    const double hx = 0.5 * (double(hi.x) - lo.x), hy = 0.5 * (double(hi.y) - lo.y), hz = 0.5 * (double(hi.z) - lo.z);
    obj.push_back(float(lo.x + hx)); obj.push_back(float(lo.y + hy)); obj.push_back(float(lo.z + hz));
    obj.push_back(float((hx * hx + hy * hy + hz * hz) * 1.001 + 1e-6));
  }
  for (int k = 0; k < n_obj; ++k)
    reflect_buckets_of(obj.data(), n_obj, k, B, fine, coarse, [&](int b, int j) {
      const int w = j < n ? (j >> 6) : pairs_s + ((j - n) >> 6), bit = j < n ? (j & 63) : ((j - n) & 63);
      uint2& e = tab[(size_t(k) * size_t(nb) + size_t(b)) * size_t(pairs) + size_t(w)];
      if (bit & 32) e.y |= 1u << (bit & 31); else e.x |= 1u << (bit & 31);
    }, n);
}
// The table as lists (nearest_list): start[k * nb + b] .. start[k * nb + b + 1] are the 4-index chunks of entry
// [k][b] in `items` (ascending sphere indices, the last chunk padded with 0xFFFF).  Objects are independent, so the
// build runs on several threads.  Returns false — no table, reflection rays scan every sphere — if the lists would exceed
// kReflListMaxChunks (a scene where most spheres reach most directions of most objects: huge or interlocked spheres).
constexpr int kReflListDim = RT_REFL_LIST_DIM;
constexpr int kReflListFineTo = 1280;     // scenes of up to this many spheres: twice as many buckets per face edge
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

// The same question per (object, BUCKET OF THE SURFACE NORMAL at the shaded point) — direction_bucket() of the unit
// normal, the cube map of the direction table.  A hit on sphere k whose normal lies in bucket b (axis a, half-angle h)
// has its shadow origin S = P + 0.001 N within chord(h) (r_k + 0.001) of Q = K + (r_k + 0.001) a: the ball (Q, rho_b)
// takes the place of the whole sphere's ball (K, rho) above, and far fewer spheres reach into the much narrower cone of
// its rays.  Sphere k ITSELF is a candidate only where some ray of the bucket can point back into it: a ray that leaves
// S with N . u >= 0 moves away from K for good, and N . u >= cos(h + angle(a, A - Q) + asin((rho_b + R) / |A - Q|)).
// Most shaded points face their lights and are shadowed by nothing: their rows are EMPTY, and a pass of such points
// skips the candidate loops of lighting() altogether (the object-level rows always held the sphere itself).
// Rows [k][6 B^2][ceil(n / 32)]; candidates are drawn from the object-level row (a superset), so the build is
// n x 6 B^2 x (a few) cone tests.  Margins: `h` carries 1e-3 rad for the kernel's FP32 bucket arithmetic
// (direction_patches) plus the FP32 error of the normal itself, ~|P| ulp / r_k.
int occluder_table_dim(int n_objects, int n_spheres) {
  if (n_spheres < 1) return 0;
  if (n_spheres >= kGridFrom) return 8;                      // (lists: 384 entries per object)
  const size_t words = size_t((n_spheres + 31) / 32);
  for (int B : {16, 8, 4}) if (size_t(n_objects) * size_t(6 * B * B) * words * 4 <= (size_t(4) << 20)) return B;
  return 0;
}
void build_occluder_buckets(const float* geom4, int n, const DevLights& L, int B, const std::vector<uint32_t>& object_rows,
                            std::vector<uint32_t>& tab) {
  const int words = (n + 31) / 32, nb = 6 * B * B;
  const double kHalfPi = 1.57079632679489661923;
  tab.assign(size_t(n) * size_t(nb) * size_t(words), 0u);
  const std::vector<DirPatch> patch = direction_patches(B);
  struct Ball { double x, y, z, r; };
  const Ball lights[2] = {{L.ppos[0], L.ppos[1], L.ppos[2], 1e-4},
                          {L.apos[0], L.apos[1], L.apos[2], double(L.arad) * 1.001 + 1e-4}};
  std::vector<int> cand;
  for (int k = 0; k < n; ++k) {
    const double kx = geom4[4 * k], ky = geom4[4 * k + 1], kz = geom4[4 * k + 2];
    const double rk = std::sqrt(double(geom4[4 * k + 3]));
    const double reach = std::fabs(kx) + std::fabs(ky) + std::fabs(kz) + rk;          // >= |P| for every point of the sphere
    const uint32_t* orow = object_rows.data() + size_t(k) * size_t(words);
    cand.clear();
    for (int c = 0; c < n; ++c) if ((orow[c >> 5] >> (c & 31)) & 1u) cand.push_back(c);
    // (a sphere too small for its normals to mean anything in FP32 keeps the object-level row in every bucket)
    const double h_extra = rk > 0.0 ? 1e-6 * reach / rk : 10.0;
    for (int b = 0; b < nb; ++b) {
      uint32_t* row = tab.data() + (size_t(k) * size_t(nb) + size_t(b)) * size_t(words);
      const DirPatch& P = patch[size_t(b)];
      const double hh = P.h + h_extra;
      bool all = !(hh < 1.0);
      const double rho = 2.0 * (rk + 0.001) * std::sin(0.5 * std::min(hh, 1.0)) * 1.001 + 1e-3 + 1e-6 * reach;
      const double qx = kx + (rk + 0.001) * P.x, qy = ky + (rk + 0.001) * P.y, qz = kz + (rk + 0.001) * P.z;
      for (const Ball& A : lights) {
        if (all) break;
        const double ax = A.x - qx, ay = A.y - qy, az = A.z - qz;
        const double d = std::sqrt(ax * ax + ay * ay + az * az);
        if (!(d > (rho + A.r) * 1.01)) { all = true; break; }
        const double ux = ax / d, uy = ay / d, uz = az / d;
        const double sn = (rho + A.r) / d, cs = std::sqrt(1.0 - sn * sn);
        const double back = rho / sn;
        const double px = qx - ux * back, py = qy - uy * back, pz = qz - uz * back;
        for (int c : cand) {
          if (c == k) {
            // can a ray of this bucket point back into the sphere it leaves?  (see above; the floor on the cosine keeps the
            // sign of the kernel's FP32 `nb` = (K - S) . u, whose magnitude is (r_k + 0.001) N . u, out of rounding's reach)
            const double phi = hh + std::acos(std::max(-1.0, std::min(1.0, P.x * ux + P.y * uy + P.z * uz))) + std::asin(sn);
            const double floor_cos = std::max(1e-3, 1e-5 * (reach + 1.0) / (rk + 0.001));
            if (!(phi < kHalfPi && std::cos(phi) >= floor_cos)) row[c >> 5] |= 1u << (c & 31);
            continue;
          }
          const double cx = geom4[4 * c] - px, cy = geom4[4 * c + 1] - py, cz = geom4[4 * c + 2] - pz;
          const double d2 = cx * cx + cy * cy + cz * cz;
          const double along = cx * ux + cy * uy + cz * uz;
          const double perp = std::sqrt(std::max(0.0, d2 - along * along));
          const double rc = std::sqrt(double(geom4[4 * c + 3]));
          if (perp * cs - along * sn <= rc * 1.001 + 1e-3 * (1.0 + std::sqrt(d2))) row[c >> 5] |= 1u << (c & 31);
        }
      }
      if (all) for (int w = 0; w < words; ++w) row[w] = orow[w];
    }
  }
}
