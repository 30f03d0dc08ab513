// TEST INFRASTRUCTURE ONLY. CPU restatement of the reference's per-pixel render path
// (/root/reference/raytracer.py), used as the parity checker by tests/, by
// __graft_entry__.smoke() and as bench.py's cpu_baseline ("port").  The product path
// (raytracer_amd/) never links, loads or calls this file.
//
// Pinning: this restatement is checked against the reference ITSELF, imported in the build
// container by oracle/ref_harness.py (golden vectors committed under tests/golden/, generator
// oracle/gen_golden.py).  In double precision it reproduces the reference's FP64 results
// bit-for-bit on those vectors: every expression below keeps the reference's operation order,
// uses libm pow() where the reference writes `**` (CPython float_pow calls the same libm pow),
// and the file must be compiled with -ffp-contract=off so no FMA is formed.
//
// The float instantiation (precision=1) is a diagnostic: it predicts which pixels an FP32
// implementation may legitimately decide differently (grazing rays), it is not an oracle.
//
// Each function cites the reference lines it follows.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

template <typename T>
struct V3 {  // raytracer.py:10-35
  T x, y, z;
};
template <typename T> inline V3<T> add(V3<T> a, V3<T> b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }   // :16-17
template <typename T> inline V3<T> sub(V3<T> a, V3<T> b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }   // :19-20
template <typename T> inline V3<T> mulv(V3<T> a, V3<T> b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }  // :23-24
template <typename T> inline V3<T> muls(V3<T> a, T s) { return {a.x * s, a.y * s, a.z * s}; }            // :25
template <typename T> inline T dot(V3<T> a, V3<T> b) { return a.x * b.x + a.y * b.y + a.z * b.z; }       // :30-31
template <typename T> inline T sq(T v) { return std::pow(v, T(2)); }                                      // `**2`
template <typename T> inline V3<T> normalize(V3<T> a) {                                                   // :33-35
  T len = std::sqrt(sq(a.x) + sq(a.y) + sq(a.z));
  return {a.x / len, a.y / len, a.z / len};
}

template <typename T>
struct SceneT {
  int n = 0;
  std::vector<V3<T>> center, color;   // center: Sphere.center, or Cube.min_point (:58)
  std::vector<V3<T>> box_max;         // Cube.max_point (:59); unused for spheres
  std::vector<int> kind;              // 0 = Sphere (:50-54), 1 = Cube (:56-60): the isinstance() of :157, :164
  std::vector<T> radius, specular, reflection, refraction, refractive_index;
  V3<T> p_pos, p_col; T p_int;                       // scene['global_light'], :79
  V3<T> a_pos, a_u, a_v, a_col; T a_int; int a_samples;  // scene['area_light'], :80
  V3<T> cam_o, cam_r, cam_u, cam_f;
};

// ---- keyed RNG (definition: oracle/ref_harness.py docstring) -------------------------------
inline uint32_t fmix32(uint32_t h) {
  h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
  return h;
}
inline uint32_t mix(uint32_t h, uint32_t v) { return fmix32(h + 0x9E3779B9u + v); }
inline uint32_t pixel_key_of(uint32_t seed32, uint32_t x, uint32_t y, uint32_t sample) {
  return fmix32(seed32 + x * 0x9E3779B9u + y * 0x85EBCA6Bu + sample * 0xC2B2AE35u);
}
inline uint32_t pair_hash(uint32_t key, uint32_t j) {
  uint32_t h = key + j * 0x9E3779B9u + 0x7F4A7C15u;
  h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
  return h;
}
struct Rng {
  int mode;           // 0 = constant 0.5 (mode C), 1 = keyed (mode K)
  uint32_t pixel_key;
  // draw i of level `level` (0 = the jitter draws of render_pixel: the pixel's own key): one hash per PAIR of
  // 16-bit draws
  double draw(uint32_t level, uint32_t i) const {
    if (mode == 0) return 0.5;
    const uint32_t key = level == 0 ? pixel_key : mix(pixel_key, level);
    const uint32_t h = pair_hash(key, i >> 1);
    return double((i & 1u) ? (h & 0xFFFFu) : (h >> 16)) * (1.0 / 65536.0);
  }
};

struct PixelStats {
  int primary_hit = -1;
  int64_t trace_levels = 0, hit_levels = 0;
  double margin = std::numeric_limits<double>::infinity();
};

// intersect_sphere, raytracer.py:93-104. Returns false for `None`.
template <typename T>
inline bool intersect_sphere(V3<T> o, V3<T> d, V3<T> c, T radius, T* t_out, double* margin) {
  V3<T> oc = sub(o, c);                       // :94
  T a = dot(d, d);                            // :95
  T b = T(2) * dot(oc, d);                    // :96
  T cc = dot(oc, oc) - sq(radius);            // :97
  T disc = sq(b) - T(4) * a * cc;             // :98
  if (margin) {
    double m = std::fabs(double(disc)) / std::fmax(1.0, double(b) * double(b));
    if (m < *margin) *margin = m;
  }
  if (disc < 0) return false;                 // :99-100
  T t = (-b - std::sqrt(disc)) / (T(2) * a);  // :101
  if (t < 0) return false;                    // :102-103
  *t_out = t;
  return true;
}

// intersect_cube, raytracer.py:106-137. Returns false for `None`; *t_out may be negative or zero (the
// reference returns t_max unchecked when t_min <= 0, :137 — a box wholly behind the origin yields t < 0,
// which find_nearest_intersection's `if t and t < min_distance` (:162) ACCEPTS).
template <typename T>
inline bool intersect_cube(V3<T> o, V3<T> d, V3<T> lo, V3<T> hi, T* t_out, double* margin = nullptr) {
  T t_min = -std::numeric_limits<T>::infinity();          // :107
  T t_max = std::numeric_limits<T>::infinity();           // :108
  const T mins[3] = {lo.x, lo.y, lo.z}, maxs[3] = {hi.x, hi.y, hi.z};
  const T org[3] = {o.x, o.y, o.z}, dir[3] = {d.x, d.y, d.z};
  if (margin) {   // diagnostic only (the sphere margin's counterpart): how close the slab intervals come to
                  // "just touching" (a silhouette edge of the box), relative to the distances involved
    double lo_all = -std::numeric_limits<double>::infinity(), hi_all = std::numeric_limits<double>::infinity();
    for (int i = 0; i < 3; ++i) {
      if (std::fabs(dir[i]) < T(1e-8)) continue;
      double t1 = double((mins[i] - org[i]) / dir[i]), t2 = double((maxs[i] - org[i]) / dir[i]);
      if (t1 > t2) { double tmp = t1; t1 = t2; t2 = tmp; }
      if (t1 > lo_all) lo_all = t1;
      if (t2 < hi_all) hi_all = t2;
    }
    if (std::isfinite(lo_all) && std::isfinite(hi_all)) {
      double m = std::fabs(hi_all - lo_all) / std::fmax(1.0, std::fmax(std::fabs(lo_all), std::fabs(hi_all)));
      if (m < *margin) *margin = m;
    }
  }
  for (int i = 0; i < 3; ++i) {                           // :110-122
    if (std::fabs(dir[i]) < T(1e-8)) {                    // :124
      if (org[i] < mins[i] || org[i] > maxs[i]) return false;   // :125-126
    } else {
      T t1 = (mins[i] - org[i]) / dir[i];                 // :128
      T t2 = (maxs[i] - org[i]) / dir[i];                 // :129
      if (t1 > t2) { T tmp = t1; t1 = t2; t2 = tmp; }     // :130-131
      if (t1 > t_min) t_min = t1;                         // :132  max(t_min, t1)
      if (t2 < t_max) t_max = t2;                         // :133  min(t_max, t2)
      if (t_min > t_max) return false;                    // :134-135
    }
  }
  *t_out = t_min > 0 ? t_min : t_max;                     // :137
  return true;
}

// cube_normal, raytracer.py:142-150
template <typename T>
inline V3<T> cube_normal(V3<T> lo, V3<T> hi, V3<T> p) {
  const T eps = T(1e-5);                                  // :143
  if (std::fabs(p.x - lo.x) < eps) return {-1, 0, 0};     // :144
  if (std::fabs(p.x - hi.x) < eps) return {1, 0, 0};      // :145
  if (std::fabs(p.y - lo.y) < eps) return {0, -1, 0};     // :146
  if (std::fabs(p.y - hi.y) < eps) return {0, 1, 0};      // :147
  if (std::fabs(p.z - lo.z) < eps) return {0, 0, -1};     // :148
  if (std::fabs(p.z - hi.z) < eps) return {0, 0, 1};      // :149
  return {0, 1, 0};                                       // :150
}

template <typename T>
struct Hit { int idx; V3<T> point, normal; };

// find_nearest_intersection, raytracer.py:152-172: :156 iterates scene['spheres'], whose members are
// dispatched on their type (:157-160, :164-169), so a Cube placed in that list is traced.
template <typename T>
inline bool find_nearest(const SceneT<T>& s, V3<T> o, V3<T> d, Hit<T>* h, double* margin) {
  bool found = false;
  T min_distance = std::numeric_limits<T>::infinity();   // :154
  T second = std::numeric_limits<T>::infinity();
  for (int i = 0; i < s.n; ++i) {                         // :156
    T t;
    if (s.kind[i] == 1) {                                 // :159-160
      if (!intersect_cube(o, d, s.center[i], s.box_max[i], &t, margin)) continue;
      if (margin && t != T(0)) { if (t < min_distance) second = min_distance; else if (t < second) second = t; }
      if (t != T(0) && t < min_distance) {                // :162
        min_distance = t;
        V3<T> p = add(o, muls(d, t));                     // :168
        h->idx = i; h->point = p;
        h->normal = cube_normal(s.center[i], s.box_max[i], p);   // :169
        found = true;
      }
      continue;
    }
    if (!intersect_sphere(o, d, s.center[i], s.radius[i], &t, margin)) continue;
    if (margin && t != T(0)) { if (t < min_distance) second = min_distance; else if (t < second) second = t; }
    if (t != T(0) && t < min_distance) {                  // :162  `if t and t < min_distance`
      min_distance = t;
      V3<T> p = add(o, muls(d, t));                       // :165
      h->idx = i; h->point = p;
      h->normal = normalize(sub(p, s.center[i]));         // :166, :139-140
      found = true;
    }
  }
  if (margin && found && std::isfinite(double(second))) {
    double gap = (double(second) - double(min_distance)) / std::fmax(1e-30, std::fabs(double(min_distance)));
    if (gap < *margin) *margin = gap;
  }
  return found;
}

template <typename T>
inline T max0(T v) { return v > 0 ? v : T(0); }           // max(0, v)

// compute_lighting, raytracer.py:174-211
template <typename T>
V3<T> compute_lighting(const SceneT<T>& s, V3<T> point, V3<T> normal, V3<T> view, int mat,
                       const Rng& rng, uint32_t level) {
  V3<T> color{0, 0, 0};
  Hit<T> sh;
  T spec_k = s.specular[mat];
  // --- scene['global_light'] (point light), :197-209 ---
  {
    V3<T> ldir = normalize(sub(s.p_pos, point));                          // :198
    V3<T> so = add(point, muls(normal, T(0.001)));                        // :199
    if (!find_nearest(s, so, ldir, &sh, (double*)nullptr) || s.refraction[sh.idx] > 0) {   // :200-202
      T diffuse = max0(dot(normal, ldir));                                // :203
      color = add(color, muls(muls(mulv(s.color[mat], s.p_col), s.p_int), diffuse));  // :204
      if (spec_k > 0) {                                                   // :206
        V3<T> r = sub(ldir, muls(normal, T(2) * dot(ldir, normal)));      // :207
        T sp = std::pow(max0(dot(view, r)), T(50));                       // :208
        color = add(color, muls(muls(muls(s.p_col, s.p_int), spec_k), sp));  // :209
      }
    }
  }
  // --- scene['area_light'], :177-196 ---
  {
    V3<T> contrib{0, 0, 0};
    for (int k = 0; k < s.a_samples; ++k) {                               // :179
      T ru = T(rng.draw(level, 2 * k) - 0.5);                             // :180
      T rv = T(rng.draw(level, 2 * k + 1) - 0.5);                         // :181
      V3<T> lp = add(add(s.a_pos, muls(s.a_u, ru)), muls(s.a_v, rv));     // :182
      V3<T> ldir = normalize(sub(lp, point));                             // :183
      V3<T> so = add(point, muls(normal, T(0.001)));                      // :184
      if (!find_nearest(s, so, ldir, &sh, (double*)nullptr) || s.refraction[sh.idx] > 0) {   // :185-187
        T diffuse = max0(dot(normal, ldir));                              // :188
        contrib = add(contrib, muls(muls(s.a_col, s.a_int), diffuse));    // :189
        if (spec_k > 0) {                                                 // :191
          V3<T> r = sub(ldir, muls(normal, T(2) * dot(ldir, normal)));    // :192
          T sp = std::pow(max0(dot(view, r)), T(50));                     // :193
          contrib = add(contrib, muls(muls(muls(s.a_col, s.a_int), spec_k), sp));  // :194
        }
      }
    }
    color = add(color, muls(contrib, T(1) / T(s.a_samples)));             // :196
  }
  return color;                                                           // :211
}

template <typename T>
V3<T> trace_ray(const SceneT<T>& s, V3<T> o, V3<T> d, int level, int D, const Rng& rng, PixelStats* st);

// compute_refraction, raytracer.py:213-230
template <typename T>
V3<T> compute_refraction(const SceneT<T>& s, V3<T> d, V3<T> hit_point, V3<T> normal, int mat, int level, int D,
                         const Rng& rng, PixelStats* st) {
  T n1 = T(1);                                                            // :214
  T n2 = s.refractive_index[mat];                                         // :215
  T cos_i = -dot(normal, d);                                              // :217
  if (cos_i < 0) {                                                        // :218-221
    cos_i = -cos_i;
    normal = muls(normal, T(-1));
    T tmp = n1; n1 = n2; n2 = tmp;
  }
  T n = n1 / n2;                                                          // :223
  T k = T(1) - n * n * (T(1) - cos_i * cos_i);                            // :224
  if (k < 0) return {0, 0, 0};                                            // :225-226 total internal reflection
  V3<T> rd = normalize(add(muls(d, n), muls(normal, n * cos_i - std::sqrt(k))));   // :228
  V3<T> ro = sub(hit_point, muls(normal, T(0.001)));                      // :229
  return trace_ray(s, ro, rd, level + 1, D, rng, st);                     // :230
}

// trace_ray, raytracer.py:232-260, with the literal depth limit 5 (:233) made a parameter D:
// `level > D` returns black.
template <typename T>
V3<T> trace_ray(const SceneT<T>& s, V3<T> o, V3<T> d, int level, int D, const Rng& rng, PixelStats* st) {
  if (level > D) return {0, 0, 0};                                        // :233-234
  st->trace_levels++;
  Hit<T> h;
  if (!find_nearest(s, o, d, &h, level == 0 ? &st->margin : nullptr)) {   // :236-238
    return {0, 0, 0};
  }
  if (level == 0) st->primary_hit = h.idx;
  V3<T> color = s.color[h.idx];                                           // :242
  T refl = s.reflection[h.idx];
  if (refl > 0) {                                                         // :245
    V3<T> rd = sub(d, muls(h.normal, T(2) * dot(d, h.normal)));           // :246
    V3<T> ro = add(h.point, muls(h.normal, T(0.001)));                    // :247
    V3<T> rc = trace_ray(s, ro, rd, level + 1, D, rng, st);               // :248
    color = add(muls(color, T(1) - refl), muls(rc, refl));                // :249
  }
  T refr = s.refraction[h.idx];
  if (refr > 0) {                                                         // :252
    V3<T> fc = compute_refraction(s, d, h.point, h.normal, h.idx, level, D, rng, st);   // :253
    color = add(muls(color, T(1) - refr), muls(fc, refr));                // :254
  }
  st->hit_levels++;
  V3<T> light = compute_lighting(s, h.point, h.normal, muls(d, T(-1)), h.idx, rng,
                                 uint32_t(level + 1));                    // :257
  return mulv(color, light);                                              // :258
}

// render_pixel, raytracer.py:289-297 (camera generalised: see raytracer_amd/scene.py Camera)
template <typename T>
V3<T> render_pixel(const SceneT<T>& s, int x, int y, int width, int height, int samples, int D,
                   int mode, uint32_t seed32, PixelStats* st) {
  T aspect = T(width) / T(height);                                        // :290
  V3<T> color{0, 0, 0};
  for (int k = 0; k < samples; ++k) {                                     // :292
    Rng rng{mode, pixel_key_of(seed32, uint32_t(x), uint32_t(y), uint32_t(k))};
    T u = ((T(x) + T(rng.draw(0, 0))) / T(width)) * T(2) - T(1);          // :293
    T v = -((T(y) + T(rng.draw(0, 1))) / T(height) * T(2) - T(1)) / aspect;  // :294
    V3<T> dir = normalize(add(add(muls(s.cam_r, u), muls(s.cam_u, v)), s.cam_f));  // :295
    st->margin = std::numeric_limits<double>::infinity();
    st->primary_hit = -1;
    color = add(color, trace_ray(s, s.cam_o, dir, 0, D, rng, st));        // :296
  }
  return muls(color, T(1) / T(samples));                                  // :297
}

template <typename T>
void build_scene(SceneT<T>& s, int n, const int32_t* kind, const float* box_max, const float* center,
                 const float* radius, const float* color,
                 const float* specular, const float* reflection, const float* refraction,
                 const float* refractive_index, const float* lights, int area_samples, const float* camera) {
  s.n = n;
  s.center.resize(n); s.color.resize(n); s.radius.resize(n); s.specular.resize(n); s.reflection.resize(n);
  s.refraction.assign(n, T(0)); s.refractive_index.assign(n, T(1));   // Material defaults, :43
  s.kind.assign(n, 0); s.box_max.assign(n, V3<T>{0, 0, 0});
  for (int i = 0; i < n; ++i) {
    if (kind) s.kind[i] = kind[i];
    if (box_max) s.box_max[i] = {T(box_max[3 * i]), T(box_max[3 * i + 1]), T(box_max[3 * i + 2])};
  }
  for (int i = 0; i < n; ++i) {
    if (refraction) s.refraction[i] = T(refraction[i]);
    if (refractive_index) s.refractive_index[i] = T(refractive_index[i]);
  }
  for (int i = 0; i < n; ++i) {
    s.center[i] = {T(center[3 * i]), T(center[3 * i + 1]), T(center[3 * i + 2])};
    s.color[i] = {T(color[3 * i]), T(color[3 * i + 1]), T(color[3 * i + 2])};
    s.radius[i] = T(radius[i]); s.specular[i] = T(specular[i]); s.reflection[i] = T(reflection[i]);
  }
  auto v3 = [](const float* p) { return V3<T>{T(p[0]), T(p[1]), T(p[2])}; };
  s.p_pos = v3(lights); s.p_col = v3(lights + 3); s.p_int = T(lights[6]);
  s.a_pos = v3(lights + 7); s.a_u = v3(lights + 10); s.a_v = v3(lights + 13);
  s.a_col = v3(lights + 16); s.a_int = T(lights[19]); s.a_samples = area_samples;
  s.cam_o = v3(camera); s.cam_r = v3(camera + 3); s.cam_u = v3(camera + 6); s.cam_f = v3(camera + 9);
}

template <typename T>
int run(int n, const int32_t* kind, const float* box_max, const float* center, const float* radius, const float* color, const float* specular,
        const float* reflection, const float* refraction, const float* refractive_index,
        const float* lights, int area_samples, const float* camera,
        int width, int height, const int32_t* pixels, int64_t n_pixels, int samples, int D, int mode,
        uint64_t seed, int threads, double* out_color, uint8_t* out_rgb8, int32_t* out_hit,
        int64_t* out_trace_levels, int64_t* out_hit_levels, double* out_margin) {
  SceneT<T> s;
  build_scene(s, n, kind, box_max, center, radius, color, specular, reflection, refraction, refractive_index, lights,
              area_samples, camera);
  uint32_t seed32 = uint32_t(seed ^ (seed >> 32));
  if (!pixels) n_pixels = int64_t(width) * height;
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
#endif
#pragma omp parallel for schedule(dynamic, 64)
  for (int64_t k = 0; k < n_pixels; ++k) {
    int x = pixels ? pixels[2 * k] : int(k % width);
    int y = pixels ? pixels[2 * k + 1] : int(k / width);
    PixelStats st;
    V3<T> c = render_pixel(s, x, y, width, height, samples, D, mode, seed32, &st);
    T ch[3] = {c.x, c.y, c.z};
    for (int j = 0; j < 3; ++j) {
      if (out_color) out_color[3 * k + j] = double(ch[j]);
      T cl = ch[j] < 0 ? T(0) : (ch[j] > 1 ? T(1) : ch[j]);          // render(), :318
      if (out_rgb8) out_rgb8[3 * k + j] = uint8_t(cl * T(255));       // :325 (truncation)
    }
    if (out_hit) out_hit[k] = st.primary_hit;
    if (out_trace_levels) out_trace_levels[k] = st.trace_levels;
    if (out_hit_levels) out_hit_levels[k] = st.hit_levels;
    if (out_margin) out_margin[k] = st.margin;
  }
  return 0;
}

}  // namespace

#ifdef _OPENMP
static const int g_threads_at_start = omp_get_max_threads();   // at library load
#endif

extern "C" {

// precision: 0 = double (the oracle), 1 = float (diagnostic).  pixels = NULL renders the full
// frame row-major.  Outputs may be NULL.  `hit`/`margin` refer to the LAST sample's primary ray.
// refraction / refractive_index: per sphere, or NULL for the Material defaults 0 / 1 (raytracer.py:43).
// kind / box_max: per object of the list the reference scans (:156), in list order: kind 0 = Sphere
// (center, radius), 1 = Cube (center holds min_point, box_max max_point); NULL = all spheres.
int rto_render_objects(int n, const int32_t* kind, const float* box_max, const float* center,
                       const float* radius, const float* color,
                       const float* specular, const float* reflection, const float* refraction,
                       const float* refractive_index, const float* lights,
                       int area_samples, const float* camera, int width, int height,
                       const int32_t* pixels, int64_t n_pixels, int samples, int max_depth, int mode,
                       uint64_t seed, int precision, int threads, double* out_color,
                       uint8_t* out_rgb8, int32_t* out_hit, int64_t* out_trace_levels,
                       int64_t* out_hit_levels, double* out_margin) {
  if (n < 0 || width <= 0 || height <= 0 || samples <= 0 || area_samples <= 0) return -1;
  if (precision == 0)
    return run<double>(n, kind, box_max, center, radius, color, specular, reflection, refraction, refractive_index, lights,
                       area_samples, camera, width, height, pixels, n_pixels, samples, max_depth, mode, seed,
                       threads, out_color, out_rgb8, out_hit, out_trace_levels, out_hit_levels, out_margin);
  return run<float>(n, kind, box_max, center, radius, color, specular, reflection, refraction, refractive_index, lights,
                    area_samples, camera, width, height, pixels, n_pixels, samples, max_depth, mode, seed,
                    threads, out_color, out_rgb8, out_hit, out_trace_levels, out_hit_levels, out_margin);
}

int rto_render_pixels2(int n, const float* center, const float* radius, const float* color,
                       const float* specular, const float* reflection, const float* refraction,
                       const float* refractive_index, const float* lights,
                       int area_samples, const float* camera, int width, int height,
                       const int32_t* pixels, int64_t n_pixels, int samples, int max_depth, int mode,
                       uint64_t seed, int precision, int threads, double* out_color,
                       uint8_t* out_rgb8, int32_t* out_hit, int64_t* out_trace_levels,
                       int64_t* out_hit_levels, double* out_margin) {
  return rto_render_objects(n, nullptr, nullptr, center, radius, color, specular, reflection, refraction,
                            refractive_index, lights, area_samples, camera, width, height, pixels, n_pixels,
                            samples, max_depth, mode, seed, precision, threads, out_color, out_rgb8, out_hit,
                            out_trace_levels, out_hit_levels, out_margin);
}

int rto_render_pixels(int n, const float* center, const float* radius, const float* color,
                      const float* specular, const float* reflection, const float* lights,
                      int area_samples, const float* camera, int width, int height,
                      const int32_t* pixels, int64_t n_pixels, int samples, int max_depth, int mode,
                      uint64_t seed, int precision, int threads, double* out_color,
                      uint8_t* out_rgb8, int32_t* out_hit, int64_t* out_trace_levels,
                      int64_t* out_hit_levels, double* out_margin) {
  return rto_render_pixels2(n, center, radius, color, specular, reflection, nullptr, nullptr, lights,
                            area_samples, camera, width, height, pixels, n_pixels, samples, max_depth, mode,
                            seed, precision, threads, out_color, out_rgb8, out_hit, out_trace_levels,
                            out_hit_levels, out_margin);
}

int rto_max_threads(void) {
#ifdef _OPENMP
  // the team size OpenMP started with (OMP_NUM_THREADS or the processors), not what an earlier
  // render's omp_set_num_threads left behind
  return g_threads_at_start;
#else
  return 1;
#endif
}

}  // extern "C"
