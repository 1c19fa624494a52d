// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_math.h header).
// Camera (rc/camera.rs), filters (rc/filter/*.rs), Film/FilmTile (rc/film.rs), PathIntegrator
// (rc/integrator/path.rs, rc/integrator/mod.rs:186-318) and the render driver (rc/renderer.rs).
#pragma once
#include <mutex>
#include <vector>
#include "orc_scene.h"

namespace orc {

// ---------------------------------------------------------------- filters
enum FilterKind { FILTER_BOX = 0, FILTER_TRIANGLE = 1, FILTER_GAUSSIAN = 2, FILTER_MITCHELL = 3 };
struct Filter {
  int kind = FILTER_BOX;
  float xw = 0.5f, yw = 0.5f, a = 2.0f, b = 1.0f / 3.0f;  // a: gaussian alpha | mitchell B ; b: mitchell C
  float mitchell_1d(float x) const {  // filter/mitchell.rs:25-41
    float B = a, C = b;
    float fx = fabsf(x) * 2.0f;
    if (fx < 1.0f) return ((12.0f - 9.0f * B - 6.0f * C) * fx * fx * fx + (-18.0f + 12.0f * B + 6.0f * C) * fx * fx + (6.0f - 2.0f * B)) * (1.0f / 6.0f);
    if (fx < 2.0f) return ((-B - 6.0f * C) * fx * fx * fx + (6.0f * B + 30.0f * C) * fx * fx + (-12.0f * B - 48.0f * C) * fx + (8.0f * B + 24.0f * C)) * (1.0f / 6.0f);
    return 0.0f;
  }
  float evaluate(float x, float y) const {
    switch (kind) {
      case FILTER_BOX: return 1.0f;                                                                      // boxfilter.rs:26-28
      case FILTER_TRIANGLE: return fmaxf(0.0f, xw - fabsf(x)) * fmaxf(0.0f, yw - fabsf(y));              // triangle.rs:29-31
      case FILTER_GAUSSIAN: {                                                                            // gaussian.rs:15-38
        float expx = expf(-a * xw * xw), expy = expf(-a * yw * yw);
        return fmaxf(expf(-a * x * x) - expx, 0.0f) * fmaxf(expf(-a * y * y) - expy, 0.0f);
      }
      default: return mitchell_1d(x * (1.0f / xw)) * mitchell_1d(y * (1.0f / yw));                       // mitchell.rs:54-56
    }
  }
};

struct B2i { int x0, y0, x1, y1; };  // [p_min, p_max)
inline bool b2i_inside_exclusive(const B2i& b, int x, int y) { return x >= b.x0 && x < b.x1 && y >= b.y0 && y < b.y1; }  // bounds.rs:73-75

// ---------------------------------------------------------------- Film (rc/film.rs)
struct FilmPixel { float xyz[3]; float weight; };
struct Film {
  int xres = 0, yres = 0;
  B2i cropped{0, 0, 0, 0};
  float filter_table[256];
  float radius_x = 0.5f, radius_y = 0.5f;
  float scale = 1.0f, max_sample_luminance = kInf;
  std::vector<FilmPixel> pixels;
  std::mutex mtx;

  void init(int xr, int yr, const float crop[4], const Filter& f, float scale_, float max_lum) {  // :58-115
    xres = xr; yres = yr; scale = scale_; max_sample_luminance = max_lum;
    int ax = f2i_sat(ceilf((float)xr * crop[0])), ay = f2i_sat(ceilf((float)yr * crop[2]));
    int bx = f2i_sat(ceilf((float)xr * crop[1])), by = f2i_sat(ceilf((float)yr * crop[3]));
    cropped = B2i{min_po(ax, bx), min_po(ay, by), max_po(ax, bx), max_po(ay, by)};
    pixels.assign((size_t)(cropped.x1 - cropped.x0) * (cropped.y1 - cropped.y0), FilmPixel{{0, 0, 0}, 0});
    radius_x = f.xw; radius_y = f.yw;
    for (int y = 0; y < 16; ++y) {
      float fy = ((float)y + 0.5f) * (f.yw / 16.0f);
      for (int x = 0; x < 16; ++x) {
        float fx = ((float)x + 0.5f) * (f.xw / 16.0f);
        filter_table[y * 16 + x] = f.evaluate(fx, fy);
      }
    }
  }
  B2i sample_bounds() const {  // :249-257
    float x0 = floorf((float)cropped.x0 + 0.5f - radius_x), y0 = floorf((float)cropped.y0 + 0.5f - radius_y);
    float x1 = ceilf((float)cropped.x1 - 0.5f + radius_x), y1 = ceilf((float)cropped.y1 - 0.5f + radius_y);
    // Bounds2f::from_points then `as i32`
    float mnx = min_po(x0, x1), mny = min_po(y0, y1), mxx = max_po(x0, x1), mxy = max_po(y0, y1);
    return B2i{f2i_sat(mnx), f2i_sat(mny), f2i_sat(mxx), f2i_sat(mxy)};
  }
};
struct FilmTile {  // :268-376
  B2i pb;
  float rx, ry, inv_rx, inv_ry, max_lum;
  const float* table;
  std::vector<RGB> contrib; std::vector<float> wsum;
  void init(const Film& film, const B2i& sb) {  // get_film_tile :152-175
    float p0x = ceilf((float)sb.x0 - 0.5f - film.radius_x), p0y = ceilf((float)sb.y0 - 0.5f - film.radius_y);
    float p1x = floorf((float)sb.x1 - 0.5f + film.radius_x + 1.0f), p1y = floorf((float)sb.y1 - 0.5f + film.radius_y + 1.0f);
    float ex0 = min_po(p0x, p1x), ey0 = min_po(p0y, p1y), ex1 = max_po(p0x, p1x), ey1 = max_po(p0y, p1y);
    // Bounds2f::intersect with the cropped pixel bounds, then `as i32` through from_points
    float ix0 = max_po(ex0, (float)film.cropped.x0), iy0 = max_po(ey0, (float)film.cropped.y0);
    float ix1 = min_po(ex1, (float)film.cropped.x1), iy1 = min_po(ey1, (float)film.cropped.y1);
    int a0 = f2i_sat(ix0), b0 = f2i_sat(iy0), a1 = f2i_sat(ix1), b1 = f2i_sat(iy1);
    pb = B2i{min_po(a0, a1), min_po(b0, b1), max_po(a0, a1), max_po(b0, b1)};
    rx = film.radius_x; ry = film.radius_y; inv_rx = 1.0f / rx; inv_ry = 1.0f / ry;
    max_lum = film.max_sample_luminance; table = film.filter_table;
    int w = pb.x1 - pb.x0, h = pb.y1 - pb.y0;
    size_t n = (w > 0 && h > 0) ? (size_t)w * h : 0;
    contrib.assign(n, rgb(0, 0, 0)); wsum.assign(n, 0.0f);
  }
  void add_sample(float pfx, float pfy, RGB colour) {  // :298-361
    if (has_nan(colour)) return;
    RGB L = lum_y(colour) > max_lum ? colour * max_lum / lum_y(colour) : colour;
    float dx = pfx - 0.5f, dy = pfy - 0.5f;
    float p0x = ceilf(dx - rx), p0y = ceilf(dy - ry);
    float p1x = floorf(dx + rx + 1.0f), p1y = floorf(dy + ry + 1.0f);
    float ix0 = max_po(min_po(p0x, p1x), (float)pb.x0), iy0 = max_po(min_po(p0y, p1y), (float)pb.y0);
    float ix1 = min_po(max_po(p0x, p1x), (float)pb.x1), iy1 = min_po(max_po(p0y, p1y), (float)pb.y1);
    int a0 = f2i_sat(ix0), b0 = f2i_sat(iy0), a1 = f2i_sat(ix1), b1 = f2i_sat(iy1);
    int x0 = min_po(a0, a1), y0 = min_po(b0, b1), x1 = max_po(a0, a1), y1 = max_po(b0, b1);
    int ifx[64], ify[64];
    for (int x = x0; x < x1 && x - x0 < 64; ++x) {
      float fx = fabsf(((float)x - dx) * inv_rx * 16.0f);
      ifx[x - x0] = (int)f2u_sat(fminf(floorf(fx), 16.0f - 1.0f));
    }
    for (int y = y0; y < y1 && y - y0 < 64; ++y) {
      float fy = fabsf(((float)y - dy) * inv_ry * 16.0f);
      ify[y - y0] = (int)f2u_sat(fminf(floorf(fy), 16.0f - 1.0f));
    }
    int w = pb.x1 - pb.x0;
    for (int y = y0; y < y1; ++y)
      for (int x = x0; x < x1; ++x) {
        float fw = table[ify[y - y0] * 16 + ifx[x - x0]];
        size_t i = (size_t)(y - pb.y0) * w + (x - pb.x0);
        contrib[i] = contrib[i] + L * fw;
        wsum[i] += fw;
      }
  }
};
inline void film_merge_tile(Film& film, const FilmTile& t) {  // :177-194
  std::lock_guard<std::mutex> lk(film.mtx);
  int fw = film.cropped.x1 - film.cropped.x0, tw = t.pb.x1 - t.pb.x0;
  for (int y = t.pb.y0; y < t.pb.y1; ++y)
    for (int x = t.pb.x0; x < t.pb.x1; ++x) {
      size_t ti = (size_t)(y - t.pb.y0) * tw + (x - t.pb.x0);
      size_t pi = (size_t)(y - film.cropped.y0) * fw + (x - film.cropped.x0);
      float xyz[3]; to_xyz(t.contrib[ti], xyz);
      film.pixels[pi].xyz[0] += xyz[0]; film.pixels[pi].xyz[1] += xyz[1]; film.pixels[pi].xyz[2] += xyz[2];
      film.pixels[pi].weight += t.wsum[ti];
    }
}
// write_image math, :196-234 (no splats)
inline RGB film_pixel_rgb(const FilmPixel& p, float scale) {
  RGB c = from_xyz(p.xyz);
  if (p.weight != 0.0f) {
    float inv = 1.0f / p.weight;
    c = rgb(fmaxf(0.0f, c.r * inv), fmaxf(0.0f, c.g * inv), fmaxf(0.0f, c.b * inv));
  }
  float z[3] = {0, 0, 0}; RGB splat = from_xyz(z);
  c = rgb(c.r + 1.0f * splat.r, c.g + 1.0f * splat.g, c.b + 1.0f * splat.b);
  return rgb(c.r * scale, c.g * scale, c.b * scale);
}

// ---------------------------------------------------------------- PerspectiveCamera (rc/camera.rs)
struct Camera {
  Transform camera_to_world, raster_to_camera;
  float lens_radius = 0, focal_distance = 1e6f;
  V3 dx_camera, dy_camera;
  void init(const Transform& c2w, const float sw[4] /*xmin,xmax,ymin,ymax*/, float lensr, float focald, float fov, int xres, int yres) {  // :30-72
    camera_to_world = c2w; lens_radius = lensr; focal_distance = focald;
    Transform camera_to_screen = xf_perspective(fov, 1e-2f, 1000.0f);
    Transform screen_to_raster = xf_mul(xf_mul(xf_scale((float)xres, (float)yres, 1.0f), xf_scale(1.0f / (sw[1] - sw[0]), 1.0f / (sw[2] - sw[3]), 1.0f)),
                                        xf_translate(v3(-sw[0], -sw[3], 0.0f)));
    Transform raster_to_screen = xf_inverse(screen_to_raster);
    raster_to_camera = xf_mul(xf_inverse(camera_to_screen), raster_to_screen);
    dx_camera = xf_point(raster_to_camera.m, v3(1, 0, 0)) - xf_point(raster_to_camera.m, v3(0, 0, 0));
    dy_camera = xf_point(raster_to_camera.m, v3(0, 1, 0)) - xf_point(raster_to_camera.m, v3(0, 0, 0));
  }
  static void default_screen_window(int xres, int yres, float sw[4]) {  // create(), :86-97
    float frame = (float)xres / (float)yres;
    if (frame > 1.0f) { sw[0] = -frame; sw[1] = frame; sw[2] = -1.0f; sw[3] = 1.0f; }
    else { sw[0] = -1.0f; sw[1] = 1.0f; sw[2] = -1.0f / frame; sw[3] = 1.0f / frame; }
  }
  // Ray::transform, ray.rs:46-71 (returns only the ray)
  Ray transform_ray(const Ray& r) const {
    const M44& m = camera_to_world.m;
    V3 o = xf_point(m, r.o);
    float x = r.o.x, y = r.o.y, z = r.o.z;  // transform_point error, transform.rs:175-188
    V3 o_error = gamma_n(3) * v3(fabsf(m.m[0][0] * x) + fabsf(m.m[0][1] * y) + fabsf(m.m[0][2] * z) + fabsf(m.m[0][3]),
                                 fabsf(m.m[1][0] * x) + fabsf(m.m[1][1] * y) + fabsf(m.m[1][2] * z) + fabsf(m.m[1][3]),
                                 fabsf(m.m[2][0] * x) + fabsf(m.m[2][1] * y) + fabsf(m.m[2][2] * z) + fabsf(m.m[2][3]));
    V3 d = xf_vector(m, r.d);
    float l2 = length_squared(d);
    if (l2 > 0.0f) { float dt = dot(vabs(d), o_error) / l2; o = o + d * dt; }
    Ray out = r; out.o = o; out.d = d;
    if (r.has_diff) {
      out.rx_o = xf_point(m, r.rx_o); out.ry_o = xf_point(m, r.ry_o);
      out.rx_d = xf_vector(m, r.rx_d); out.ry_d = xf_vector(m, r.ry_d);
    }
    return out;
  }
  Ray generate_ray_differential(P2 p_film, P2 p_lens) const {  // :150-202
    V3 p_camera = xf_point(raster_to_camera.m, v3(p_film.x, p_film.y, 0.0f));
    Ray ray = ray_new(v3(0, 0, 0), normalize(p_camera));
    if (lens_radius > 0.0f) {
      P2 pl = concentric_sample_disk(p_lens); pl.x = lens_radius * pl.x; pl.y = lens_radius * pl.y;
      float ft = focal_distance / ray.d.z;
      V3 p_focus = ray.o + ft * ray.d;
      ray.o = v3(pl.x, pl.y, 0.0f);
      ray.d = normalize(p_focus - ray.o);
    }
    if (lens_radius > 0.0f) {
      P2 pl = concentric_sample_disk(p_lens); pl.x = lens_radius * pl.x; pl.y = lens_radius * pl.y;
      V3 origin = v3(pl.x, pl.y, 0.0f);
      V3 dx = normalize(p_camera + dx_camera);
      float ft_x = focal_distance / dx.z;
      V3 p_focus_x = ft_x * dx;
      V3 dy = normalize(p_camera + dy_camera);
      float ft_y = focal_distance / dy.z;
      V3 p_focus_y = ft_y * dy;
      ray.rx_o = origin; ray.ry_o = origin;
      ray.rx_d = normalize(p_focus_x - origin); ray.ry_d = normalize(p_focus_y - origin);
    } else {
      ray.rx_o = ray.o; ray.ry_o = ray.o;
      ray.rx_d = normalize(p_camera + dx_camera); ray.ry_d = normalize(p_camera + dy_camera);
    }
    ray.has_diff = true;
    return transform_ray(ray);
  }
};
inline void scale_differentials(Ray& r, float s) {  // ray.rs:73-80
  if (!r.has_diff) return;
  r.rx_o = r.o + (r.rx_o - r.o) * s; r.ry_o = r.o + (r.ry_o - r.o) * s;
  r.rx_d = r.d + (r.rx_d - r.d) * s; r.ry_d = r.d + (r.ry_d - r.d) * s;
}

// ---------------------------------------------------------------- PathIntegrator
struct PathStats {
  uint64_t camera_rays = 0, zero_radiance = 0, nee_total = 0, path_len_sum = 0, scrubbed = 0, pdf_wi_tests = 0;
  TraceCounters closest, shadow, mis;  // by ray class: path-continuation closest hits, shadow any-hits, MIS closest hits
  void add(const PathStats& o) {
    camera_rays += o.camera_rays; zero_radiance += o.zero_radiance; nee_total += o.nee_total; path_len_sum += o.path_len_sum;
    scrubbed += o.scrubbed; pdf_wi_tests += o.pdf_wi_tests; closest.add(o.closest); shadow.add(o.shadow); mis.add(o.mis);
  }
};
struct PathIntegrator {
  const Scene* scene = nullptr;
  LightDistribution* distrib = nullptr;
  int max_depth = 5;       // stored as u8 in the reference (path.rs:27)
  // NOT in the reference - a test hook for the MIS-invariance check (tests/test_invariants_cpu.py): 0 = the reference's estimate_direct (both
  // strategies, power heuristic); 1 = light sampling alone with weight 1; 2 = BSDF sampling alone with weight 1. All three are unbiased estimators
  // of the same direct lighting; they agree in the mean only if f, Li and BOTH pdfs (Light::pdf_li, Bsdf::pdf) are mutually consistent.
  int mis_mode = 0;
  float rr_threshold = 1.0f;
  B2i pixel_bounds{0, 0, 0, 0};
  RGB estimate_direct(const SurfaceInteraction& it, const Bsdf& bsdf, P2 u_scattering, const Light& light, int light_index, P2 u_light, PathStats& st) const;
  RGB uniform_sample_one_light(const SurfaceInteraction& it, const Bsdf& bsdf, ZeroTwoSequence& sampler, const Distribution1D* d, PathStats& st) const;
  RGB li(Ray ray, ZeroTwoSequence& sampler, PathStats& st) const;
};

}  // namespace orc
