// rtx_pbrt.inl — pbrt-v3 scene files for the host layer. Included by rtx_host.cpp (shares its matrix helpers).
//
// What rustracer does between `rustracer-cli scene.pbrt` and `renderer::render`: tokenise and parse the file (rc/pbrt/lexer.rs,
// rc/pbrt/parser.rs), run the directive state machine (rc/api.rs: transform stack, graphics state, named textures / materials /
// coordinate systems) and create cameras, films, shapes, materials, textures and lights from parameter lists (the `create`
// functions of each module, cited below). Here the same directives fill an rtxh_scene and an rtxh_render_params.
// Supported: Identity Translate Scale Rotate LookAt Transform ConcatTransform CoordinateSystem CoordSysTransform, Camera
// "perspective", Film "image", Sampler "02sequence"|"lowdiscrepancy", Integrator "path", PixelFilter box|triangle|gaussian|mitchell,
// Accelerator "bvh", WorldBegin/End, Attribute/TransformBegin/End, ReverseOrientation, Shape trianglemesh|plymesh, Material,
// MakeNamedMaterial, NamedMaterial, Texture (constant scale mix imagemap checkerboard uv fbm), LightSource point|distant|infinite,
// AreaLightSource diffuse|area, ObjectBegin/End, ObjectInstance (written out as world-space triangles), Include. Anything else the
// reference implements but this backend does not (other shapes, other integrators / samplers, spectral parameter types) is an
// error, never silently skipped.

namespace {

struct PbrtParam { std::string type, name; std::vector<float> nums; std::vector<std::string> strs; };
struct PbrtParams {
  std::vector<PbrtParam> v;
  const PbrtParam* find(const std::string& name, std::initializer_list<const char*> types) const {
    for (const PbrtParam& p : v)
      if (p.name == name) for (const char* t : types) if (p.type == t) return &p;
    return nullptr;
  }
  float one_float(const std::string& n, float d) const { const PbrtParam* p = find(n, {"float"}); return p && !p->nums.empty() ? p->nums[0] : d; }   // paramset.rs:19-31 find_one_*: values[0]
  int one_int(const std::string& n, int d) const { const PbrtParam* p = find(n, {"integer"}); return p && !p->nums.empty() ? (int)p->nums[0] : d; }
  bool one_bool(const std::string& n, bool d) const { const PbrtParam* p = find(n, {"bool"}); return p && !p->strs.empty() ? p->strs[0] == "true" : d; }
  std::string one_string(const std::string& n, const std::string& d) const { const PbrtParam* p = find(n, {"string"}); return p && !p->strs.empty() ? p->strs[0] : d; }
  std::string texture(const std::string& n) const { const PbrtParam* p = find(n, {"texture"}); return p && !p->strs.empty() ? p->strs[0] : std::string(); }
  bool one_rgb(const std::string& n, float out[3]) const {
    const PbrtParam* p = find(n, {"rgb", "color"});
    if (p && p->nums.size() >= 3) { out[0] = p->nums[0]; out[1] = p->nums[1]; out[2] = p->nums[2]; return true; }
    return false;
  }
  bool one_point(const std::string& n, float out[3]) const {
    const PbrtParam* p = find(n, {"point3", "point", "vector3", "vector"});
    if (p && p->nums.size() >= 3) { out[0] = p->nums[0]; out[1] = p->nums[1]; out[2] = p->nums[2]; return true; }
    return false;
  }
  const std::vector<float>* floats(const std::string& n, std::initializer_list<const char*> types) const { const PbrtParam* p = find(n, types); return p ? &p->nums : nullptr; }
};

struct PbrtGraphicsState {  // rc/api.rs:300-311
  std::map<std::string, int> float_textures, spectrum_textures;
  PbrtParams material_params; std::string material = "matte";
  std::map<std::string, int> named_materials; std::string current_named_material;
  PbrtParams area_light_params; std::string area_light;
  bool reverse_orientation = false;
};

static thread_local bool g_flatten_instances = false;  // rtxh_set_flatten_instances, per calling thread
struct PbrtObjQuadric { int kind; Xf xf; float radius, zmin, zmax, phimax; int reverse, mat, emitter /* unlisted emitter id or -1 */; };
struct PbrtSoup {  // triangle soup: world space for the scene, instance space for an ObjectBegin .. ObjectEnd block
  std::vector<float> P, N, UV, S; std::vector<int32_t> idx, tri_mat, tri_light, tri_alpha /* 2 per triangle */; std::vector<uint8_t> tri_flags;
  std::vector<PbrtObjQuadric> quadrics;  // Shape "sphere" / "disk" / "cylinder" inside an object definition (instance space)
  bool any_n = false, any_uv = false, any_s = false;
  size_t n_verts() const { return P.size() / 3; }
  // appends nv vertices (attributes may be null: zero-filled once any mesh carries them); returns the first vertex index
  size_t add_verts(size_t nv, const float* p, const float* n, const float* uv, const float* s) {
    const size_t v0 = n_verts();
    auto put = [&](std::vector<float>& a, size_t per, bool& any, const float* src) {
      if (src) { if (!any) { a.assign(v0 * per, 0.0f); any = true; } a.insert(a.end(), src, src + nv * per); }
      else if (any) a.resize(a.size() + nv * per, 0.0f);
    };
    P.insert(P.end(), p, p + 3 * nv);
    put(N, 3, any_n, n); put(UV, 2, any_uv, uv); put(S, 3, any_s, s);
    return v0;
  }
};

struct PbrtLoader {
  rtxh_scene* scene = nullptr; bool scene_handed_over = false;
  ~PbrtLoader() { if (scene && !scene_handed_over) rtxh_scene_free(scene); }
  rtxh_render_params* rp = nullptr;
  std::string film_filename, err;
  int max_prims = 4, warnings = 0;
  // options (rc/api.rs:276-298 defaults)
  std::string film_name = "image", filter_name = "box", sampler_name = "halton", accel_name = "bvh", integrator_name = "path", camera_name = "perspective";
  PbrtParams film_p, filter_p, sampler_p, accel_p, integrator_p, camera_p;
  Xf camera_to_world{mat_identity(), mat_identity()};
  bool in_world = false, world_ended = false;
  // state
  Xf ctm{mat_identity(), mat_identity()};
  std::map<std::string, Xf> named_cs;
  std::vector<Xf> pushed_transforms;
  PbrtGraphicsState gs; std::vector<PbrtGraphicsState> pushed_gs;
  PbrtSoup world;
  std::map<std::string, PbrtSoup> instances; std::string current_instance; bool in_instance = false;  // RenderOptions::instances / current_instance (api.rs:175-177)
  std::map<std::string, int> object_ids; size_t n_instances = 0;  // objects already handed to the host layer, instances placed (two-level instancing)
  int n_lights = 0, n_spheres = 0; size_t instanced_triangles = 0;

  bool fail_(const std::string& m) { if (err.empty()) err = m; return false; }
  std::string first_warning;  // what the reference logs and carries on from (warn!): counted; the first one's text is left in rtxh_last_error() after a successful load
  void warn(const std::string& m) { if (warnings == 0) first_warning = m; warnings += 1; }

  // ------------------------------------------------------------------ tokens (rc/pbrt/lexer.rs)
  struct Tok { int kind; std::string s; };  // 0 word / number, 1 quoted string, 2 '[', 3 ']'
  static bool tokenize(const std::string& src, std::vector<Tok>& out, std::string& err) {
    size_t i = 0, n = src.size();
    while (i < n) {
      char c = src[i];
      if (c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\0') { ++i; continue; }
      if (c == '#') { while (i < n && src[i] != '\n') ++i; continue; }
      if (c == '[') { out.push_back({2, "["}); ++i; continue; }
      if (c == ']') { out.push_back({3, "]"}); ++i; continue; }
      if (c == '"') {
        size_t j = i + 1;
        while (j < n && src[j] != '"') ++j;  // any character but the quote (lexer.rs:264-268)
        if (j >= n) { err = "unterminated string"; return false; }
        out.push_back({1, src.substr(i + 1, j - i - 1)}); i = j + 1; continue;
      }
      size_t j = i;
      while (j < n && src[j] != '\0' && !strchr(" \t\r\n[]\"#", src[j])) ++j;
      out.push_back({0, src.substr(i, j - i)}); i = j;
    }
    return true;
  }
  static bool is_number(const std::string& s, float& v) { char* e = nullptr; v = strtof(s.c_str(), &e); return !s.empty() && e && *e == 0; }

  // parameter list: "type name" value | [ values ]   (rc/pbrt/parser.rs:150-260)
  bool parse_params(const std::vector<Tok>& t, size_t& i, PbrtParams& out) {
    while (i < t.size() && t[i].kind == 1) {
      std::string decl = t[i].s; size_t sp = decl.find_first_of(" \t");
      if (sp == std::string::npos) return fail_("parameter declaration \"" + decl + "\" needs a type and a name");
      PbrtParam p; p.type = decl.substr(0, sp); p.name = decl.substr(decl.find_first_not_of(" \t", sp));
      ++i;
      if (i >= t.size()) return fail_("missing value of parameter " + p.name);
      auto take = [&](const Tok& k) -> bool {
        float v;
        if (k.kind == 1) { p.strs.push_back(k.s); return true; }
        if (k.kind == 0 && is_number(k.s, v)) { p.nums.push_back(v); return true; }
        if (k.kind == 0 && (k.s == "true" || k.s == "false")) { p.strs.push_back(k.s); return true; }
        return false;
      };
      if (t[i].kind == 2) {
        ++i;
        while (i < t.size() && t[i].kind != 3) { if (!take(t[i])) return fail_("bad value in parameter " + p.name); ++i; }
        if (i >= t.size()) return fail_("unterminated parameter array " + p.name);
        ++i;
      } else { if (!take(t[i])) return fail_("bad value of parameter " + p.name); ++i; }
      // ParamSet::init (paramset.rs:141-152): "spectrum" values are SPD file names, one Spectrum::from_sampled each (an unreadable file is a
      // warning and gives black, :257-266; inline samples are a TODO in the reference and an error here); "blackbody" values are
      // (temperature, scale) pairs. Both land in the same `spectra` list as rgb / color values, so they become "rgb" here.
      if (p.type == "spectrum") {
        if (p.strs.empty()) return fail_("inline samples in spectrum parameter \"" + p.name + "\" are not supported (nor by the reference, paramset.rs:144)");
        p.nums.clear();
        for (const std::string& f : p.strs) {
          float rgb[3] = {0.0f, 0.0f, 0.0f}; std::string text;
          if (!read_file(resolve(f), text)) warn("unable to read SPD file, using black");
          else {
            std::vector<float> wl, vals, all; std::istringstream lines(text); std::string line;
            while (std::getline(lines, line)) {  // read_float_file, floatfile.rs:8-35
              if (!line.empty() && line[0] == '#') continue;
              std::istringstream toks(line); std::string tok;
              while (toks >> tok) { float v; if (is_number(tok, v)) all.push_back(v); else warn("unexpected text in float file"); }
            }
            for (size_t k = 0; k + 1 < all.size(); k += 2) { wl.push_back(all[k]); vals.push_back(all[k + 1]); }
            if (all.size() % 2) warn("extra value in spectrum file");
            if (wl.empty() || rtxh_spectrum_from_sampled(wl.data(), vals.data(), (int32_t)wl.size(), rgb) != RT_OK) return fail_("bad SPD file " + f + ": " + rtxh_last_error());
          }
          p.nums.insert(p.nums.end(), rgb, rgb + 3);
        }
        p.strs.clear(); p.type = "rgb";
      } else if (p.type == "blackbody") {
        std::vector<float> out_rgb;
        for (size_t k = 0; k + 1 < p.nums.size(); k += 2) {
          float rgb[3];
          if (rtxh_spectrum_blackbody(p.nums[k], p.nums[k + 1], rgb) != RT_OK) return fail_(rtxh_last_error());
          out_rgb.insert(out_rgb.end(), rgb, rgb + 3);
        }
        p.nums = out_rgb; p.type = "rgb";
      }
      static const char* known[] = {"integer", "float", "bool", "string", "texture", "rgb", "color", "point", "point2", "point3", "vector", "vector2", "vector3", "normal", "normal3"};
      bool ok = false; for (const char* k : known) if (p.type == k) ok = true;
      if (!ok) return fail_("parameter type \"" + p.type + "\" is not supported (xyz: \"not implemented yet\" in the reference too, paramset.rs:157-160)");
      out.v.push_back(std::move(p));
    }
    return true;
  }

  // ------------------------------------------------------------------ textures (TextureParams, rc/paramset.rs:356-460)
  // constant textures and materials are values: equal ones share one id (a Shape directive creates its material anew, api.rs:313-337)
  std::map<std::array<uint32_t, 3>, int> const_cache; std::map<std::vector<int32_t>, int> material_cache;
  int const_tex(const float v[3]) {
    std::array<uint32_t, 3> key; memcpy(key.data(), v, 12);
    auto it = const_cache.find(key); if (it != const_cache.end()) return it->second;
    const float map[4] = {1, 1, 0, 0};
    const int id = rtxh_scene_add_texture(scene, RT_TEX_CONST, v, -1, -1, -1, -1, map);
    const_cache[key] = id; return id;
  }
  int spectrum_texture(const PbrtParams& gp, const PbrtParams& mp, const std::string& n, float d0, float d1, float d2) {
    std::string name = gp.texture(n); if (name.empty()) name = mp.texture(n);
    if (!name.empty()) { auto it = gs.spectrum_textures.find(name); if (it != gs.spectrum_textures.end()) return it->second; warn("spectrum texture not found"); }
    float v[3] = {d0, d1, d2};
    mp.one_rgb(n, v); gp.one_rgb(n, v);
    return const_tex(v);
  }
  int float_texture(const PbrtParams& gp, const PbrtParams& mp, const std::string& n, float d) {
    std::string name = gp.texture(n); if (name.empty()) name = mp.texture(n);
    if (!name.empty()) { auto it = gs.float_textures.find(name); if (it != gs.float_textures.end()) return it->second; warn("float texture not found"); }
    float x = gp.one_float(n, mp.one_float(n, d));
    float v[3] = {x, x, x};
    return const_tex(v);
  }
  int float_texture_or_none(const PbrtParams& gp, const PbrtParams& mp, const std::string& n) {
    std::string name = gp.texture(n); if (name.empty()) name = mp.texture(n);
    if (!name.empty()) { auto it = gs.float_textures.find(name); if (it != gs.float_textures.end()) return it->second; warn("float texture not found"); return -1; }
    const PbrtParam* f = gp.find(n, {"float"}); if (!f) f = mp.find(n, {"float"});   // paramset.rs:457-465
    if (!f || f->nums.empty()) return -1;
    const float v[3] = {f->nums[0], f->nums[0], f->nums[0]};
    return const_tex(v);
  }
  static bool tp_bool(const PbrtParams& gp, const PbrtParams& mp, const std::string& n, bool d) { return gp.one_bool(n, mp.one_bool(n, d)); }
  static float tp_float(const PbrtParams& gp, const PbrtParams& mp, const std::string& n, float d) { return gp.one_float(n, mp.one_float(n, d)); }
  static std::string tp_string(const PbrtParams& gp, const PbrtParams& mp, const std::string& n, const std::string& d) { return gp.one_string(n, mp.one_string(n, d)); }

  std::string base_dir;
  std::string resolve(const std::string& f) const { return (!f.empty() && f[0] != '/' && !base_dir.empty()) ? base_dir + "/" + f : f; }

  // image file -> MIP pyramid id (read_image, imageio.rs:16-33: png tga hdr pfm by extension). Like the reference, a failed read gives
  // a 1x1 texel (grey for textures, imagemap.rs:62-69; the radiance itself for environment maps, infinite.rs:62-69) - with a warning.
  int load_mip(const std::string& file, bool flip_y, float scale, bool gamma, const float mul[3], bool to_float, int trilinear, float max_aniso, int wrap, bool grey_on_failure) {
    int32_t w = 0, h = 0; float* rgb = nullptr;
    std::vector<float> px;
    if (!file.empty() && rtxh_image_read(resolve(file).c_str(), &w, &h, &rgb) == RT_OK) { px.assign(rgb, rgb + (size_t)w * h * 3); rtxh_free(rgb); }
    else { warn("image not readable: " + file); w = h = 1; if (grey_on_failure) px = {0.18f, 0.18f, 0.18f}; else px = {1.0f, 1.0f, 1.0f}; g_err.clear(); }
    if (flip_y)
      for (int y = 0; y < h / 2; ++y) for (int x = 0; x < w * 3; ++x) std::swap(px[(size_t)y * w * 3 + x], px[(size_t)(h - 1 - y) * w * 3 + x]);
    for (size_t i = 0; i < px.size(); i += 3) {
      float c[3] = {px[i], px[i + 1], px[i + 2]};
      if (gamma) for (int k = 0; k < 3; ++k) c[k] = c[k] <= 0.04045f ? c[k] / 12.92f : std::pow((c[k] + 0.055f) * 1.0f / 1.055f, 2.4f);  // spectrum.rs:379-385
      for (int k = 0; k < 3; ++k) c[k] = scale * c[k] * (mul ? mul[k] : 1.0f);
      if (to_float) { float y = 0.212671f * c[0] + 0.715160f * c[1] + 0.072169f * c[2]; c[0] = c[1] = c[2] = y; }  // convert_to_float = y(), imagemap.rs:214-216
      px[i] = c[0]; px[i + 1] = c[1]; px[i + 2] = c[2];
    }
    return rtxh_scene_add_mipmap(scene, w, h, px.data(), trilinear, max_aniso, wrap);
  }

  bool make_texture(const std::string& name, const std::string& type, const std::string& cls, const PbrtParams& p) {  // api.rs:783-843, 1201-1259
    const bool is_float = type == "float";
    if (!is_float && type != "color" && type != "spectrum") { warn("texture type unknown"); return true; }
    const PbrtParams none;
    const float map[4] = {p.one_float("uscale", 1.0f), p.one_float("vscale", 1.0f), p.one_float("udelta", 0.0f), p.one_float("vdelta", 0.0f)};
    const float zero3[3] = {0, 0, 0};
    auto tex = [&](const std::string& n, float d) { return is_float ? float_texture(p, none, n, d) : spectrum_texture(p, none, n, d, d, d); };
    auto uv_mapping_only = [&]() { return p.one_string("mapping", "uv") == "uv"; };
    int id = -1;
    if (cls == "constant") {
      float v[3] = {1, 1, 1};
      if (is_float) { float x = p.one_float("value", 1.0f); v[0] = v[1] = v[2] = x; } else p.one_rgb("value", v);
      id = const_tex(v);
    } else if (cls == "scale") id = rtxh_scene_add_texture(scene, RT_TEX_SCALE, zero3, tex("tex1", 1.0f), tex("tex2", 1.0f), -1, -1, map);
    else if (cls == "mix") {
      int t1 = tex("tex1", 0.0f), t2 = tex("tex2", 1.0f), am = float_texture(p, none, "amount", 0.5f);
      id = rtxh_scene_add_texture(scene, RT_TEX_MIX, zero3, t1, t2, am, -1, map);
    } else if (cls == "imagemap") {  // imagemap.rs:100-139
      if (!uv_mapping_only()) return fail_("imagemap: only \"uv\" mapping is implemented (as in the reference)");
      const std::string wrap = p.one_string("wrap", "repeat");
      const std::string file = p.one_string("filename", "");
      const bool is_ldr = file.size() > 4 && (file.substr(file.size() - 4) == ".tga" || file.substr(file.size() - 4) == ".png");
      int mip = load_mip(file, true, p.one_float("scale", 1.0f), p.one_bool("gamma", is_ldr), nullptr, is_float, p.one_bool("trilinear", false) ? 1 : 0,
                         p.one_float("maxanisotropy", 8.0f), wrap == "black" ? RT_WRAP_BLACK : (wrap == "clamp" ? RT_WRAP_CLAMP : RT_WRAP_REPEAT), true);
      if (mip < 0) return fail_(rtxh_last_error());
      id = rtxh_scene_add_texture(scene, RT_TEX_IMAGE, zero3, -1, -1, -1, mip, map);
    } else if (cls == "checkerboard" && !is_float) {  // checkerboard.rs:44-95
      if (p.one_int("dimension", 2) != 2 || !uv_mapping_only()) return fail_("checkerboard: only dimension 2 with \"uv\" mapping is supported");
      const std::string aa = p.one_string("aamode", "closedform");
      id = rtxh_scene_add_texture(scene, RT_TEX_CHECKER, zero3, spectrum_texture(p, none, "tex1", 1, 1, 1), spectrum_texture(p, none, "tex2", 0, 0, 0), aa == "none" ? 0 : 1, -1, map);
    } else if (cls == "uv" && !is_float) {
      if (!uv_mapping_only()) return fail_("uv texture: only \"uv\" mapping is implemented (as in the reference)");
      id = rtxh_scene_add_texture(scene, RT_TEX_UV, zero3, -1, -1, -1, -1, map);
    } else if (cls == "fbm") {  // fbm.rs:25-44; the reference stores the CTM as world-to-texture (texture/mod.rs:96-100): identity only here
      for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) if (ctm.m.a[r][c] != (r == c ? 1.0f : 0.0f)) return fail_("fbm texture under a non-identity transform is not supported");
      const float v[3] = {p.one_float("omega", 0.5f), 0, 0};
      id = rtxh_scene_add_texture(scene, RT_TEX_FBM, v, -1, -1, p.one_int("octaves", 8), -1, map);
    } else return fail_("texture class \"" + cls + "\" is not supported");
    if (id < 0) return fail_(rtxh_last_error());
    (is_float ? gs.float_textures : gs.spectrum_textures)[name] = id;
    return true;
  }

  // ------------------------------------------------------------------ materials (make_material, api.rs:1141-1183 + each create())
  int make_material(const std::string& name, const PbrtParams& gp, const PbrtParams& mp) {
    int slots[RT_N_SLOTS]; for (int& s : slots) s = -1;
    int kind = RT_MAT_MATTE; int remap = tp_bool(gp, mp, "remaproughness", true) ? 1 : 0;
    auto S = [&](int slot, const char* n, float d0, float d1, float d2) { slots[slot] = spectrum_texture(gp, mp, n, d0, d1, d2); };
    auto F = [&](int slot, const char* n, float d) { slots[slot] = float_texture(gp, mp, n, d); };
    auto eta_or_index = [&]() { const int e = float_texture_or_none(gp, mp, "eta"); slots[RT_SLOT_ETA] = e >= 0 ? e : float_texture(gp, mp, "index", 1.5f); };  // glass.rs:32-34, uber.rs:39-41
    if (name == "plastic") { kind = RT_MAT_PLASTIC; S(RT_SLOT_KD, "Kd", .25f, .25f, .25f); S(RT_SLOT_KS, "Ks", .25f, .25f, .25f); F(RT_SLOT_ROUGHNESS, "roughness", 0.1f); }
    else if (name == "metal") {  // metal.rs:23-47; default eta / k = Spectrum::from_sampled over the measured copper tables (:25-29)
      float ce[3], ck[3]; rtxh_copper(ce, ck);
      kind = RT_MAT_METAL; S(RT_SLOT_ETA, "eta", ce[0], ce[1], ce[2]); S(RT_SLOT_K, "k", ck[0], ck[1], ck[2]); F(RT_SLOT_ROUGHNESS, "roughness", 0.01f);
      slots[RT_SLOT_UROUGH] = float_texture_or_none(gp, mp, "uroughness"); slots[RT_SLOT_VROUGH] = float_texture_or_none(gp, mp, "vroughness");
    } else if (name == "mirror") { kind = RT_MAT_MIRROR; S(RT_SLOT_KR, "Kr", .9f, .9f, .9f); }
    else if (name == "glass") { kind = RT_MAT_GLASS; S(RT_SLOT_KR, "Kr", 1, 1, 1); S(RT_SLOT_KT, "Kt", 1, 1, 1); eta_or_index(); F(RT_SLOT_UROUGH, "uroughness", 0.0f); F(RT_SLOT_VROUGH, "vroughness", 0.0f); }
    else if (name == "uber") {
      kind = RT_MAT_UBER; S(RT_SLOT_KD, "Kd", .25f, .25f, .25f); S(RT_SLOT_KS, "Ks", .25f, .25f, .25f); S(RT_SLOT_KR, "Kr", 0, 0, 0); S(RT_SLOT_KT, "Kt", 0, 0, 0);
      F(RT_SLOT_ROUGHNESS, "roughness", 0.1f); eta_or_index(); S(RT_SLOT_OPACITY, "opacity", 1, 1, 1);
      slots[RT_SLOT_UROUGH] = float_texture_or_none(gp, mp, "uroughness"); slots[RT_SLOT_VROUGH] = float_texture_or_none(gp, mp, "vroughness");
    } else if (name == "substrate") { kind = RT_MAT_SUBSTRATE; S(RT_SLOT_KD, "Kd", .5f, .5f, .5f); S(RT_SLOT_KS, "Ks", .5f, .5f, .5f); F(RT_SLOT_UROUGH, "uroughness", 0.1f); F(RT_SLOT_VROUGH, "vroughness", 0.1f); }
    else if (name == "translucent") {
      kind = RT_MAT_TRANSLUCENT; S(RT_SLOT_KD, "Kd", .25f, .25f, .25f); S(RT_SLOT_KS, "Ks", .25f, .25f, .25f); S(RT_SLOT_REFLECT, "reflect", .5f, .5f, .5f); S(RT_SLOT_TRANSMIT, "transmit", .5f, .5f, .5f);
      F(RT_SLOT_ROUGHNESS, "roughness", 0.1f);
    } else if (name == "disney") {  // disney.rs:43-80
      kind = RT_MAT_DISNEY; S(RT_SLOT_KD, "color", .5f, .5f, .5f); F(RT_SLOT_KS, "metallic", 0.0f); F(RT_SLOT_ETA, "eta", 1.5f); F(RT_SLOT_ROUGHNESS, "roughness", 0.5f);
      F(RT_SLOT_KR, "speculartint", 0.0f); F(RT_SLOT_UROUGH, "anisotropic", 0.0f); F(RT_SLOT_KT, "sheen", 0.0f); F(RT_SLOT_SIGMA, "sheentint", 0.5f); F(RT_SLOT_VROUGH, "clearcoat", 0.0f);
      F(RT_SLOT_K, "clearcoatgloss", 1.0f); F(RT_SLOT_OPACITY, "spectrans", 0.0f); S(RT_SLOT_REFLECT, "scatterdistance", 0, 0, 0); F(RT_SLOT_TRANSMIT, "flatness", 0.0f); F(RT_SLOT_AMOUNT, "difftrans", 1.0f);
      slots[RT_SLOT_M1] = tp_bool(gp, mp, "thin", false) ? 1 : 0;
    } else if (name == "mix") {
      kind = RT_MAT_MIX; S(RT_SLOT_AMOUNT, "amount", .5f, .5f, .5f);
      for (int k = 0; k < 2; ++k) {
        const std::string nm = tp_string(gp, mp, k == 0 ? "namedmaterial1" : "namedmaterial2", "");
        auto it = gs.named_materials.find(nm);
        if (it != gs.named_materials.end()) slots[RT_SLOT_M1 + k] = it->second;
        else { warn("named material undefined, using matte"); slots[RT_SLOT_M1 + k] = make_material("matte", gp, mp); }
      }
    } else {
      if (name == "fourier") { fail_("material \"fourier\" is not supported"); return -1; }
      if (name != "matte") warn("unknown material, using matte");
      kind = RT_MAT_MATTE; S(RT_SLOT_KD, "Kd", .5f, .5f, .5f); F(RT_SLOT_SIGMA, "sigma", 0.0f);
    }
    const int bump = kind == RT_MAT_MIX ? -1 : float_texture_or_none(gp, mp, "bumpmap");
    std::vector<int32_t> key(slots, slots + RT_N_SLOTS); key.push_back(kind); key.push_back(remap); key.push_back(bump);
    auto hit = material_cache.find(key); if (hit != material_cache.end()) return hit->second;
    const int id = rtxh_scene_add_material(scene, kind, slots, remap, bump);
    material_cache[key] = id; return id;
  }
  int current_material(const PbrtParams& shape_params) {  // GraphicsState::create_material, api.rs:313-337
    if (!gs.current_named_material.empty()) {
      auto it = gs.named_materials.find(gs.current_named_material);
      if (it != gs.named_materials.end()) return it->second;
      warn("no such named material, using matte");
      return make_material("matte", shape_params, gs.material_params);
    }
    return make_material(gs.material, shape_params, gs.material_params);
  }

  // ------------------------------------------------------------------ shapes (make_shapes api.rs:1093-1139; mesh.rs:76-179; plymesh.rs)
  bool add_shape(const std::string& name, const PbrtParams& p) {
    std::vector<float> vp, vn, vuv, vs; std::vector<int32_t> vi;
    if (name == "sphere" || name == "disk" || name == "cylinder") {  // an analytic primitive under the CTM, with one DiffuseAreaLight if an area light is active (api.rs:933-946)
      const int kind = name == "sphere" ? 0 : (name == "disk" ? 1 : 2);
      const float radius = p.one_float("radius", 1.0f);
      // Sphere::create (sphere.rs:53-68): zmin zmax phimax; Disk::create (disk.rs:48-62): height innerradius phimax; Cylinder::create
      // (cylinder.rs:26-46) reads "z_min" "z_max" "phi_max" - not pbrt's names: a reference quirk, kept
      const float zmin = kind == 0 ? p.one_float("zmin", -radius) : (kind == 1 ? p.one_float("height", 0.0f) : p.one_float("z_min", -1.0f));
      const float zmax = kind == 0 ? p.one_float("zmax", radius) : (kind == 1 ? p.one_float("innerradius", 0.0f) : p.one_float("z_max", 1.0f));
      const float phimax = kind == 2 ? p.one_float("phi_max", 360.0f) : p.one_float("phimax", 360.0f);
      const int mat = current_material(p);
      if (mat < 0) return fail_(err.empty() ? rtxh_last_error() : err);
      int light = -1;
      if (!gs.area_light.empty()) {
        if (gs.area_light != "area" && gs.area_light != "diffuse") return fail_("area light \"" + gs.area_light + "\" unknown");
        light = n_lights;
      }
      if (in_instance) {
        // Inside an object definition: kept in instance space and placed by every ObjectInstance (instantiate). The reference wraps the object's aggregate in a
        // TransformedPrimitive (api.rs:1053-1090) whatever it holds; a quadric carries its own object-to-world transform, so an instance of it is the same
        // quadric under instance_to_world * object_to_world - written out, like a masked mesh, with rounding as the only difference. Its area light stays
        // with the primitive and out of the light list (api.rs:954-964): an unlisted emitter.
        int emitter = -1;
        if (light >= 0) {
          float L[3] = {1, 1, 1}, sc[3] = {1, 1, 1};
          gs.area_light_params.one_rgb("L", L); gs.area_light_params.one_rgb("scale", sc);
          for (int c = 0; c < 3; ++c) L[c] *= sc[c];
          emitter = rtxh_scene_add_emitter(scene, L, gs.area_light_params.one_bool("twosided", false) ? 1 : 0);
          if (emitter < 0) return fail_(rtxh_last_error());
        }
        instances[current_instance].quadrics.push_back(PbrtObjQuadric{kind, ctm, radius, zmin, zmax, phimax, gs.reverse_orientation ? 1 : 0, mat, emitter});
        return true;
      }
      const int k = rtxh_scene_add_quadric(scene, kind, &ctm.m.a[0][0], &ctm.inv.a[0][0], radius, zmin, zmax, phimax, gs.reverse_orientation ? 1 : 0, mat, light);
      if (k < 0) return fail_(rtxh_last_error());
      if (light >= 0) {
        float L[3] = {1, 1, 1}, sc[3] = {1, 1, 1};
        gs.area_light_params.one_rgb("L", L); gs.area_light_params.one_rgb("scale", sc);
        for (int c = 0; c < 3; ++c) L[c] *= sc[c];
        if (rtxh_scene_add_light(scene, RT_LIGHT_DIFFUSE_AREA, -2 - k, L, gs.area_light_params.one_bool("twosided", false) ? 1 : 0, nullptr, -1, nullptr, nullptr) < 0) return fail_(rtxh_last_error());
        n_lights++;
      }
      n_spheres++;
      return true;
    }
    if (name == "trianglemesh") {
      const std::vector<float>* ind = p.floats("indices", {"integer"});
      const std::vector<float>* pts = p.floats("P", {"point3", "point"});
      if (!ind || ind->empty() || !pts || pts->empty()) { warn("trianglemesh without indices / P: no shape"); return true; }  // mesh.rs:107-114
      vp = *pts; for (float f : *ind) vi.push_back((int32_t)f);
      const std::vector<float>* uv = p.floats("uv", {"point2", "float"}); if (!uv) uv = p.floats("st", {"point2", "float"});
      if (uv) vuv = *uv;
      const std::vector<float>* s = p.floats("S", {"vector3", "vector"}); if (s && s->size() == vp.size()) vs = *s;
      const std::vector<float>* n = p.floats("N", {"normal3", "normal"}); if (n && n->size() == vp.size()) vn = *n;
      if (!vuv.empty() && vuv.size() / 2 < vp.size() / 3) vuv.clear();
    } else if (name == "plymesh") {
      rtxh_ply ply;
      if (rtxh_ply_read(resolve(p.one_string("filename", "")).c_str(), &ply) != RT_OK) return fail_(rtxh_last_error());
      vp.assign(ply.P, ply.P + (size_t)ply.n_verts * 3); vi.assign(ply.idx, ply.idx + (size_t)ply.n_tris * 3);
      if (ply.N) vn.assign(ply.N, ply.N + (size_t)ply.n_verts * 3);
      if (ply.UV) vuv.assign(ply.UV, ply.UV + (size_t)ply.n_verts * 2);
      rtxh_ply_free(&ply);
    } else return fail_("shape \"" + name + "\" is not supported (triangle meshes, spheres, disks and cylinders; cone / paraboloid / hyperboloid / curve are unimplemented!() in the reference too)");
    // "alpha" / "shadowalpha" (TriangleMesh::create mesh.rs:134-156, plymesh.rs:143-165): a named float texture (unknown name: logged, no mask),
    // else the constant-0 texture when the float parameter is exactly 0
    auto mask = [&](const char* n) -> int {
      const std::string tn = p.texture(n);
      if (!tn.empty()) { auto it = gs.float_textures.find(tn); if (it != gs.float_textures.end()) return it->second; warn("alpha texture not found"); return -1; }
      if (p.one_float(n, 1.0f) == 0.0f) { const float z[3] = {0.0f, 0.0f, 0.0f}; return const_tex(z); }
      return -1;
    };
    const int alpha_tex = mask("alpha"), shadow_alpha_tex = mask("shadowalpha");
    const size_t nv = vp.size() / 3, nt = vi.size() / 3;
    for (int32_t i : vi) if (i < 0 || (size_t)i >= nv) return fail_("triangle index out of range");
    if (nt == 0) return true;
    const int mat = current_material(p);
    if (mat < 0) return fail_(err.empty() ? rtxh_last_error() : err);
    // TriangleMesh::new (mesh.rs:47-74): points go to world space (instance space inside an object definition); normals and tangents are kept as given
    PbrtSoup& soup = in_instance ? instances[current_instance] : world;
    std::vector<float> wp(vp.size());
    for (size_t v = 0; v < nv; ++v) xf_point(ctm.m, &vp[3 * v], &wp[3 * v]);
    const size_t v0 = soup.add_verts(nv, wp.data(), vn.empty() ? nullptr : vn.data(), vuv.empty() ? nullptr : vuv.data(), vs.empty() ? nullptr : vs.data());
    const uint8_t flags = (uint8_t)(((gs.reverse_orientation != swaps_handedness(ctm.m)) ? RT_TRI_FLIP : 0) | (!vn.empty() ? RT_TRI_HAS_N : 0) | (!vuv.empty() ? RT_TRI_HAS_UV : 0) | (!vs.empty() ? RT_TRI_HAS_S : 0));
    // area light per triangle (make_area_light, api.rs:1185-1199; DiffuseAreaLight::create, light/diffuse.rs:39-51)
    float L[3] = {1, 1, 1}, sc[3] = {1, 1, 1}; bool emit = false, two_sided = false;
    if (!gs.area_light.empty()) {
      if (gs.area_light != "area" && gs.area_light != "diffuse") return fail_("area light \"" + gs.area_light + "\" unknown");
      gs.area_light_params.one_rgb("L", L); gs.area_light_params.one_rgb("scale", sc); two_sided = gs.area_light_params.one_bool("twosided", false); emit = true;
      for (int k = 0; k < 3; ++k) L[k] *= sc[k];
    }
    // Inside an object definition the shapes keep their area light - they glow when a camera ray or a specular bounce reaches them - but the light never
    // enters the scene's list (api.rs:954-964: `area_lights` is dropped when there is a current instance): an emitter that nothing samples
    int32_t unlisted = -1;
    if (emit && in_instance) {
      unlisted = rtxh_scene_add_emitter(scene, L, two_sided ? 1 : 0);
      if (unlisted < 0) return fail_(rtxh_last_error());
    }
    for (size_t t = 0; t < nt; ++t) {
      const int32_t tri_index = (int32_t)(soup.idx.size() / 3);
      for (int k = 0; k < 3; ++k) soup.idx.push_back((int32_t)(v0 + vi[3 * t + k]));
      soup.tri_mat.push_back(mat); soup.tri_flags.push_back(flags); soup.tri_alpha.push_back(alpha_tex); soup.tri_alpha.push_back(shadow_alpha_tex);
      if (unlisted >= 0) soup.tri_light.push_back(-2 - unlisted);
      else if (emit) {
        if (rtxh_scene_add_light(scene, RT_LIGHT_DIFFUSE_AREA, tri_index, L, two_sided ? 1 : 0, nullptr, -1, nullptr, nullptr) < 0) return fail_(rtxh_last_error());
        soup.tri_light.push_back(n_lights++);
      } else soup.tri_light.push_back(-1);
    }
    return true;
  }
  static bool swaps_handedness(const Mat& m) {  // transform.rs:255-261
    const float det = m.a[0][0] * (m.a[1][1] * m.a[2][2] - m.a[1][2] * m.a[2][1]) - m.a[0][1] * (m.a[1][0] * m.a[2][2] - m.a[1][2] * m.a[2][0]) + m.a[0][2] * (m.a[1][0] * m.a[2][1] - m.a[1][1] * m.a[2][0]);
    return det < 0.0f;
  }

  // ObjectInstance (api.rs:1053-1090 + TransformedPrimitive, primitive.rs:79-118). The reference keeps one BVH per object and sends each ray
  // through instance_to_world^-1; here the instance is written out: with 288 GB of HBM the copies are cheap and every ray stays in the
  // single-level traversal kernels. Points go through instance_to_world, normals through its inverse transpose and tangents through its
  // linear part (SurfaceInteraction::transform, interaction.rs:156-190 - all linear, so transforming the vertex attributes commutes with
  // the barycentric interpolation); a mirroring transform flips the winding-derived normal, hence the toggled RT_TRI_FLIP.
  bool instantiate(const std::string& name) {
    auto it = instances.find(name);
    if (it == instances.end()) return fail_("Unable to find instance named " + name);
    const PbrtSoup& o = it->second;
    if (o.idx.empty() && o.quadrics.empty()) return true;
    // The reference's form (the default): one tree per object, a TransformedPrimitive per instance (primitive.rs:79-118) over WHATEVER the definition collected -
    // triangles, masked or not, and quadrics (round 6: rtxh_object_add_quadric / rtxh_object_set_alpha; a quadric keeps its own object_to_world = the CTM inside
    // the definition, so a ray meets two Transform * Ray roundings, as in the reference) - rtxh_scene_add_object once, rtxh_scene_add_instance per use, nothing
    // copied. Everything is written out instead when the caller asked for it (rtxh_set_flatten_instances: every ray then stays in the single-level kernels; a quadric
    // then goes under the PRODUCT instance_to_world * object_to_world - one rounding, not the reference's two).
    if (!g_flatten_instances) {
      auto id = object_ids.find(name);
      if (id == object_ids.end()) {
        const int k = rtxh_scene_add_object(scene, o.P.data(), (int32_t)o.n_verts(), o.idx.data(), (int32_t)(o.idx.size() / 3), o.any_n ? o.N.data() : nullptr,
                                            o.any_uv ? o.UV.data() : nullptr, o.any_s ? o.S.data() : nullptr, o.tri_mat.data(), o.tri_flags.data());
        if (k < 0) return fail_(rtxh_last_error());
        for (const PbrtObjQuadric& q : o.quadrics)
          if (rtxh_object_add_quadric(scene, k, q.kind, &q.xf.m.a[0][0], &q.xf.inv.a[0][0], q.radius, q.zmin, q.zmax, q.phimax, q.reverse, q.mat, q.emitter) < 0) return fail_(rtxh_last_error());
        bool masked = false;
        for (int32_t a : o.tri_alpha) masked = masked || a >= 0;
        if (masked && rtxh_object_set_alpha(scene, k, o.tri_alpha.data()) != RT_OK) return fail_(rtxh_last_error());
        bool emits = false;
        for (int32_t l : o.tri_light) emits = emits || l <= -2;
        if (emits) {
          std::vector<int32_t> em(o.tri_light.size());
          for (size_t t = 0; t < em.size(); ++t) em[t] = o.tri_light[t] <= -2 ? -2 - o.tri_light[t] : -1;
          if (rtxh_scene_object_emitters(scene, k, em.data()) != RT_OK) return fail_(rtxh_last_error());
        }
        id = object_ids.emplace(name, k).first;
      }
      if (rtxh_scene_add_instance(scene, id->second, &ctm.m.a[0][0], &ctm.inv.a[0][0]) < 0) return fail_(rtxh_last_error());
      n_instances += 1;
      return true;
    }
    for (const PbrtObjQuadric& q : o.quadrics) {  // written out: the object's quadrics under instance_to_world * object_to_world
      const Xf w = xf_mul(ctm, q.xf);
      if (rtxh_scene_add_quadric(scene, q.kind, &w.m.a[0][0], &w.inv.a[0][0], q.radius, q.zmin, q.zmax, q.phimax, q.reverse, q.mat, q.emitter >= 0 ? -2 - q.emitter : -1) < 0) return fail_(rtxh_last_error());
      n_spheres++;
    }
    if (o.idx.empty()) return true;
    // Written-out instances cost memory in proportion to instances x mesh size, where the reference's TransformedPrimitive shares one BVH. A scene
    // that instantiates its way past the budget is refused with a message instead of exhausting the host (RTX_INSTANCE_TRIANGLE_BUDGET, default 2^28
    // triangles ~ 40 GB of device geometry and BVH - a fraction of the 288 GB the design counts on).
    static const size_t budget = getenv("RTX_INSTANCE_TRIANGLE_BUDGET") ? (size_t)strtoull(getenv("RTX_INSTANCE_TRIANGLE_BUDGET"), nullptr, 10) : ((size_t)1 << 28);
    instanced_triangles += o.idx.size() / 3;
    if (world.idx.size() / 3 + o.idx.size() / 3 > budget)
      return fail_("ObjectInstance \"" + name + "\": writing the instances out would exceed " + std::to_string(budget) + " triangles (" + std::to_string(instanced_triangles) +
                   " instanced so far; RTX_INSTANCE_TRIANGLE_BUDGET raises the limit, two-level instancing - the default - copies nothing)");
    const size_t nv = o.n_verts();
    std::vector<float> p(3 * nv), n, sv;
    for (size_t v = 0; v < nv; ++v) xf_point(ctm.m, &o.P[3 * v], &p[3 * v]);
    if (o.any_n) {
      n.resize(3 * nv); const Mat& i = ctm.inv;  // Transform::transform_normal: the transpose of the inverse (transform.rs:244-253)
      for (size_t v = 0; v < nv; ++v) { const float* x = &o.N[3 * v];
        for (int r = 0; r < 3; ++r) n[3 * v + r] = i.a[0][r] * x[0] + i.a[1][r] * x[1] + i.a[2][r] * x[2]; }
    }
    if (o.any_s) {
      sv.resize(3 * nv); const Mat& m = ctm.m;
      for (size_t v = 0; v < nv; ++v) { const float* x = &o.S[3 * v];
        for (int r = 0; r < 3; ++r) sv[3 * v + r] = m.a[r][0] * x[0] + m.a[r][1] * x[1] + m.a[r][2] * x[2]; }
    }
    const size_t v0 = world.add_verts(nv, p.data(), o.any_n ? n.data() : nullptr, o.any_uv ? o.UV.data() : nullptr, o.any_s ? sv.data() : nullptr);
    const uint8_t toggle = swaps_handedness(ctm.m) ? RT_TRI_FLIP : 0;
    for (size_t t = 0; t < o.idx.size() / 3; ++t) {
      for (int k = 0; k < 3; ++k) world.idx.push_back((int32_t)(v0 + o.idx[3 * t + k]));
      world.tri_mat.push_back(o.tri_mat[t]); world.tri_light.push_back(o.tri_light[t]); world.tri_flags.push_back((uint8_t)(o.tri_flags[t] ^ toggle));  // (an emitter of the object stays unlisted: -2 - k)
      world.tri_alpha.push_back(o.tri_alpha[2 * t]); world.tri_alpha.push_back(o.tri_alpha[2 * t + 1]);
    }
    return true;
  }

  // ------------------------------------------------------------------ lights (make_light, api.rs:494-514)
  bool add_light(const std::string& name, const PbrtParams& p) {
    float c[3] = {1, 1, 1}, sc[3] = {1, 1, 1};
    p.one_rgb("scale", sc);
    if (name == "point") {  // point.rs:28-35: pos = (translate(from) * l2w)(0,0,0)
      p.one_rgb("I", c); float from[3] = {0, 0, 0}; p.one_point("from", from);
      Xf t = xf_mul(xf_translate(from[0], from[1], from[2]), ctm);
      const float z[3] = {0, 0, 0}; float pos[3]; xf_point(t.m, z, pos);
      for (int k = 0; k < 3; ++k) c[k] *= sc[k];
      if (rtxh_scene_add_light(scene, RT_LIGHT_POINT, -1, c, 0, pos, -1, nullptr, nullptr) < 0) return fail_(rtxh_last_error());
    } else if (name == "distant") {  // distant.rs:34-41: w_light = l2w * (from - to)
      p.one_rgb("L", c); float from[3] = {0, 0, 0}, to[3] = {0, 0, 1}; p.one_point("from", from); p.one_point("to", to);
      const float d[3] = {from[0] - to[0], from[1] - to[1], from[2] - to[2]};
      const Mat& m = ctm.m;
      float w[3] = {m.a[0][0] * d[0] + m.a[0][1] * d[1] + m.a[0][2] * d[2], m.a[1][0] * d[0] + m.a[1][1] * d[1] + m.a[1][2] * d[2], m.a[2][0] * d[0] + m.a[2][1] * d[1] + m.a[2][2] * d[2]};
      for (int k = 0; k < 3; ++k) c[k] *= sc[k];
      if (rtxh_scene_add_light(scene, RT_LIGHT_DISTANT, -1, c, 0, w, -1, nullptr, nullptr) < 0) return fail_(rtxh_last_error());
    } else if (name == "infinite") {  // infinite.rs:47-127: texels * (L * scale), MIPMap(trilinear = false, max_aniso = 0, Repeat); no y flip
      p.one_rgb("L", c); for (int k = 0; k < 3; ++k) c[k] *= sc[k];
      const int mip = load_mip(p.one_string("mapname", ""), false, 1.0f, false, c, false, 0, 0.0f, RT_WRAP_REPEAT, false);
      if (mip < 0) return fail_(rtxh_last_error());
      const float one[3] = {1, 1, 1};  // the radiance scale is already in the texels
      if (rtxh_scene_add_light(scene, RT_LIGHT_INFINITE, -1, one, 0, nullptr, mip, &ctm.m.a[0][0], &ctm.inv.a[0][0]) < 0) return fail_(rtxh_last_error());
    } else return fail_("light \"" + name + "\" is not supported");
    n_lights += 1;
    return true;
  }

  // ------------------------------------------------------------------ options -> rtxh_render_params
  bool finish_options() {
    rtxh_render_params& o = *rp; memset(&o, 0, sizeof o);
    if (film_name != "image") return fail_("Film \"" + film_name + "\" unknown.");
    o.xres = film_p.one_int("xresolution", 1280); o.yres = film_p.one_int("yresolution", 720);  // film.rs:118-145
    o.crop[0] = 0; o.crop[1] = 1; o.crop[2] = 0; o.crop[3] = 1;
    if (const std::vector<float>* cr = film_p.floats("cropwindow", {"float"})) {
      if (cr->size() == 4) {
        auto cl = [](float v) { return v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v); };
        o.crop[0] = cl(std::fmin((*cr)[0], (*cr)[1])); o.crop[1] = cl(std::fmax((*cr)[0], (*cr)[1])); o.crop[2] = cl(std::fmin((*cr)[2], (*cr)[3])); o.crop[3] = cl(std::fmax((*cr)[2], (*cr)[3]));
      } else warn("cropwindow expects 4 values");
    }
    o.film_scale = film_p.one_float("scale", 1.0f); o.max_sample_luminance = film_p.one_float("maxsampleluminance", std::numeric_limits<float>::infinity());
    film_filename = film_p.one_string("filename", ""); film_filename = film_filename.empty() ? "image.png" : "rt-" + film_filename;  // film.rs:118-123
    if (filter_name == "box") { o.filter_kind = 0; o.filter_params[0] = filter_p.one_float("xwidth", 0.5f); o.filter_params[1] = filter_p.one_float("ywidth", 0.5f); }
    else if (filter_name == "triangle") { o.filter_kind = 1; o.filter_params[0] = filter_p.one_float("xwidth", 2.0f); o.filter_params[1] = filter_p.one_float("ywidth", 2.0f); }
    else if (filter_name == "gaussian") { o.filter_kind = 2; o.filter_params[0] = filter_p.one_float("xwidth", 2.0f); o.filter_params[1] = filter_p.one_float("ywidth", 2.0f); o.filter_params[2] = filter_p.one_float("alpha", 2.0f); }
    else if (filter_name == "mitchell") { o.filter_kind = 3; o.filter_params[0] = filter_p.one_float("xwidth", 2.0f); o.filter_params[1] = filter_p.one_float("ywidth", 2.0f); o.filter_params[2] = filter_p.one_float("B", 1.0f / 3.0f); o.filter_params[3] = filter_p.one_float("C", 1.0f / 3.0f); }
    else return fail_("Filter \"" + filter_name + "\" unknown.");
    if (camera_name != "perspective") return fail_("Camera \"" + camera_name + "\" unknown.");
    {  // PerspectiveCamera::create, camera.rs:86-107: "frameaspectratio" (default xres / yres) gives the screen window, "screenwindow" overrides it
      const float frame = camera_p.one_float("frameaspectratio", (float)o.xres / (float)o.yres);
      const bool explicit_frame = camera_p.find("frameaspectratio", {"float"}) != nullptr;
      float sw[4];
      if (frame > 1.0f) { sw[0] = -frame; sw[1] = frame; sw[2] = -1.0f; sw[3] = 1.0f; } else { sw[0] = -1.0f; sw[1] = 1.0f; sw[2] = -1.0f / frame; sw[3] = 1.0f / frame; }
      bool explicit_window = false;
      if (const std::vector<float>* w = camera_p.floats("screenwindow", {"float"})) {
        if (w->size() == 4) { for (int k = 0; k < 4; ++k) sw[k] = (*w)[k]; explicit_window = true; } else warn("screenwindow should have 4 values");
      }
      if (explicit_frame || explicit_window) { if (!(sw[1] > sw[0])) return fail_("Camera: empty screen window"); memcpy(o.screen_window, sw, 16); }
    }
    float fov = camera_p.one_float("fov", 90.0f); const float half = camera_p.one_float("halffov", -1.0f);  // camera.rs:108-112
    if (half > 0.0f) fov = 2.0f * half;
    o.fov = fov; o.lens_radius = camera_p.one_float("lensradius", 0.0f); o.focal_distance = camera_p.one_float("focaldistance", 1e6f);
    memcpy(o.cam_to_world, camera_to_world.m.a, 64); memcpy(o.cam_to_world_inv, camera_to_world.inv.a, 64);
    if (sampler_name != "lowdiscrepancy" && sampler_name != "02sequence") return fail_("Sampler \"" + sampler_name + "\" unknown.");  // api.rs:205-215
    o.spp = sampler_p.one_int("pixelsamples", 16); o.sampler_dims = sampler_p.one_int("dimensions", 4);
    if (integrator_name != "path") return fail_("Integrator \"" + integrator_name + "\" is not supported (this backend is the path integrator)");
    o.max_depth = integrator_p.one_int("maxdepth", 5); o.rr_threshold = integrator_p.one_float("rrthreshold", 1.0f);
    const std::string ls = integrator_p.one_string("lightsamplestrategy", "spatial");
    // PathIntegrator::preprocess (path.rs:86-94) compares the string with "uniform" only: every other value - "spatial", pbrt's "power", a typo -
    // gets the SpatialLightDistribution
    o.light_strategy = ls == "uniform" ? 1 : 0;
    if (const std::vector<float>* pb = integrator_p.floats("pixelbounds", {"integer"})) {
      if (pb->size() == 4) { for (int k = 0; k < 4; ++k) o.pixel_bounds[k] = (int32_t)(*pb)[k]; o.has_pixel_bounds = 1; }
      else warn("pixelbounds expects 4 values");
    }
    if (accel_name == "kdtree") return fail_("Accelerator \"kdtree\" is not implemented (nor in the reference, api.rs:262-263)");
    if (accel_name != "bvh") warn("unknown accelerator, using bvh");  // api.rs:266-269
    const std::string split = accel_p.one_string("splitmethod", "sah");  // bvh/mod.rs:63-78
    if (split == "middle") return fail_("BVH split method \"middle\" is not supported");
    if (split != "sah") warn("unknown split method, using sah");
    max_prims = accel_p.one_int("maxnodeprims", 4);
    o.rank = 0; o.world_size = 1; o.flags = 0;
    return true;
  }

  // ------------------------------------------------------------------ directives (rc/api.rs:516-1090)
  bool run(const std::vector<Tok>& t, int depth) {
    size_t i = 0;
    auto nums = [&](size_t n, float* out) -> bool {
      size_t got = 0; bool bracket = false;
      if (i < t.size() && t[i].kind == 2) { bracket = true; ++i; }
      while (got < n && i < t.size()) { float v; if (t[i].kind != 0 || !is_number(t[i].s, v)) break; out[got++] = v; ++i; }
      if (bracket) { if (i >= t.size() || t[i].kind != 3) return false; ++i; }
      return got == n;
    };
    auto str = [&](std::string& out) -> bool { if (i < t.size() && t[i].kind == 1) { out = t[i++].s; return true; } return false; };
    while (i < t.size()) {
      if (t[i].kind != 0) return fail_("expected a directive, found \"" + t[i].s + "\"");
      const std::string d = t[i++].s;
      float f[16]; std::string a, b, c; PbrtParams p;
      if (d == "Identity") ctm = Xf{mat_identity(), mat_identity()};
      else if (d == "Translate") { if (!nums(3, f)) return fail_("Translate needs 3 numbers"); ctm = xf_mul(ctm, xf_translate(f[0], f[1], f[2])); }
      else if (d == "Scale") { if (!nums(3, f)) return fail_("Scale needs 3 numbers"); ctm = xf_mul(ctm, xf_scale(f[0], f[1], f[2])); }
      else if (d == "Rotate") {  // Transform::rotate, transform.rs:30-56
        if (!nums(4, f)) return fail_("Rotate needs 4 numbers");
        float len = std::sqrt(f[1] * f[1] + f[2] * f[2] + f[3] * f[3]);
        const float ax = f[1] / len, ay = f[2] / len, az = f[3] / len;
        const float rad = f[0] * (3.14159265358979323846f / 180.0f), s = std::sin(rad), co = std::cos(rad);
        Mat m = mat_identity();
        m.a[0][0] = ax * ax + (1.0f - ax * ax) * co; m.a[0][1] = ax * ay * (1.0f - co) - az * s; m.a[0][2] = ax * az * (1.0f - co) + ay * s;
        m.a[1][0] = ax * ay * (1.0f - co) + az * s; m.a[1][1] = ay * ay + (1.0f - ay * ay) * co; m.a[1][2] = ay * az * (1.0f - co) - ax * s;
        m.a[2][0] = ax * az * (1.0f - co) - ay * s; m.a[2][1] = ay * az * (1.0f - co) + ax * s; m.a[2][2] = az * az + (1.0f - az * az) * co;
        Mat tr; for (int r = 0; r < 4; ++r) for (int q = 0; q < 4; ++q) tr.a[r][q] = m.a[q][r];
        ctm = xf_mul(ctm, Xf{m, tr});
      } else if (d == "LookAt") {
        if (!nums(9, f)) return fail_("LookAt needs 9 numbers");
        Xf la; if (rtxh_look_at(f, f + 3, f + 6, &la.m.a[0][0], &la.inv.a[0][0]) != RT_OK) return fail_(rtxh_last_error());
        ctm = xf_mul(ctm, la);
      } else if (d == "Transform" || d == "ConcatTransform") {  // column-major in the file (api.rs:596-600)
        if (!nums(16, f)) return fail_(d + " needs 16 numbers");
        Mat m; for (int r = 0; r < 4; ++r) for (int q = 0; q < 4; ++q) m.a[r][q] = f[4 * q + r];
        Xf x{m, mat_inverse(m)};
        ctm = d == "Transform" ? x : xf_mul(ctm, x);
      } else if (d == "CoordinateSystem") { if (!str(a)) return fail_("CoordinateSystem needs a name"); named_cs[a] = ctm; }
      else if (d == "CoordSysTransform") { if (!str(a)) return fail_("CoordSysTransform needs a name"); auto it = named_cs.find(a); if (it != named_cs.end()) ctm = it->second; else warn("unknown coordinate system"); }
      else if (d == "Camera" || d == "Film" || d == "Sampler" || d == "Integrator" || d == "PixelFilter" || d == "Accelerator") {
        if (in_world) return fail_("Options cannot be set inside world block.");
        if (!str(a) || !parse_params(t, i, p)) return fail_(err.empty() ? d + " needs a name" : err);
        if (d == "Camera") { camera_name = a; camera_p = p; camera_to_world = xf_inverse(ctm); named_cs["camera"] = camera_to_world; }
        else if (d == "Film") { film_name = a; film_p = p; }
        else if (d == "Sampler") { sampler_name = a; sampler_p = p; }
        else if (d == "Integrator") { integrator_name = a; integrator_p = p; }
        else if (d == "PixelFilter") { filter_name = a; filter_p = p; }
        else { accel_name = a; accel_p = p; }
      } else if (d == "WorldBegin") { if (in_world) return fail_("WorldBegin inside world block"); in_world = true; named_cs["world"] = ctm; ctm = Xf{mat_identity(), mat_identity()}; }
      else if (d == "WorldEnd") { if (!in_world) return fail_("Scene description must be inside world block."); world_ended = true; return true; }
      else if (d == "AttributeBegin") { if (!in_world) return fail_("Scene description must be inside world block."); pushed_gs.push_back(gs); pushed_transforms.push_back(ctm); }
      else if (d == "AttributeEnd") { if (pushed_gs.empty()) { warn("unmatched AttributeEnd"); continue; } gs = pushed_gs.back(); pushed_gs.pop_back(); ctm = pushed_transforms.back(); pushed_transforms.pop_back(); }
      else if (d == "TransformBegin") pushed_transforms.push_back(ctm);
      else if (d == "TransformEnd") { if (pushed_transforms.empty()) { warn("unmatched TransformEnd"); continue; } ctm = pushed_transforms.back(); pushed_transforms.pop_back(); }
      else if (d == "ReverseOrientation") gs.reverse_orientation = !gs.reverse_orientation;
      else if (d == "Texture") {
        if (!in_world) return fail_("Scene description must be inside world block.");
        if (!str(a) || !str(b) || !str(c) || !parse_params(t, i, p)) return fail_(err.empty() ? "Texture needs name, type and class" : err);
        if (!make_texture(a, b, c, p)) return false;
      } else if (d == "Material") { if (!str(a) || !parse_params(t, i, p)) return fail_(err.empty() ? "Material needs a name" : err); gs.material = a; gs.material_params = p; gs.current_named_material.clear(); }
      else if (d == "MakeNamedMaterial") {
        if (!str(a) || !parse_params(t, i, p)) return fail_(err.empty() ? "MakeNamedMaterial needs a name" : err);
        const std::string type = p.one_string("type", "");
        if (type.empty()) return fail_("No parameter string \"type\" found in named_material");
        const int m = make_material(type, p, PbrtParams());
        if (m < 0) return fail_(err.empty() ? rtxh_last_error() : err);
        gs.named_materials[a] = m;
      } else if (d == "NamedMaterial") { if (!str(a)) return fail_("NamedMaterial needs a name"); gs.current_named_material = a; }
      else if (d == "LightSource") { if (!in_world) return fail_("Scene description must be inside world block."); if (!str(a) || !parse_params(t, i, p)) return fail_(err.empty() ? "LightSource needs a name" : err); if (!add_light(a, p)) return false; }
      else if (d == "AreaLightSource") { if (!str(a) || !parse_params(t, i, p)) return fail_(err.empty() ? "AreaLightSource needs a name" : err); gs.area_light = a; gs.area_light_params = p; }
      else if (d == "Shape") { if (!in_world) return fail_("Scene description must be inside world block."); if (!str(a) || !parse_params(t, i, p)) return fail_(err.empty() ? "Shape needs a name" : err); if (!add_shape(a, p)) return false; }
      else if (d == "Include") {
        if (!str(a)) return fail_("Include needs a file name");
        if (depth > 16) return fail_("Include nesting too deep");
        std::string text; if (!read_file(resolve(a), text)) return fail_("cannot open " + a);
        std::vector<Tok> sub; if (!tokenize(text, sub, err)) return false;
        if (!run(sub, depth + 1)) return false;
        if (world_ended) return true;
      } else if (d == "ObjectBegin") {  // api.rs:1018-1032
        if (!in_world) return fail_("Scene description must be inside world block.");
        if (!str(a)) return fail_("ObjectBegin needs a name");
        pushed_gs.push_back(gs); pushed_transforms.push_back(ctm);
        if (in_instance) return fail_("ObjectBegin called inside of instance definition");
        in_instance = true; current_instance = a; instances[a] = PbrtSoup();
        object_ids.erase(a);  // a name defined again: later ObjectInstance calls place the new definition (instances.insert(name, Vec::new()), api.rs:1030)
      } else if (d == "ObjectEnd") {  // api.rs:1034-1049
        if (!in_instance) return fail_("ObjectEnd called outside of instance definition");
        in_instance = false; current_instance.clear();
        if (!pushed_gs.empty()) { gs = pushed_gs.back(); pushed_gs.pop_back(); ctm = pushed_transforms.back(); pushed_transforms.pop_back(); }
      } else if (d == "ObjectInstance") {
        if (!in_world) return fail_("Scene description must be inside world block.");
        if (!str(a)) return fail_("ObjectInstance needs a name");
        if (in_instance) return fail_("ObjectInstance called inside of instance definition");
        if (!instantiate(a)) return false;
      }
      else if (d == "MakeNamedMedium" || d == "MediumInterface" || d == "TransformTimes" || d == "ActiveTransform") return fail_("directive " + d + " is not supported");
      else return fail_("unknown directive \"" + d + "\"");
    }
    return true;
  }
  static bool read_file(const std::string& path, std::string& out) {
    FILE* f = fopen(path.c_str(), "rb"); if (!f) return false;
    char buf[65536]; size_t n; out.clear();
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) out.append(buf, n);
    fclose(f); return true;
  }
};

int pbrt_load_text(const std::string& text, const std::string& base_dir, rtxh_pbrt_result* out) {
  memset(out, 0, sizeof *out);
  PbrtLoader L; L.base_dir = base_dir; L.rp = &out->params;
  L.scene = rtxh_scene_new();
  if (!L.scene) return fail(RT_ERR_INVALID, "out of memory");
  std::vector<PbrtLoader::Tok> toks;
  bool ok = PbrtLoader::tokenize(text, toks, L.err) && L.run(toks, 0);
  if (ok && !L.world_ended) { L.err = "missing WorldEnd"; ok = false; }
  if (ok) ok = L.finish_options();
  if (ok && L.world.idx.empty() && L.n_spheres == 0 && L.n_instances == 0) { L.err = "the scene holds no triangles"; ok = false; }
  if (ok) {
    const PbrtSoup& w = L.world;
    const int32_t nv = (int32_t)w.n_verts(), nt = (int32_t)(w.idx.size() / 3);
    if (nt > 0 && rtxh_scene_set_mesh(L.scene, w.P.data(), nv, w.idx.data(), nt, w.any_n ? w.N.data() : nullptr, w.any_uv ? w.UV.data() : nullptr, w.any_s ? w.S.data() : nullptr,
                            w.tri_mat.data(), w.tri_light.data(), w.tri_flags.data()) != RT_OK) { L.err = rtxh_last_error(); ok = false; }
    bool any_mask = false; for (int32_t a : w.tri_alpha) any_mask |= a >= 0;
    if (ok && any_mask && rtxh_scene_set_alpha(L.scene, w.tri_alpha.data()) != RT_OK) { L.err = rtxh_last_error(); ok = false; }
    if (ok && rtxh_scene_commit(L.scene, L.max_prims) != RT_OK) { L.err = rtxh_last_error(); ok = false; }
  }
  if (!ok) { std::string m = L.err.empty() ? std::string("pbrt: parse error") : "pbrt: " + L.err; return fail(RT_ERR_INVALID, m); }
  out->scene = L.scene; L.scene_handed_over = true; out->max_prims_per_node = L.max_prims; out->n_warnings = L.warnings;
  snprintf(out->film_filename, sizeof out->film_filename, "%s", L.film_filename.c_str());
  if (L.warnings > 0) g_err = "pbrt warning: " + L.first_warning; else g_err.clear();
  return RT_OK;
}

}  // namespace

// no C++ exception may cross the C ABI: a damaged file can ask for any amount of memory
static int pbrt_load_guarded(const std::string& text, const std::string& base_dir, rtxh_pbrt_result* out) {
  try { return pbrt_load_text(text, base_dir, out); }
  catch (const std::bad_alloc&) { memset(out, 0, sizeof *out); return fail(RT_ERR_OOM, "pbrt: out of memory while building the scene"); }
  catch (const std::exception& e) { memset(out, 0, sizeof *out); return fail(RT_ERR_INVALID, std::string("pbrt: ") + e.what()); }
}
void rtxh_set_flatten_instances(int32_t on) { g_flatten_instances = on != 0; }
int rtxh_pbrt_load(const char* path, rtxh_pbrt_result* out) {
  if (!path || !out) return fail(RT_ERR_INVALID, "null argument");
  g_err.clear();
  std::string text;
  if (!PbrtLoader::read_file(path, text)) return fail(RT_ERR_INVALID, std::string("cannot open ") + path);
  std::string dir = path; size_t sl = dir.find_last_of('/'); dir = sl == std::string::npos ? std::string() : dir.substr(0, sl);
  return pbrt_load_guarded(text, dir, out);
}
int rtxh_pbrt_parse(const char* text, const char* base_dir, rtxh_pbrt_result* out) {
  if (!text || !out) return fail(RT_ERR_INVALID, "null argument");
  g_err.clear();
  return pbrt_load_guarded(text, base_dir ? base_dir : "", out);
}
int rtxh_pbrt_tokens(const char* text, char* out, uint64_t capacity) {
  if (!text) return fail(RT_ERR_INVALID, "null argument");
  g_err.clear();
  std::vector<PbrtLoader::Tok> toks; std::string err;
  if (!PbrtLoader::tokenize(text, toks, err)) return fail(RT_ERR_INVALID, "pbrt: " + err);
  std::string dump;
  for (const PbrtLoader::Tok& t : toks) {
    float v; char buf[64];
    if (t.kind == 1) { std::string e; for (char c : t.s) { if (c == '\n') e += "\\n"; else e += c; } dump += "S " + e + "\n"; }
    else if (t.kind == 2 || t.kind == 3) dump += t.s + "\n";
    else if (PbrtLoader::is_number(t.s, v)) { snprintf(buf, sizeof buf, "N %.9g\n", v); dump += buf; }
    else dump += "K " + t.s + "\n";
  }
  if (out && capacity) { snprintf(out, capacity, "%s", dump.c_str()); }
  return (int)toks.size();
}
