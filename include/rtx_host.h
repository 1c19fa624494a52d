/* rtx_host.h — host-side layer above the C ABI of rtx_hip.h.
 *
 * In the reference this layer is Rust: `RealApi` accumulates shapes, materials and lights
 * (rustracer-core/src/api.rs:913-966), `world_end` builds camera + film + sampler + integrator +
 * Scene(BVH) (api.rs:977-1010) and calls `renderer::render`. No Rust toolchain exists in this
 * image, so the same responsibilities are written in C++ here, with the reference's names and
 * defaults, and everything device-side goes through rtx_hip.h:
 *   - BVH::new / recursive_build (SAH, 12 buckets) / flatten_bvh      rc/bvh/mod.rs:80-358
 *   - TriangleMesh world-space vertices, one DiffuseAreaLight per emissive triangle  rc/api.rs:933-946
 *   - MIPMap pyramid construction (power-of-two images)                rc/mipmap.rs:158-187
 *   - InfiniteAreaLight sampling distribution                          rc/light/infinite.rs:78-101
 *   - PerspectiveCamera::new, Film::new (filter table, bounds), filters rc/camera.rs:30-72, rc/film.rs:58-115,249-257
 *   - Scene::new light preprocess (world bounding sphere)              rc/scene.rs:29-49
 * This library contains no rendering arithmetic of its own: radiance comes from the HIP kernels only.
 */
#ifndef RTX_HOST_H
#define RTX_HOST_H
#include <stdint.h>
#include "rtx_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rtxh_scene rtxh_scene;

typedef struct rtxh_render_params {
  int32_t xres, yres;
  float crop[4];           /* cropwindow xmin xmax ymin ymax                       film.rs:126-136 */
  int32_t filter_kind;     /* 0 box 1 triangle 2 gaussian 3 mitchell               rc/filter       */
  float filter_params[4];  /* xwidth ywidth alpha|B C                                             */
  float film_scale, max_sample_luminance;
  float cam_to_world[16], cam_to_world_inv[16]; /* Transform{m, m_inv} of CTM.inverse()            */
  float fov, lens_radius, focal_distance;       /* camera.rs:84-113                                */
  int32_t spp, sampler_dims;                    /* zerotwosequence.rs:58-63                        */
  int32_t max_depth; float rr_threshold; int32_t light_strategy; /* path.rs:49-53                  */
  int32_t pixel_bounds[4]; /* "pixelbounds" x0 x1 y0 y1, used when has_pixel_bounds != 0                     */
  int32_t rank, world_size; /* film sharding (tile rows interleaved over ranks)                    */
  uint32_t flags;           /* RT_FLAG_*                                                           */
  float screen_window[4];   /* xmin xmax ymin ymax; xmax <= xmin => create()'s default from the aspect ratio, camera.rs:86-107 */
  int32_t has_pixel_bounds; /* PathIntegrator::create (path.rs:53-69) intersects the sample bounds with "pixelbounds" whenever the
                               parameter is given: a degenerate or non-overlapping window renders nothing, it is not ignored */
} rtxh_render_params;

rtxh_scene* rtxh_scene_new(void);
void rtxh_scene_free(rtxh_scene*);
/* Triangle soup of all `Shape "trianglemesh"` calls, world space; tri_light = index into the light
 * list of that triangle's DiffuseAreaLight or -1; tri_flags = RT_TRI_* (per-mesh attributes). */
int rtxh_scene_set_mesh(rtxh_scene*, const float* P, int32_t n_verts, const int32_t* indices, int32_t n_tris, const float* N, const float* UV,
                        const float* S, const int32_t* tri_material, const int32_t* tri_light, const uint8_t* tri_flags);
/* Shape "sphere" (rc/shapes/sphere.rs:53-68): radius, zmin, zmax, phimax (degrees) as the parameters give them (Sphere::new clamps and converts),
 * under the object-to-world transform o2w (row-major 4x4, with its inverse as Transform{m, m_inv} holds it). `light`: index of its DiffuseAreaLight in the light list or -1;
 * that light is added by rtxh_scene_add_light(RT_LIGHT_DIFFUSE_AREA, tri = -2 - k, ...) for sphere k. Returns k. */
int rtxh_scene_add_sphere(rtxh_scene*, const float* o2w16, const float* w2o16, float radius, float z_min, float z_max, float phi_max, int32_t reverse_orientation,
                          int32_t material, int32_t light);
/* The reference's other quadrics through the same record: kind 0 = sphere (as above), 1 = Shape "disk" (z_min := height, z_max := innerradius;
 * rc/shapes/disk.rs:48-62), 2 = Shape "cylinder" (z_min, z_max as given; rc/shapes/cylinder.rs:26-46). Area lights as for spheres (tri = -2 - k). */
int rtxh_scene_add_quadric(rtxh_scene*, int32_t kind, const float* o2w16, const float* w2o16, float radius, float z_min, float z_max, float phi_max,
                           int32_t reverse_orientation, int32_t material, int32_t light);
/* Scene files (rtxh_pbrt_load / rtxh_pbrt_parse), per calling thread: on != 0 writes every ObjectInstance out as world-space triangles (memory in
 * proportion to instances x mesh, but every ray stays in the single-level traversal kernels); 0 - the default - keeps the reference's form, one tree per
 * object and a TransformedPrimitive per instance (objects whose meshes carry alpha masks are written out either way). */
void rtxh_set_flatten_instances(int32_t on);
/* An emitter that is in no light list: a shape with an AreaLightSource inside an ObjectBegin block keeps its DiffuseAreaLight - it glows when a camera ray
 * or a specular bounce reaches it - but the light never enters Scene::lights (rc/api.rs:954-964). rtxh_scene_add_emitter returns its index k;
 * rtxh_scene_object_emitters gives every triangle of an object its emitter (k or -1); a top-level triangle names emitter k as light -2 - k in
 * rtxh_scene_set_mesh's tri_light (what a written-out instance of such an object becomes). */
int rtxh_scene_add_emitter(rtxh_scene*, const float* rgb3, int32_t two_sided);
int rtxh_scene_object_emitters(rtxh_scene*, int32_t object, const int32_t* tri_emitter /* one per triangle of the object */);
/* ObjectBegin ... ObjectEnd (rc/api.rs:1019-1051): a triangle mesh in OBJECT space (vertices as the shapes' own CTMs leave them), arrays as for
 * rtxh_scene_set_mesh without lights (an emitting shape inside an object definition is an unlisted emitter: rtxh_scene_object_emitters above).
 * Returns the object's index; nothing is rendered until an instance places it. */
int rtxh_scene_add_object(rtxh_scene*, const float* P, int32_t nv, const int32_t* idx, int32_t nt, const float* N, const float* UV, const float* S,
                          const int32_t* tri_material, const uint8_t* tri_flags);
/* ObjectInstance (rc/api.rs:1053-1090): a TransformedPrimitive over `object` with primitive_to_world = o2w (and its inverse w2o, row-major 4x4). The
 * object's tree is traversed in object space (rc/primitive.rs:90-101) - nothing is copied. Returns the instance's index. */
int rtxh_scene_add_instance(rtxh_scene*, int32_t object, const float* o2w16, const float* w2o16);
/* What else an object definition may hold (round 6; rc/api.rs:1019-1051 collects EVERY primitive between ObjectBegin and ObjectEnd, and rc/primitive.rs:79-118 wraps
 * whatever that is): a quadric in object space - o2w / w2o = the CTM inside the definition; emitter = -1 or an unlisted emitter (rtxh_scene_add_emitter); returns its
 * primitive id inside the object, after the object's triangles - and alpha / shadowalpha masks on the object's triangles ({alpha, shadowalpha} float-texture ids or -1
 * per triangle; NULL removes them). rtxh_scene_add_object with nt == 0 creates an object that holds quadrics only. */
int rtxh_object_add_quadric(rtxh_scene*, int32_t object, int32_t kind, const float* o2w16, const float* w2o16, float radius, float z_min, float z_max, float phi_max,
                            int32_t reverse_orientation, int32_t material, int32_t emitter);
int rtxh_object_set_alpha(rtxh_scene*, int32_t object, const int32_t* tri_alpha2);
/* Alpha masks of the meshes, after rtxh_scene_set_mesh: per triangle {alpha, shadowalpha} float-texture ids or -1 (TriangleMesh::create,
 * rc/shapes/mesh.rs:134-156: a named float texture, or the constant 0 when the float parameter is 0). NULL removes all masks. */
int rtxh_scene_set_alpha(rtxh_scene*, const int32_t* tri_alpha2);
/* on != 0: MIP pyramids (rtxh_scene_add_mipmap) and environment-map sampling tables (rtxh_scene_add_light, infinite) are built
 * by the GPU (rt_mip_build, rt_env_distribution in rtx_hip.h) - bit-identical tables, milliseconds instead of tenths of a second for
 * a 2048 x 1024 map. Off by default: the host build needs no device. The setting belongs to the calling thread (scene builds on
 * different threads do not affect each other) and applies to the scenes that thread builds afterwards, rtxh_pbrt_load included. */
void rtxh_set_device_ingest(int32_t on);
int rtxh_scene_add_mipmap(rtxh_scene*, int32_t width, int32_t height, const float* rgb, int32_t trilinear, float max_aniso, int32_t wrap);
int rtxh_scene_add_texture(rtxh_scene*, int32_t kind, const float* value3, int32_t tex1, int32_t tex2, int32_t amount, int32_t mip, const float* mapping4);
int rtxh_scene_add_material(rtxh_scene*, int32_t kind, const int32_t* slots16, int32_t remap_roughness, int32_t bump_texture /* or -1 */);
int rtxh_scene_add_light(rtxh_scene*, int32_t kind, int32_t tri, const float* rgb3, int32_t two_sided, const float* vec3, int32_t mip,
                         const float* l2w16, const float* w2l16);
/* BVH::create(prims, "sah", maxnodeprims) + Scene::new; flattens everything into an rt_scene_desc. */
int rtxh_scene_commit(rtxh_scene*, int32_t max_prims_per_node);
/* Same, with the tree built on the GPU by rt_bvh_build (linear BVH: milliseconds instead of seconds for 10^6
 * triangles, more node visits per ray afterwards; rtx_hip.h). Fails without a GPU. ms_device (may be NULL): kernel time. */
int rtxh_scene_commit_device_bvh(rtxh_scene*, int32_t max_prims_per_node, float* ms_device);
/* Uploads the flattened scene to the current HIP device (rt_scene_create). Fails without a GPU. */
int rtxh_scene_upload(rtxh_scene*, int32_t device);

/* Inspection of the host-side products (CPU only; used by the parity tests against the oracle). */
int rtxh_scene_bvh_sizes(rtxh_scene*, int32_t* n_nodes, int32_t* n_prims);
int rtxh_scene_bvh_get(rtxh_scene*, float* bounds6, uint32_t* offset, uint16_t* n_prims, uint8_t* axis, int32_t* ordered);
int rtxh_camera_film_setup(const rtxh_render_params*, float* raster_to_camera16, float* dxdy6, float* filter_table256, int32_t* sample_bounds4,
                           int32_t* cropped4);
int rtxh_mip_level(rtxh_scene*, int32_t mip, int32_t level, int32_t* w, int32_t* h, float* rgb_out /* may be NULL */);
/* Copies of the unflattened tables (tests compare a parsed .pbrt scene with the same scene built call by call).
 * out == NULL: only *n_items. Items: rt_texture, rt_material, rtxh_light_info, float[3] (P N S), float[2] (UV),
 * int32[3] (indices), int32 (tri material / light), uint8 (tri flags). */
enum { RTXH_TABLE_TEXTURES = 0, RTXH_TABLE_MATERIALS, RTXH_TABLE_LIGHTS, RTXH_TABLE_P, RTXH_TABLE_N, RTXH_TABLE_UV, RTXH_TABLE_S, RTXH_TABLE_INDICES,
       RTXH_TABLE_TRI_MATERIAL, RTXH_TABLE_TRI_LIGHT, RTXH_TABLE_TRI_FLAGS,
       /* sampling tables of the first infinite light (floats): func, per-row cdf, row integrals, marginal cdf */
       RTXH_TABLE_ENV_FUNC, RTXH_TABLE_ENV_CDF, RTXH_TABLE_ENV_ROW_INT, RTXH_TABLE_ENV_MARG_CDF,
       RTXH_TABLE_INSTANCES,         /* rtxh_instance_info per ObjectInstance, in order */
       RTXH_TABLE_EMITTERS,          /* rtxh_emitter_info per emitter that is in no light list (rtxh_scene_add_emitter) */
       RTXH_TABLE_QUADRICS,          /* per quadric (rtxh_scene_add_quadric): rt_sphere, then int32 material, int32 light (>= 0 a listed light, -2 - k unlisted emitter k, -1 none) */
       RTXH_TABLE_OBJECT_BASE = 1000 /* + 8 * object + {0 P, 1 N, 2 UV, 3 S, 4 indices, 5 tri material, 6 tri flags, 7 tri emitter (k or -1; empty when no
                                        triangle of the object emits)}: the object-space soup of one object */,
       RTXH_TABLE_OBJECT_EXTRA_BASE = 100000 /* + 2 * object + {0 the object's quadrics (as RTXH_TABLE_QUADRICS; light: -1 or -2 - k), 1 {alpha, shadowalpha} per triangle (empty: none)} */ };
typedef struct rtxh_emitter_info { float rgb[3]; int32_t two_sided; } rtxh_emitter_info;
typedef struct rtxh_instance_info { int32_t object; float o2w[16], w2o[16]; } rtxh_instance_info;
typedef struct rtxh_light_info { int32_t kind, tri; float rgb[3]; int32_t two_sided; float vec[3]; int32_t mip; float l2w[12], w2l[12]; } rtxh_light_info;
int rtxh_scene_inspect(rtxh_scene*, int32_t table, void* out, uint64_t capacity_bytes, uint64_t* n_items);
int rtxh_look_at(const float* pos, const float* look, const float* up, float* m16, float* m_inv16);

/* renderer::render through the HIP backend. film_xyzw: (y1-y0)*(x1-x0)*4 floats over the cropped pixel bounds
 * (host pointer, or device pointer with RT_FLAG_FILM_ON_DEVICE). */
int rtxh_render(rtxh_scene*, const rtxh_render_params*, void* hip_stream, float* film_xyzw, rt_stats* stats);
/* The same frame on several GPUs of this process (rt_multi_* in rtx_hip.h): the scene is replicated on `devices` (kept for later calls with the
 * same list), host threads pull chunks of tile rows, the film is gathered on devices[0] and returned in host memory (or in memory of devices[0]
 * with RT_FLAG_FILM_ON_DEVICE). rank / world_size of the parameters are ignored. */
int rtxh_render_multi(rtxh_scene*, const rtxh_render_params*, const int32_t* devices, int32_t n_devices, int32_t chunks_per_device, float* film_xyzw,
                      rt_stats* total, rt_stats* per_device /* n_devices entries or NULL */);
/* Kernel-level pass-throughs on the uploaded scene (prim indices are leaf-order). */
int rtxh_trace(rtxh_scene*, const float* rays, uint64_t n, int32_t any_hit, float* hits4_or_occ, uint64_t counters[2]);
int rtxh_trace_device(rtxh_scene*, const void* d_rays, uint64_t n, void* d_hits, int32_t reps, void* hip_stream, float* ms_per_launch);
int rtxh_light_distribution(rtxh_scene*, int32_t n_voxels[3], float* func, float* cdf, float* func_int);
/* rt_scene_query on the uploaded scene (uploads it first if need be; rtx_hip.h: RT_QUERY_*). */
int rtxh_scene_query(rtxh_scene*, int32_t what);
/* rt_link_tables on the scene's description (host only: nothing is uploaded). */
int rtxh_scene_link_tables(rtxh_scene*, int32_t mid, uint32_t* link_kept, uint32_t* link_full, uint64_t capacity_words, double* stats);
/* sizeof() of a struct of this header or of rtx_hip.h by its C name, or -1 (a binding checks its mirrors against the library it loaded). */
int rtxh_sizeof(const char* struct_name);
const char* rtxh_last_error(void);

/* ---- input formats on the caller's side of the path ------------------------------------
 * plymesh::create (rc/shapes/plymesh.rs:18-186, parser = ply-rs 0.1.3): ascii, binary_little_endian and
 * binary_big_endian PLY with elements `vertex` and `face`. As in the reference only `float` vertex properties are
 * taken (x y z required; nx ny nz; u v | s t | texture_u texture_v | texture_s texture_t), other properties are
 * parsed and ignored; `face.vertex_indices` must be a list of int or uint (any count type); triangles are kept, a
 * quad (a b c d) becomes (a b c) (d a c) (:101-123), other faces are dropped; any other element is an error (the
 * reference panics, :97). Returns arrays owned by the library until rtxh_ply_free. */
typedef struct rtxh_ply {
  int32_t n_verts, n_tris;
  float* P;       /* n_verts x 3 */
  float* N;       /* n_verts x 3 or NULL */
  float* UV;      /* n_verts x 2 or NULL */
  int32_t* idx;   /* n_tris x 3 */
  int32_t n_dropped_faces;
} rtxh_ply;
int rtxh_ply_read(const char* path, rtxh_ply* out);
void rtxh_ply_free(rtxh_ply* ply);
/* read_image_pfm (rc/imageio.rs:179-246): "PF" (rgb) or "Pf" (grey, replicated) header, width, height, scale whose
 * sign gives the byte order and whose magnitude multiplies the samples when != 1; rows are flipped so that row 0 is
 * the top of the image. rgb: width*height*3 floats owned by the library until rtxh_free. */
int rtxh_pfm_read(const char* path, int32_t* width, int32_t* height, float** rgb);
void rtxh_free(void* p);
/* read_image (rc/imageio.rs:16-33): decoder by extension - png PNG tga TGA (8-bit RGB / 255 after the `image` crate's
 * to_rgb8: palette and grey expanded, alpha dropped, 16-bit samples (c + 128) / 257; imageio.rs:94-112), hdr (Radiance
 * RGBE, c * 2^(e - 136); imageio.rs:114-132), pfm (above), exr EXR (imageio.rs:134-160: first layer with R, G, B;
 * scan-line files with NONE / RLE / ZIPS / ZIP compression and half / float / uint samples; PIZ, PXR24, B44, DWA, tiled and
 * multi-part files fail with RT_ERR_UNSUPPORTED and a message naming the feature). Any other extension:
 * "Unsupported file format", as the reference. Row 0 = top of the image. */
int rtxh_image_read(const char* path, int32_t* width, int32_t* height, float** rgb);

/* ---- sampled spectra (rc/spectrum.rs:108-126,168-211, rc/cie.rs): the reference's Spectrum is RGB; spectral data is converted once, when the
 * scene is described. from_sampled: n (wavelength nm, value) samples, increasing wavelengths, -> RGB through the CIE 1931 observer. blackbody:
 * `scale * from_sampled(normalised Planck curve at T)` as a "blackbody" parameter gives (paramset.rs:291-310). copper: the default eta and k of
 * Material "metal" (Metal::create, rc/material/metal.rs:25-29 over the measured tables :84-200). */
int rtxh_spectrum_from_sampled(const float* lambda_nm, const float* values, int32_t n, float rgb[3]);
int rtxh_spectrum_blackbody(float temperature_kelvin, float scale, float rgb[3]);
void rtxh_copper(float eta_rgb[3], float k_rgb[3]);

/* ---- pbrt-v3 scene description ----------------------------------------------------------
 * What `rustracer scene.pbrt` does before renderer::render: tokenise + parse (rc/pbrt/lexer.rs:185-275,
 * rc/pbrt/parser.rs:20-320), run the directive state machine of RealApi (rc/api.rs:516-1010: CTM, named coordinate
 * systems, attribute / transform stacks, GraphicsState with named textures and materials, RenderOptions) and build
 * camera, film, filter, sampler, integrator, shapes, materials, textures and lights from their parameter lists
 * (RenderOptions::make_* rc/api.rs:179-256, make_shapes / make_material / make_*_texture rc/api.rs:1093-1259,
 * TextureParams rc/paramset.rs:356-470, and each module's create()). The result is a committed rtxh_scene (BVH
 * built with the file's Accelerator parameters) plus the rtxh_render_params rtxh_render takes.
 * Errors like the reference's (unknown Film / Filter / Sampler / Camera, options inside the world block, ...) and
 * every directive or class this backend does not implement (non-triangle shapes, object instancing, integrators
 * other than "path", spectral parameter types, alpha masks) fail with a message; nothing is skipped silently.
 * n_warnings counts the conditions the reference only logs (missing named texture, unknown material -> matte, ...); after a successful load with
 * n_warnings > 0 rtxh_last_error() holds the text of the first one. */
typedef struct rtxh_pbrt_result {
  rtxh_scene* scene;            /* owned by the caller: rtxh_scene_free */
  rtxh_render_params params;
  int32_t max_prims_per_node;   /* Accelerator "integer maxnodeprims" (bvh/mod.rs:76), already applied */
  int32_t n_warnings;
  char film_filename[512];      /* Film::create naming: "rt-" + filename, or "image.png" (film.rs:118-123) */
} rtxh_pbrt_result;
int rtxh_pbrt_load(const char* path, rtxh_pbrt_result* out);
/* Same from memory; relative file names (Include, plymesh, imagemap, mapname) resolve against base_dir. */
int rtxh_pbrt_parse(const char* text, const char* base_dir, rtxh_pbrt_result* out);
/* The lexer alone (rc/pbrt/lexer.rs:185-275; comments dropped as in rc/pbrt/mod.rs:39-43): one token per line,
 * "K <word>", "N <number %.9g>", "S <string>", "[" or "]". Returns the token count or < 0. */
int rtxh_pbrt_tokens(const char* text, char* out, uint64_t capacity);

#ifdef __cplusplus
}
#endif
#endif
