/* rtx_hip.h — C ABI of the MI355X (gfx950) wavefront path-tracing backend.
 *
 * This is the drop-in boundary for ONE path of abusch/rustracer: the call
 *     renderer::render(scene, integrator, camera, num_threads, sampler, 16)
 * made once per WorldEnd (rustracer-core/src/api.rs:1003-1010, implemented in
 * rustracer-core/src/renderer.rs:22-143). A host (Rust in the reference; the C++ layer of
 * include/rtx_host.h here) keeps scene parsing, the SAH BVH build and the Film; it flattens
 * BVH nodes, triangles, material/texture tables and the light list into the plain arrays below
 * and hands them over. Plain pointers and sizes only; nothing here knows about torch.
 *
 * `rc/` = rustracer-core/src/ of the reference.
 */
#ifndef RTX_HIP_H
#define RTX_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_OK 0
#define RT_ERR_INVALID (-1)   /* bad argument / inconsistent description            */
#define RT_ERR_NO_DEVICE (-2) /* no usable gfx950 device; there is NO CPU fallback  */
#define RT_ERR_HIP (-3)       /* a HIP call failed; see rt_last_error()             */
#define RT_ERR_OOM (-4)
#define RT_ERR_UNSUPPORTED (-5) /* input the reference accepts and this backend does not  */

/* -- flattened BVH node, 32 B: replaces LinearBVHNode (rc/bvh/mod.rs:582-598) ------------ */
typedef struct rt_bvh_node {
  float bmin[3];
  float bmax[3];
  uint32_t offset;  /* leaf: first primitive (leaf order); interior: second child index */
  uint16_t n_prims; /* 0 => interior                                                    */
  uint8_t axis;     /* interior: split axis                                             */
  uint8_t pad;
} rt_bvh_node;

/* -- per-triangle metadata: replaces GeometricPrimitive{shape,material,area_light}
 *    (rc/primitive.rs:34-76) and Triangle{reverse_orientation,swaps_handedness}
 *    (rc/shapes/mesh.rs:175-180) ---------------------------------------------------------- */
#define RT_TRI_FLIP 1u   /* reverse_orientation ^ transform_swaps_handedness */
#define RT_TRI_HAS_N 2u  /* mesh.n is Some                                   */
#define RT_TRI_HAS_UV 4u /* mesh.uv is Some (else default uvs, mesh.rs:201-211) */
#define RT_TRI_HAS_S 8u  /* mesh.s is Some                                   */
#define RT_TRI_HAS_ALPHA 16u        /* mesh.alpha_mask is Some: tri_alpha[2 i] is its float texture (rc/shapes/mesh.rs:38,134-144)          */
#define RT_TRI_HAS_SHADOW_ALPHA 32u /* mesh.shadow_alpha_mask is Some: tri_alpha[2 i + 1] (mesh.rs:39,146-156); shadow rays only (:577-581) */
#define RT_PRIM_SPHERE 64u /* this primitive is an analytic sphere (rc/shapes/sphere.rs), not a triangle: its tri_p slot holds the world bounding
                              box (p0 = min, p1 = max) and, as the bits of p2.x, its index into rt_scene_desc::spheres                          */
#define RT_PRIM_INSTANCE 128u /* this primitive is an object instance (TransformedPrimitive, rc/primitive.rs:79-118): its tri_p slot holds the world
                                 bounding box (p0 = min, p1 = max) and, as the bits of p2.x, its index into rt_scene_desc::instances              */
typedef struct rt_tri_meta {
  int32_t material; /* index into materials[]                 */
  int32_t light;    /* index into lights[] (area light) or -1 */
  uint32_t flags;
  uint32_t source_index; /* triangle index before BVH re-ordering (diagnostics) */
} rt_tri_meta;

/* -- Shape "sphere": replaces Sphere (rc/shapes/sphere.rs:15-68). o2w / w2o: the object-to-world matrix and its inverse, row-major 4x4
 *    (Transform{m, m_inv}: the inverse as the host computed it - a Gauss-Jordan inverse in f32 need not have an exact last row, and
 *    Transform * Point divides by w whenever w != 1, transform.rs:264-286); z_min .. phi_max as Sphere::new leaves them. -------------------- */
typedef struct rt_sphere {
  float o2w[16], w2o[16];
  float radius, z_min, z_max, theta_min, theta_max, phi_max;
  int32_t reverse_orientation, swaps_handedness;
  /* the same record carries the reference's two other quadrics: kind 1 = Disk (rc/shapes/disk.rs: height, radius, inner_radius, phi_max),
   * kind 2 = Cylinder (rc/shapes/cylinder.rs: radius, z_min, z_max, phi_max); kind 0 = Sphere */
  int32_t kind;
  float height, inner_radius;
} rt_sphere;

/* -- ObjectInstance: replaces TransformedPrimitive{primitive, primitive_to_world} (rc/primitive.rs:79-118, rc/api.rs:1053-1090). The object is what
 *    object_instance wraps: the BVH aggregate over the object's primitives - n_nodes nodes from nodes[node_base], child / primitive offsets relative to
 *    the object's own first node / first primitive - or, for an object of exactly one primitive, that primitive itself (n_nodes = 0). Its n_prims
 *    primitives (triangles in OBJECT space, leaf order) sit in the tri_* arrays from prim_base on, after the n_top_prims primitives of the top level.
 *    A hit inside instance k carries the id n_top_prims + (n_prims of instances 0 .. k-1) + its leaf-order index in the object (rt_trace_closest):
 *    ids below n_top_prims name top-level primitives, as before. ------------------------------------------------------------------------------ */
typedef struct rt_instance {
  float o2w[16], w2o[16]; /* primitive_to_world and its inverse as the host holds them, row-major */
  uint32_t node_base, n_nodes, prim_base, n_prims;
} rt_instance;

/* -- textures: replaces dyn Texture<T> (rc/texture/{constant,scale,mix,imagemap,checkerboard,uv,fbm}.rs) --
 * checkerboard (2D): tex1, tex2, mapping, amount = AAMethod (0 none, 1 closedform); uv: mapping;
 * fbm: value[0] = omega, amount = octaves (texture space = world space). */
enum { RT_TEX_CONST = 0, RT_TEX_SCALE = 1, RT_TEX_MIX = 2, RT_TEX_IMAGE = 3, RT_TEX_CHECKER = 4, RT_TEX_UV = 5, RT_TEX_FBM = 6 };
typedef struct rt_texture {
  int32_t kind;
  float value[3];             /* constant; float textures use value[0]                 */
  int32_t tex1, tex2, amount; /* scale / mix / checkerboard operands (see above)       */
  int32_t image;              /* imagemap: index into images[]                         */
  float mapping[4];           /* UVMapping2D su sv du dv (rc/texture/mod.rs:38-61)      */
} rt_texture;

/* -- MIP pyramid built by the host: replaces MIPMap<Spectrum> (rc/mipmap.rs:46-53) --------- */
enum { RT_WRAP_REPEAT = 0, RT_WRAP_BLACK = 1, RT_WRAP_CLAMP = 2 };
#define RT_MAX_MIP_LEVELS 16
typedef struct rt_image {
  int32_t n_levels;
  int32_t width[RT_MAX_MIP_LEVELS], height[RT_MAX_MIP_LEVELS];
  uint64_t offset[RT_MAX_MIP_LEVELS]; /* texel offset of each level inside `texels` (RGB f32) */
  const float* texels;                /* all levels, row-major, 3 floats per texel            */
  uint64_t n_texels;
  int32_t trilinear;
  float max_anisotropy;
  int32_t wrap;
} rt_image;

/* -- materials: replaces dyn Material (rc/material/*.rs) ----------------------------------- */
enum { RT_MAT_MATTE = 0, RT_MAT_PLASTIC, RT_MAT_METAL, RT_MAT_MIRROR, RT_MAT_GLASS, RT_MAT_UBER, RT_MAT_SUBSTRATE, RT_MAT_MIX, RT_MAT_TRANSLUCENT,
       RT_MAT_DISNEY /* rc/material/disney.rs; slots: KD color, KS metallic, ETA eta, ROUGHNESS roughness, KR speculartint, UROUGH anisotropic,
                        KT sheen, SIGMA sheentint, VROUGH clearcoat, K clearcoatgloss, OPACITY spectrans, REFLECT scatterdistance,
                        TRANSMIT flatness, AMOUNT difftrans, M1 = thin (0 / 1, not an id) */ };
enum { RT_SLOT_KD = 0, RT_SLOT_KS, RT_SLOT_KR, RT_SLOT_KT, RT_SLOT_SIGMA, RT_SLOT_ROUGHNESS, RT_SLOT_UROUGH, RT_SLOT_VROUGH,
       RT_SLOT_ETA, RT_SLOT_K, RT_SLOT_OPACITY, RT_SLOT_REFLECT, RT_SLOT_TRANSMIT, RT_SLOT_AMOUNT, RT_SLOT_M1, RT_SLOT_M2, RT_N_SLOTS };
typedef struct rt_material {
  int32_t kind;
  int32_t slot[RT_N_SLOTS]; /* texture ids (material ids for M1/M2); -1 = absent */
  int32_t remap_roughness;
  int32_t bump;             /* "bumpmap" float texture id or -1 (material::bump, rc/material/mod.rs:50-92); ignored by mix */
} rt_material;

/* -- lights: replaces dyn Light (rc/light/{diffuse,point,distant,infinite}.rs) ------------- */
enum { RT_LIGHT_DIFFUSE_AREA = 0, RT_LIGHT_POINT = 1, RT_LIGHT_DISTANT = 2, RT_LIGHT_INFINITE = 3 };
typedef struct rt_light {
  int32_t kind;
  int32_t prim;      /* area: emitting triangle, LEAF-ORDER index          */
  float rgb[3];      /* area: L_emit; point: I; distant: L                 */
  int32_t two_sided;
  float vec[3];      /* point: position; distant: normalised direction     */
  float area;        /* area: Shape::area()                                */
  float world_radius; /* distant / infinite: scene bounding-sphere radius  */
  int32_t image;     /* infinite: Lmap pyramid                             */
  float l2w[12], w2l[12]; /* infinite: 3x4 light<->world                   */
  /* infinite: Distribution2D over (2*w) x (2*h) (rc/light/infinite.rs:80-101), host-built */
  int32_t dist_nu, dist_nv;
  const float* dist_func;     /* nv*nu                */
  const float* dist_cdf;      /* nv*(nu+1)            */
  const float* dist_func_int; /* nv                   */
  const float* marg_func;     /* nv                   */
  const float* marg_cdf;      /* nv+1                 */
  float marg_func_int;
} rt_light;

/* -- the flattened scene handed over at WorldEnd -------------------------------------------- */
typedef struct rt_scene_desc {
  uint32_t n_nodes;
  const rt_bvh_node* nodes; /* pre-order, left child = i+1 (rc/bvh/mod.rs:314-358)          */
  uint32_t n_tris;          /* primitives in leaf order: triangles, and spheres where tri_meta[i].flags has RT_PRIM_SPHERE      */
  const float* tri_p;       /* n_tris*9, world space, LEAF ORDER (p0 p1 p2)                  */
  const float* tri_n;       /* n_tris*9 or NULL                                              */
  const float* tri_uv;      /* n_tris*6 or NULL                                              */
  const float* tri_s;       /* n_tris*9 or NULL                                              */
  const rt_tri_meta* tri_meta;
  const int32_t* tri_alpha; /* n_tris*2 {alpha, shadowalpha} float-texture ids, read where the RT_TRI_HAS_*ALPHA flags are set; NULL if no
                               mesh carries a mask. A hit whose mask evaluates to 0 is no hit (Triangle::intersect mesh.rs:353-370,
                               intersect_p :534-582) - in BVH traversal and in Shape::pdf_wi's re-intersection alike */
  uint32_t n_spheres; const rt_sphere* spheres; /* the analytic spheres among the n_tris primitives (RT_PRIM_SPHERE); an area light's `prim` may name one */
  uint32_t n_textures; const rt_texture* textures;
  uint32_t n_images; const rt_image* images;
  uint32_t n_materials; const rt_material* materials;
  uint32_t n_lights; const rt_light* lights; /* order = Scene::lights (rc/scene.rs:24)       */
  /* object instances (two-level traversal). With n_instances > 0: nodes[0 .. n_top_nodes) is the top-level tree and tri_*[0 .. n_top_prims) its
   * primitives (some of them RT_PRIM_INSTANCE); the objects' trees and primitives follow in the same arrays (n_nodes, n_tris count everything).
   * Without instances the three fields are 0 / NULL. */
  uint32_t n_instances; const rt_instance* instances;
  uint32_t n_top_nodes, n_top_prims;
  /* Emitters that are no lights: lights[n_lights .. n_lights + n_unlisted_lights) are DiffuseAreaLight records (rgb, two_sided) a primitive's `light`
   * may name although Scene::lights does not hold them - a shape with an AreaLightSource inside an ObjectBegin block keeps its area light (it glows when a
   * camera ray or a specular bounce hits it) while the light itself never reaches the scene's list (rc/api.rs:954-964): it is never sampled, has no entry in
   * the light distribution, and a BSDF-sampled ray that reaches it contributes nothing (no sampled light is this one, rc/integrator/mod.rs:293-307). */
  uint32_t n_unlisted_lights;
} rt_scene_desc;

/* -- what renderer::render receives through &dyn Camera / Film / Sampler / Integrator ------- */
typedef struct rt_camera {        /* PerspectiveCamera (rc/camera.rs:18-27)        */
  float raster_to_camera[16];     /* 4x4 projective                                */
  float camera_to_world[16];
  float dx_camera[3], dy_camera[3];
  float lens_radius, focal_distance;
} rt_camera;
typedef struct rt_film_desc {     /* Film (rc/film.rs:45-55)                       */
  int32_t cropped_pixel_bounds[4]; /* x0 y0 x1 y1                                  */
  int32_t sample_bounds[4];        /* Film::get_sample_bounds, x0 y0 x1 y1         */
  float filter_radius[2];
  float filter_table[256];         /* 16x16, film.rs:16-17,92-102                  */
  float max_sample_luminance;
} rt_film_desc;
typedef struct rt_sampler_desc {  /* ZeroTwoSequence (rc/sampler/zerotwosequence.rs) */
  int32_t spp;                    /* rounded up to a power of two by the callee    */
  int32_t dimensions;             /* "dimensions", default 4                       */
} rt_sampler_desc;
typedef struct rt_path_desc {     /* PathIntegrator (rc/integrator/path.rs:25-31)  */
  int32_t max_depth;
  float rr_threshold;
  int32_t light_strategy;         /* 0 "spatial" (voxel CDF, rc/lightdistrib.rs), 1 "uniform" */
  int32_t pixel_bounds[4];        /* x0 y0 x1 y1                                    */
} rt_path_desc;
/* Film sharding for multi-GPU: this call renders the bands r of the sample rows with r % world_size == rank; pixels of
 * other rows stay zero in the output. A band is RT_SHARD_ROWS(H, world_size) rows high: 4 rows on a sharded frame (round 6; rounds 4 - 5 cut the reference's 16-row
 * tile rows, or 8 rows where those did not divide over the ranks). What a rank's rows cost depends on what they show, and the slowest rank sets the frame time: on the
 * headline scene the 8-way split's slowest rank took 1.037 of the mean with 16-row bands, 1.022 with 8, 1.013 with 4 (profiles/r06_shard_band_sweep.txt), and 1080 rows
 * are 270 bands - 34 or 33 per rank of 8. Which samples a device renders changes nothing in the film: every pixel's sampler is keyed by the pixel (pixel-keyed mode). */
typedef struct rt_shard { int32_t rank, world_size; } rt_shard;
#define RT_SHARD_ROWS(H, world_size) ((world_size) > 1 ? 4 : 16)

typedef struct rt_stats {
  uint64_t camera_rays;
  uint64_t rays_closest, rays_shadow, rays_mis;  /* ray casts by class            */
  uint64_t nodes_closest, nodes_shadow, nodes_mis; /* BVH node visits (0 unless counting is on) */
  uint64_t tris_closest, tris_shadow, tris_mis;    /* triangle tests                */
  uint64_t paths_scrubbed;                       /* NaN / negative / inf samples set to black (renderer.rs:115-126) */
  double ms_total;                               /* rt_render wall time, device-synchronised */
  double ms_sampler, ms_raygen, ms_trace_closest, ms_trace_any, ms_trace_mis, ms_shade, ms_resolve, ms_film, ms_lightdist; /* HIP-event times */
  uint64_t launches_trace_closest;               /* number of trace_closest launches (path + MIS) */
  uint64_t n_passes;
  /* path vertices by the shade front-end that served them: constant-Kd matte under area lights, Lambert under any light,
   * two-lobe materials (matte with sigma, plastic, metal, mirror), everything else */
  uint64_t vertices_lambert_const, vertices_lambert, vertices_two_lobe, vertices_generic;
  /* ms_shade split (RT_FLAG_TIME_KERNELS): the shade launches of each front-end, the queue binning, the miss bin */
  double ms_shade_lambert_const, ms_shade_lambert, ms_shade_two_lobe, ms_shade_generic, ms_shade_bin, ms_shade_miss;
  /* measurement builds only (make ABLATE=1; zero otherwise): wave cycles per section of the shade kernel, [front-end 0..3][section 0..7] =
   * state + interaction, emission + differentials, material, light pick, light-sampling half, BSDF-sampling half, continuation + stores, loop tail */
  uint64_t shade_section_cycles[32];
  /* BSDF-sampled MIS rays toward an infinite light are traced for occlusion only (their own launch of the any-hit kernel): of rays_mis / nodes_mis /
   * tris_mis, the part that belongs to those rays, and of ms_trace_mis the time of their launches. nodes_ / tris_mis_any are filled by frames rendered with
   * RT_FLAG_COUNT_TRAVERSAL | RT_FLAG_COUNT_AS_RENDERED (plain RT_FLAG_COUNT_TRAVERSAL walks every MIS ray as the reference does: closest hit). */
  uint64_t rays_mis_any, nodes_mis_any, tris_mis_any;
  double ms_trace_mis_any;
  /* rt_multi_render, `total` only: wall time from the moment the last device finished its chunks to the merged frame being in place (the additions on
   * devices[0] and the final copy); per-device wall times are per_device[k].ms_total */
  double ms_gather;
  /* of rays_mis: BSDF-sampled rays toward a sphere light that miss the sphere's world box and were therefore not cast (Sphere::pdf_wi is non-zero for any
   * direction, sphere.rs:310-334; such a ray's term is zero whatever it hits). Zero on frames that count the reference's walk. */
  uint64_t rays_mis_not_cast;
  /* of rays_closest: path rays that follow a NON-specular bounce at the depth limit and were therefore not cast. PathIntegrator::li traces the next ray
   * before it tests the depth (path.rs:100-137) and reads the hit only to add emitted light after a specular bounce: after any other bounce the ray at
   * bounces == max_depth is never read. The film is the one the cast rays give; zero on frames that count the reference's walk. */
  uint64_t rays_tail_not_cast;
  /* launches per stage of this frame, as issued (round 5; launches_trace_closest above = launches_trace_path + launches_trace_mis): the closest-hit launches of
   * the path rays, the any-hit launches of the shadow rays, the closest-hit launches of the BSDF-sampled MIS rays, the any-hit launches of the MIS rays toward
   * an infinite light, and the k_shade launches of all front-ends together (binning and the miss bin not counted). What a per-launch average divides by. */
  uint64_t launches_trace_path, launches_trace_shadow, launches_trace_mis, launches_trace_mis_any, launches_shade;
} rt_stats;

#define RT_FLAG_COUNT_TRAVERSAL 1u /* fill nodes_ and tris_ counters (slower)                  */
#define RT_FLAG_FILM_ON_DEVICE 2u  /* film_xyzw is a device pointer (HBM-resident output)      */
#define RT_FLAG_TIME_KERNELS 4u    /* fill the per-kernel ms_ fields with HIP events           */
#define RT_FLAG_COUNT_AS_RENDERED 8u /* with RT_FLAG_COUNT_TRAVERSAL: count the walks an uncounted frame runs (occlusion-only MIS rays walk as
                                        intersect_p does) instead of the reference's (every MIS ray a closest-hit walk): what a roofline figure divides by
                                        the time of */
#define RT_FLAG_REF_STREAM 16u /* round 6: the frame with the REFERENCE'S sampler stream - one PCG32 stream per 16 x 16 tile, consumed by the tile's pixels and samples in order
                                (rc/renderer.rs:83-84) - instead of the pixel-keyed one: one lane per tile walks the reference's loop (slow by construction; for the configuration
                                the reference itself runs). The film then equals the oracle's SAMPLER_REF mode sample for sample: weights exact, radiance inside the image gate.
                                Single device, plain-triangle scenes; stats: camera_rays and the three ray counts. */

typedef struct rt_scene rt_scene;

/* Uploads the flattened scene (copies every array; the caller may free its own afterwards).
 * device < 0 => the current HIP device. Replaces Scene::new + BVH ownership (rc/scene.rs:29-49). */
int rt_scene_create(const rt_scene_desc* desc, int device, rt_scene** out);
void rt_scene_destroy(rt_scene* scene);

/* BVH::new (rc/bvh/mod.rs:80-135) on the device, for callers that want the tree in milliseconds rather than the best
 * tree: a linear BVH (63-bit Morton order of the centroids, one radix sort, boxes fitted bottom-up), leaves of up to
 * max_prims_per_node triangles, written in the layout flatten_bvh gives (rc/bvh/mod.rs:314-358: pre-order, first child
 * = i + 1, `offset` = second child, `axis` = the axis the children are ordered along). The SAH tree of the host layer
 * traces faster and is what parity on node / triangle counts refers to; closest hits do not depend on the builder.
 * tri_p: n_tris x 9 floats, world space, host memory. nodes: room for 2 * n_tris - 1 records. ordered: n_tris source
 * triangle indices in leaf order. RT_ERR_UNSUPPORTED if the tree needs more than the 64-entry traversal stack. */
int rt_bvh_build(const float* tri_p, uint32_t n_tris, int32_t max_prims_per_node, rt_bvh_node* nodes, uint32_t* n_nodes, int32_t* ordered,
                 float* ms_device /* may be NULL */);

/* MIPMap::new (rc/mipmap.rs:75-187) on the device: the Lanczos zoom of a non-power-of-two image (taps computed by the caller: first
 * source texel + 4 weights per output texel and axis; NULL and px = width, py = height for power-of-two images) and the box-filtered
 * levels. lvl_w / lvl_h / lvl_off: level geometry (texel offsets into texels_out, which receives every level). Host pointers. */
int rt_mip_build(const float* rgb, int32_t width, int32_t height, int32_t px, int32_t py, const int32_t* s_first, const float* s_wts, const int32_t* t_first,
                 const float* t_wts, int32_t wrap, int32_t n_levels, const int32_t* lvl_w, const int32_t* lvl_h, const uint64_t* lvl_off, float* texels_out);
/* The sampling tables of InfiniteAreaLight::new (rc/light/infinite.rs:78-101) on the device: func = luminance of the filtered map
 * lookup * sin(theta) at width x height (twice the map's resolution), one Distribution1D per row (rc/distribution1d.rs:11-42) and the
 * marginal one over the row integrals. mode / il / delta: MIPMap::lookup's level choice for the constant filter width (0: level 0,
 * 1: the last level's texel, 2: levels il and il + 1 blended by delta); sin_theta: height values. All pointers are host memory. */
int rt_env_distribution(const rt_image* image, int32_t width, int32_t height, int32_t mode, int32_t il, float delta, const float* sin_theta,
                        float* func, float* cdf, float* func_int, float* marg_cdf, float* marg_func_int);

/* Renders one frame: the body of renderer::render (rc/renderer.rs:22-143) — preprocess (light
 * distribution), per-pixel sampler tables, camera rays, PathIntegrator::li for every sample,
 * radiance scrubbing and Film::add_sample/merge. Blocking; calls on one rt_scene from several threads are safe and take turns (they share the
 * scene's workspace), calls on different rt_scene objects run concurrently. The light distribution is built by the first frame that needs it
 * and kept (the scene is immutable). film_xyzw: W*H*4 floats over the
 * cropped pixel bounds, (X, Y, Z, filter_weight_sum) per pixel as Film's Pixel (film.rs:38-43).
 * `stream` is a hipStream_t (NULL = the null stream). */
int rt_render(rt_scene* scene, const rt_camera* camera, const rt_film_desc* film, const rt_sampler_desc* sampler,
              const rt_path_desc* path, const rt_shard* shard, uint32_t flags, void* stream, float* film_xyzw, rt_stats* stats);

/* Several GPUs of one node from one process (north_star: "partition the film across the 8 GPUs of one node"). The reference's
 * render loop hands 16 x 16 tiles to worker threads from a shared queue and merges finished tiles into the film (rc/renderer.rs:47-71,
 * rc/film.rs:177-194); here the workers are GPUs. rt_multi_create replicates the scene on every listed device (a device may be listed more than
 * once). rt_multi_render cuts the frame into n_devices * chunks_per_device chunks of interleaved bands of RT_SHARD_ROWS rows, lets one host thread per
 * device pull chunks from a shared counter (chunks_per_device = 1: the static split; > 1: the dynamic queue for frames whose rows differ in
 * cost), and sends only the film rows a chunk can have touched to devices[0] over xGMI (hipMemcpyPeerAsync), where they are summed in chunk
 * order - the gather that replaces Film::merge_film_tile. No collective while paths are traced. film_xyzw: host memory, or memory of
 * devices[0] with RT_FLAG_FILM_ON_DEVICE. total / per_device (n_devices entries) may be NULL; total->ms_total is the wall time of the call.
 * One process per GPU (torch.distributed / RCCL harness) uses rt_render with rt_shard instead. */
typedef struct rt_multi rt_multi;
int rt_multi_create(const rt_scene_desc* desc, const int32_t* devices, int32_t n_devices, rt_multi** out);
void rt_multi_destroy(rt_multi* multi);
int rt_multi_render(rt_multi* multi, const rt_camera* camera, const rt_film_desc* film, const rt_sampler_desc* sampler, const rt_path_desc* path,
                    int32_t chunks_per_device, uint32_t flags, float* film_xyzw, rt_stats* total, rt_stats* per_device);

/* Kernel-level entry points used by the parity tests.
 * rays: n*8 floats (o.xyz, t_max, d.xyz, unused). closest: hits n*4 floats (t, prim as int bits
 * or -1, b0, b1) — BVH::intersect (rc/bvh/mod.rs:366-433). any: hits n uint32 0/1 —
 * BVH::intersect_p (:435-501). counters (optional): {node visits, triangle tests}; with counters the kernels
 * that walk the tree one node per step (the reference's visit sequence) run, without them the kernels rt_render
 * uses (same hits). Host pointers. */
int rt_trace_closest(rt_scene* scene, const float* rays, uint64_t n, float* hits, uint64_t counters[2]);
int rt_trace_any(rt_scene* scene, const float* rays, uint64_t n, uint32_t* occluded, uint64_t counters[2]);
/* Same kernels on device-resident buffers, timed with HIP events on `stream` (bench): returns the
 * average milliseconds per launch over `reps` launches in *ms_per_launch. */
int rt_trace_closest_device(rt_scene* scene, const void* d_rays, uint64_t n, void* d_hits, int reps, void* stream, float* ms_per_launch);

/* ZeroTwoSequence::start_pixel in the pixel-keyed mode: for pixels
 * [pixel0, pixel0+n_pixels) returns the 12 scramble words and the post-shuffle sample index
 * permutation of each of the 2*dimensions tables. scrambles: n_pixels*3*dimensions u32
 * (1D dims first, then 2D pairs); perms: n_pixels*2*dimensions*spp u16. Host pointers. */
int rt_sampler_tables(int32_t spp, int32_t dimensions, uint64_t pixel0, uint64_t n_pixels, uint32_t* scrambles, uint16_t* perms);
/* Same result from the single-kernel statement of the algorithm (one lane walks a pixel's whole RNG stream in
 * order). Not used by rt_render; it is the on-device cross-check of the segmented sampler over large pixel ranges. */
int rt_sampler_tables_plain(int32_t spp, int32_t dimensions, uint64_t pixel0, uint64_t n_pixels, uint32_t* scrambles, uint16_t* perms);

/* offset_ray_origin (rc/geometry/mod.rs:203-220) with its next_float_up / next_float_down steps (rc/lib.rs:227-262) on n points: p, p_error, normal and
 * direction w as n x 3 floats each, out n x 3. The device function every spawned ray goes through (Interaction::spawn_ray / spawn_ray_to, rc/interaction.rs:
 * 56-74), exposed for the parity tests (zeros of both signs, infinities, denormals, NaN). Host pointers. */
int rt_offset_ray_origin(const float* p, const float* p_error, const float* n, const float* w, uint64_t count, float* out);

/* Dense voxel light distribution of SpatialLightDistribution (rc/lightdistrib.rs:101-179):
 * n_voxels[3]; func: nvox*n_lights, cdf: nvox*(n_lights+1), func_int: nvox (host pointers, may be NULL
 * to query n_voxels only). */
int rt_light_distribution(rt_scene* scene, int32_t n_voxels[3], float* func, float* cdf, float* func_int);

/* What rt_scene_create decided about a scene (measurement / tests): RT_QUERY_LDS_RESIDENT - 1 if the traversal kernels keep the whole tree and its
 * primitives in LDS (k_trace; the roofline of such a scene's traversal is VALU issue, its HBM bytes are ray records only), else 0. < 0: bad argument. */
enum { RT_QUERY_LDS_RESIDENT = 0, RT_QUERY_LDS_NODES_TESTED = 1 /* LDS-resident scenes: the nodes the stackless walks test (<= n_nodes: interior nodes whose test rarely fails are passed over) */,
       RT_QUERY_LDS_OCCLUSION = 2 /* 1 if the scene is too large for RT_QUERY_LDS_RESIDENT but fits ONE workgroup's 160 KB per CU (<= 2816 nodes, <= 1408 plain triangles): occlusion rays walk an LDS copy of it, closest-hit rays its bounds and link tables;
                                      RT_QUERY_LDS_NODES_TESTED then counts the occlusion walk's nodes */ };
int rt_scene_query(rt_scene* scene, int32_t what);
/* The tables rt_scene_create hands the stackless LDS walks of a small (<= 256 nodes, <= 128 primitives) or mid-size (<= 2816 / 1408) scene, computed on the host alone - no
 * device is touched (tests, offline inspection). link_kept / link_full: 9 * n_nodes + 9 words each (rows 0 - 7: closest hit by direction octant, their 8 start nodes,
 * row 8: occlusion rays, its start node; a word = (first tested node inside the node's subtree << 16) | first tested node after it, a leaf's word = bit 31 | its primitive
 * range | the same low half) over the nodes the calibration kept / over all nodes. stats (27 doubles, may be NULL): per set of calibration rays 0 - 8 the number of rays, their
 * simulated node tests with every node tested, and with the kept ones. mid != 0: the mid-size packing of a leaf's primitive range. RT_ERR_INVALID when capacity_words is short. */
int rt_link_tables(const rt_scene_desc* desc, int32_t mid, uint32_t* link_kept, uint32_t* link_full, uint64_t capacity_words, double* stats);
/* sizeof() of an ABI struct by its C name ("rt_stats", "rt_scene_desc", ...), or -1: lets a binding in another language check its mirror of the
 * header against the library it actually loaded (rustracer_amd/host.py does at load time; tests/test_abi_cpu.py checks every struct). */
int rt_sizeof(const char* struct_name);

const char* rt_last_error(void);
/* 1 if a gfx950 device is visible to this process, else 0 (never falls back to a CPU path). */
int rt_device_available(void);
const char* rt_version(void);

#ifdef __cplusplus
}
#endif
#endif /* RTX_HIP_H */
