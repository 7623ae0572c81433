/*
 * rt2022.h — C ABI of the MI355X path-tracing hot loop.
 *
 * Drop-in boundary for the per-pixel loop of Jerx2y/Raytracer-2022:
 * the body of the render-thread closure, raytracer/src/main.rs:133-159
 * (for y in rows { for x { for s { get_ray; ray_color } } } → Vec<Color>),
 * and the tone map that consumes it, main.rs:280-299.
 *
 * The reference has no FFI of its own: what crosses the seam
 * there is `cam: Camera`, `world: BvhNode`, `lights: HittableList`,
 * `background`, the shuffled row list and the image/sample constants, and what
 * comes back is one `Vec<Color>` of un-normalised f64 RGB sums per worker.
 * Here the same things cross as plain pointers and sizes:
 *
 *   reference (Rust)                          this ABI
 *   ----------------------------------------  ---------------------------------
 *   BvhNode / Arc<dyn Hittable> object graph  rt_scene_desc (flattened pools)
 *   Camera (basic/camera.rs:9-20)             rt_camera (the same 10 fields)
 *   consts + background + line_id slice       rt_params
 *   Vec<Color> sent over mpsc (main.rs:157)   double *out_rgb_sum (caller-owned)
 *   write_color (main.rs:280-299)             rt_write_color / rt_tonemap_device
 *   panic!/unwrap                             negative int + rt_last_error()
 *
 * Ownership: the caller owns every host buffer for the duration of the call
 * only; the library owns device memory behind rt_scene; output buffers are
 * caller-allocated. Threading: rt_render* are re-entrant on one rt_scene for
 * disjoint row sets (one host thread / process per GPU).
 *
 * All arithmetic is f64 like the reference. Integer outputs (pixel indices,
 * u8 colours, counters) are bit-exact against the CPU oracle in oracle/.
 */
#ifndef RT2022_H
#define RT2022_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT2022_ABI_VERSION 3

/* ---------------------------------------------------------------- refs --- */
/* A hittable reference = one `Arc<dyn Hittable>` of the reference, as a tagged
 * 32-bit id:  bit 31 = FlipFace applied to this object (hittable/mod.rs:267-292),
 * bits 30..27 = kind, bits 26..0 = index into that kind's pool. */
#define RT_REF_FLIP      0x80000000u
#define RT_REF_KIND_SHIFT 27
#define RT_REF_INDEX_MASK 0x07FFFFFFu
#define RT_MAKE_REF(kind, index) ((((uint32_t)(kind)) << RT_REF_KIND_SHIFT) | ((uint32_t)(index) & RT_REF_INDEX_MASK))
#define RT_REF_KIND(ref)  ((((uint32_t)(ref)) >> RT_REF_KIND_SHIFT) & 0xFu)
#define RT_REF_INDEX(ref) (((uint32_t)(ref)) & RT_REF_INDEX_MASK)

enum rt_kind {
    RT_KIND_NODE          = 0,  /* BvhNode            hittable/bvh/mod.rs:12-16   */
    RT_KIND_SPHERE        = 1,  /* Sphere<M>          hittable/sphere.rs:12-19    */
    RT_KIND_MOVING_SPHERE = 2,  /* MovingSphere<M>    hittable/sphere.rs:93-103   */
    RT_KIND_RECT          = 3,  /* XY/XZ/YZRect<M>    hittable/aarect.rs          */
    RT_KIND_BOX           = 4,  /* Boxes              hittable/boxes.rs:12-16     */
    RT_KIND_TRIANGLE      = 5,  /* Triangle<M>        hittable/triangle.rs:11-19  */
    RT_KIND_RING          = 6,  /* Ring<M>            hittable/ring.rs:11-20      */
    RT_KIND_MEDIUM        = 7,  /* ConstantMedium<H,T> hittable/constantmedium.rs */
    RT_KIND_TRANSLATE     = 8,  /* Translate<H>       hittable/mod.rs:135-175     */
    RT_KIND_ROTATE_Y      = 9,  /* RotateY<H>         hittable/mod.rs:177-265     */
    RT_KIND_ZOOM          = 10, /* Zoom<H>            hittable/mod.rs:294-331     */
    RT_KIND_LIST          = 11, /* HittableList       hittable/mod.rs:70-133      */
    RT_KIND_COUNT         = 12
};

/* ------------------------------------------------------------- geometry --- */
/* BvhNode flattened: aabbox + two child refs (NODE or object). 64 B. */
typedef struct rt_bvh_node {
    double   bmin[3];
    double   bmax[3];
    uint32_t left;
    uint32_t right;
    uint32_t _pad[2];                 /* one node = one 64-byte line = four 16-byte loads */
} rt_bvh_node;

typedef struct rt_sphere {            /* 40 B */
    double   center[3];
    double   radius;
    uint32_t mat;
    uint32_t _pad;
} rt_sphere;

typedef struct rt_moving_sphere {     /* 80 B */
    double   center0[3];
    double   center1[3];
    double   time0, time1;
    double   radius;
    uint32_t mat;
    uint32_t _pad;
} rt_moving_sphere;

enum rt_rect_axis { RT_RECT_XY = 0, RT_RECT_XZ = 1, RT_RECT_YZ = 2 };

/* XYRect{x0,x1,y0,y1,k} / XZRect{x0,x1,z0,z1,k} / YZRect{y0,y1,z0,z1,k}. 48 B. */
typedef struct rt_rect {
    double   a0, a1, b0, b1, k;
    uint32_t axis;
    uint32_t mat;
} rt_rect;

/* Boxes::new(p0,p1,m): six rects in the order of boxes.rs:24-66. 56 B. */
typedef struct rt_box {
    double   p0[3];
    double   p1[3];
    uint32_t mat;
    uint32_t _pad;
} rt_box;

typedef struct rt_triangle {          /* 80 B */
    double   a[3], b[3], c[3];
    uint32_t mat;
    uint32_t _pad;
} rt_triangle;

typedef struct rt_ring {              /* 40 B */
    double   r, t;
    double   dis_min, dis_max;        /* (r-t)^2, (r+t)^2   ring.rs:29-30 */
    uint32_t mat;
    uint32_t _pad;
} rt_ring;

/* ConstantMedium{boundary, phase_function: Isotropic, neg_inv_density}. 16 B.
 * `mat` must be an RT_MAT_ISOTROPIC material. The boundary may be any hittable
 * that contains no further medium (the reference's are a Sphere and
 * Translate<RotateY<Boxes>>, scene.rs:230-254, 316-329); a medium nested inside a
 * boundary is refused with RT_ERR_UNSUPPORTED. */
typedef struct rt_medium {
    uint32_t boundary;
    uint32_t mat;
    double   neg_inv_density;
} rt_medium;

/* Translate{offset} / RotateY{sin,cos} / Zoom{rate} around `child`. 32 B.
 * p = offset | {sin_theta, cos_theta, 0} | {rate, 0, 0}. */
typedef struct rt_xform {
    uint32_t kind;                    /* RT_KIND_TRANSLATE / ROTATE_Y / ZOOM */
    uint32_t child;
    double   p[3];
} rt_xform;

/* HittableList as an object: children = list_items[first .. first+count). */
typedef struct rt_list {
    uint32_t first;
    uint32_t count;
} rt_list;

#define RT_MAX_XFORM_DEPTH 4

/* ------------------------------------------------------------ shading ---- */
enum rt_material_kind {
    RT_MAT_LAMBERTIAN    = 0,  /* material/mod.rs:27-66   tex = albedo          */
    RT_MAT_METAL         = 1,  /* material/mod.rs:68-97   albedo, param = fuzz  */
    RT_MAT_DIELECTRIC    = 2,  /* material/mod.rs:99-148  param = ir            */
    RT_MAT_DIFFUSE_LIGHT = 3,  /* material/mod.rs:150-181 tex = emit            */
    RT_MAT_ISOTROPIC     = 4   /* material/mod.rs:183-214 tex = albedo          */
};

typedef struct rt_material {          /* 40 B */
    uint32_t kind;
    uint32_t tex;
    double   albedo[3];
    double   param;
} rt_material;

enum rt_texture_kind {
    RT_TEX_SOLID   = 0,        /* texture/mod.rs:14-29   color                  */
    RT_TEX_CHECKER = 1,        /* texture/mod.rs:31-60   a = odd tex, b = even  */
    RT_TEX_NOISE   = 2,        /* texture/mod.rs:62-79   a = perlin id, scale   */
    RT_TEX_IMAGE   = 3         /* texture/mod.rs:81-139  a = image id           */
};

typedef struct rt_texture {           /* 48 B */
    uint32_t kind;
    uint32_t a;
    uint32_t b;
    uint32_t _pad;
    double   color[3];
    double   scale;
} rt_texture;

/* ImageTexture: RGB8, rows stored bottom-up exactly as texture/mod.rs:94-99
 * builds `pixel_color`; texel (i,j) at image_data[offset + 3*(j*width+i)]. */
typedef struct rt_image {
    uint32_t width, height;
    uint64_t offset;
} rt_image;

/* Perlin tables, texture/perlin.rs:8-13. */
typedef struct rt_perlin {
    double  randvec[256][3];
    int32_t perm_x[256];
    int32_t perm_y[256];
    int32_t perm_z[256];
} rt_perlin;

/* ---------------------------------------------------------- the scene ---- */
typedef struct rt_scene_desc {
    uint32_t abi_version;             /* RT2022_ABI_VERSION */
    uint32_t root;                    /* world (main.rs:90), normally a NODE ref */

    uint32_t n_nodes;          const rt_bvh_node      *nodes;
    uint32_t n_spheres;        const rt_sphere        *spheres;
    uint32_t n_moving_spheres; const rt_moving_sphere *moving_spheres;
    uint32_t n_rects;          const rt_rect          *rects;
    uint32_t n_boxes;          const rt_box           *boxes;
    uint32_t n_triangles;      const rt_triangle      *triangles;
    uint32_t n_rings;          const rt_ring          *rings;
    uint32_t n_media;          const rt_medium        *media;
    uint32_t n_xforms;         const rt_xform         *xforms;
    uint32_t n_lists;          const rt_list          *lists;
    uint32_t n_list_items;     const uint32_t         *list_items;

    /* `lights: HittableList` (main.rs:89,120). n_lights == 0 selects the
     * documented cosine-only mode (the reference would panic, hittable/mod.rs:130). */
    uint32_t n_lights;         const uint32_t         *lights;

    uint32_t n_materials;      const rt_material      *materials;
    uint32_t n_textures;       const rt_texture       *textures;
    uint32_t n_images;         const rt_image         *images;
    uint64_t image_data_bytes; const uint8_t          *image_data;
    uint32_t n_perlins;        const rt_perlin        *perlins;
} rt_scene_desc;

/* Camera, the ten fields of basic/camera.rs:9-20 (built on the host by
 * Camera::new, camera.rs:24-62; consumed by get_ray, camera.rs:64-73). */
typedef struct rt_camera {
    double origin[3];
    double lower_left_corner[3];
    double horizontal[3];
    double vertical[3];
    double u[3], v[3], w[3];
    double lens_radius;
    double time0, time1;
} rt_camera;

/* Per-call constants of main.rs:33-51 + the row slice of main.rs:111-116. */
typedef struct rt_params {
    uint32_t width, height;           /* IMAGE_WIDTH / IMAGE_HEIGHT            */
    uint32_t spp;                     /* SAMPLES_PER_PIXEL                     */
    uint32_t max_depth;               /* MAX_DEPTH                             */
    double   background[3];
    double   t_min;                   /* 0.001 in main.rs:243                  */
    uint64_t seed;                    /* build-defined: the reference is unseeded */
    /* Rows to render, in output order (the reference's shuffled line ids).
     * A row id g addresses frame g / height, image row y = g % height
     * (y up, like main.rs:142); n_frames frames share scene and camera and
     * differ only in their RNG key. */
    uint32_t n_frames;                /* >= 1                                  */
    uint32_t n_rows;
    const uint32_t *row_ids;          /* host pointer (rt_render) or device pointer (rt_render_device) */
    /* 0: each pixel's samples are summed 0..spp in order (main.rs:144-151).
     * k>0: samples are summed in consecutive chunks of k, and the chunk sums
     * are then added in chunk order — same value to ~1 ulp, finer work items.
     * k=1 is the reference's order again, bit for bit (0 + L0 + L1 + ...), with
     * the finest work items: what a throughput-minded caller should pass. */
    uint32_t spp_chunk;
    uint32_t flags;                   /* RT_FLAG_* */
    /* Per-worker progress (the indicatif bars of main.rs:102-127,154-155: one per render thread, advanced as its rows
     * finish). NULL = none. Called on the thread that called rt_render* (rt_render_multi: on the device's own host
     * thread, concurrently with the other devices' — `worker` is the device's place in the set, 0 otherwise), between
     * passes — every few milliseconds of device time — with the camera paths started so far and their total
     * (n_rows * width * spp), and once more with done == total when the call's last pass has completed. It must
     * not call back into the library on the same scene; it never affects results. */
    void (*progress_cb)(void *user, uint32_t worker, uint64_t paths_done, uint64_t paths_total);
    void *progress_user;
} rt_params;

#define RT_FLAG_COUNTERS 0x1u         /* fill the counter fields of rt_stats */
#define RT_FLAG_KERNEL_TIMES 0x2u     /* fill rt_stats.trace_ms / shade_ms (HIP events around every pass of the default engine) */
#define RT_FLAG_ASYNC 0x4u            /* rt_render_device only: return as soon as the call is accepted; a host thread of the library
                                         drives the passes, rt_render_wait(scene, stream) joins it (see rt_render_device) */

typedef struct rt_stats {
    uint64_t paths;                               /* camera rays                         */
    uint64_t rays;                                /* world.hit calls from ray_color (main.rs:243) */
    uint64_t node_visits;                         /* AABB::hit calls                     */
    uint64_t prim_tests[RT_KIND_COUNT];           /* Hittable::hit calls by kind (leaf + boundary) */
    uint64_t light_pdf_tests;                     /* pdf_value re-intersections          */
    uint64_t rng_draws;                           /* 64-bit words drawn on the path      */
    double   ms;                                  /* device (or CPU) time of the call    */
    /* How the call was carried out (always filled; speed only, never results): */
    uint32_t spp_chunk;                           /* samples per work item actually used (params->spp_chunk, 0 resolved to spp) */
    uint32_t passes;                              /* shade + trace pass pairs of the wavefront engine (0: megakernel, CPU) */
    uint64_t pool_slots;                          /* path slots of the wavefront pool the call ran with */
    double   trace_ms, shade_ms;                  /* RT_FLAG_KERNEL_TIMES: device time summed over the call's traversal (wf_trace) and
                                                     shading (wf_shade) launches; 0 otherwise */
    uint64_t partial_bytes;                       /* HBM the call's per-sample partial sums took (spp_chunk > 0): 24 B x pixels x
                                                     ceil(spp / spp_chunk), or 24 B x pixels x the planes of the ring the library
                                                     switches to when that would exceed 40 % of the device's memory (same sums, bit for bit) */
} rt_stats;

typedef struct rt_scene rt_scene;     /* opaque: device-resident scene */

/* ------------------------------------------------------- entry points ---- */
/* Error codes (negative). */
#define RT_OK               0
#define RT_ERR_INVALID     -1   /* bad argument / malformed scene              */
#define RT_ERR_UNSUPPORTED -2   /* scene shape the device path does not cover  */
#define RT_ERR_DEVICE      -3   /* HIP runtime error (no GPU, OOM, launch)     */
#define RT_ERR_NOMEM       -4

/* Validates and copies the flattened scene into HBM on the current HIP device. */
int rt_scene_create(const rt_scene_desc *desc, rt_scene **out);
int rt_scene_destroy(rt_scene *scene);

/* Renders params->n_rows rows; out_rgb_sum (host) receives n_rows*width*3
 * un-normalised f64 sums in row_ids order — the Vec<Color> of main.rs:137-157.
 * Synchronous. stats may be NULL. */
int rt_render(rt_scene *scene, const rt_camera *cam, const rt_params *params,
              double *out_rgb_sum, rt_stats *stats);

/* Same, with params->row_ids and d_out_rgb_sum resident in HBM and every launch
 * issued on `hip_stream` (a hipStream_t; NULL = default stream), so it orders with
 * the caller's other work on that stream. The default engine drives its passes
 * from the host and polls a completion word, so the call returns when the frame's
 * last pass has been issued and observed (it synchronises `hip_stream`); `stats`,
 * if given, is filled by the next rt_render_wait on the same stream. One host
 * thread per (scene, stream). The device-resident row ids are checked like
 * rt_render's host ones (one small reduction kernel): an id >= height * n_frames
 * is RT_ERR_INVALID, not a silently mis-keyed frame.
 * The scene lives on the HIP device that was current at rt_scene_create; the call
 * makes that device current for its duration and restores the caller's, and
 * `hip_stream`, the row ids and the output must belong to that device.
 * RT_FLAG_ASYNC (a flag bit older callers never set: the ABI version stays 3): the call validates its arguments, copies `cam` and `params` and returns; the passes are
 * driven by a host thread of the library (one per call in flight), so the caller's thread is free — to tonemap or gather
 * the previous frame, or to start a frame on another stream of the same scene, whose passes then fill the chip while
 * this one's last rays drain. The device buffers (row ids, output) and `stats` stay the caller's until
 * rt_render_wait(scene, hip_stream), which joins the thread, returns the call's error if it had one (engine errors
 * surface there, not here) and fills `stats`; the next rt_render_device on the same (scene, stream) and
 * rt_scene_destroy wait likewise. params->progress_cb, if any, is called on the library's thread. Results are the
 * synchronous call's, bit for bit. */
int rt_render_device(rt_scene *scene, const rt_camera *cam, const rt_params *params,
                     double *d_out_rgb_sum, void *hip_stream, rt_stats *stats);
int rt_render_wait(rt_scene *scene, void *hip_stream);

/* ---- one call, several GPUs ------------------------------------------------------------------
 * The reference's main() drives all its workers from one place (main.rs:109-183: spawn, join, concatenate).
 * Two ways to do the same over GPUs:
 *   - one process per GPU (what bench.py and torch.distributed do): each process creates its own rt_scene and
 *     renders its share of the rows (film.py deals them); nothing below is needed;
 *   - ONE process, ONE call: an rt_scene_set holds a copy of the scene on every device whose bit is set in
 *     device_mask (bit d = HIP device d); rt_render_multi deals params->row_ids cyclically over those devices
 *     (entry i goes to the (i mod n)-th of them — with the reference's shuffled row list every device gets the same
 *     mix of cheap and dear rows), renders the shares concurrently, one host thread and one stream per device, and
 *     writes out_rgb_sum in row_ids order exactly as rt_render would. Results do not depend on the mask: the RNG is
 *     keyed per pixel and sample. stats, if given, carries the counters summed over the devices and the longest
 *     device time. */
typedef struct rt_scene_set rt_scene_set;
int rt_scene_set_create(const rt_scene_desc *desc, uint64_t device_mask, rt_scene_set **out);
int rt_scene_set_destroy(rt_scene_set *set);
int rt_render_multi(rt_scene_set *set, const rt_camera *cam, const rt_params *params, double *out_rgb_sum, rt_stats *stats);

/* write_color (main.rs:280-299): NaN→0, sqrt(c/spp), clamp [0,0.999], *255.999, floor. */
void rt_write_color(const double rgb_sum[3], int32_t spp, uint8_t out_rgb[3]);
/* Device form over n_pixels sums → n_pixels*3 bytes, on hip_stream. */
int rt_tonemap_device(const double *d_rgb_sum, uint64_t n_pixels, int32_t spp,
                      uint8_t *d_rgb8, void *hip_stream);

/* Last error message of the calling thread ("" if none). */
const char *rt_last_error(void);
/* ABI version the library was built with. */
uint32_t rt_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* RT2022_H */
