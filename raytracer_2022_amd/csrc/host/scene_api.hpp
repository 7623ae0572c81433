// scene_api.hpp — host-side mirror of the reference's scene-builder surface
// (SURVEY.md §8b "API surface to keep"): the same type names, constructor
// argument order and meaning as raytracer/src/{basic,hittable,material,texture},
// in C++ because no Rust toolchain exists in this pipeline. The objects only
// describe the scene; `Flattener` turns the `Arc<dyn Hittable>` graph into the
// POD pools of include/rt2022.h that the HIP kernels (and the oracle) read.
//
//   Rust                                       here
//   Arc<dyn Hittable> / Arc::new(x)            HittablePtr / make<T>(...)
//   HittableList::add                          HittableList::add
//   BvhNode::new_list(&list, t0, t1)           BvhNode::new_list(list, t0, t1, rng)
//   rand::thread_rng()                         an explicit, seeded HostRng
//   panic!/unwrap                              rt2022::Error (mapped to RT_ERR_* at the C ABI)
#ifndef RT2022_SCENE_API_HPP
#define RT2022_SCENE_API_HPP

#include <cstdint>
#include <map>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../../include/rt2022.h"
#include "../rt_math.h"

namespace rt2022 {

using rtm::Vec3;
using Point3 = Vec3;
using Color = Vec3;

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};

// ------------------------------------------------------------------ RNG ----
// Stand-in for rand::thread_rng() in scene / BVH construction: one seeded
// stream consumed in the reference's order (scene.rs, bvh/mod.rs:35, perlin.rs).
struct HostRng : rtm::Rng {
    explicit HostRng(uint64_t seed) : rtm::Rng(rtm::mix64(seed ^ 0x5CE4E5B9BF58476Dull)) {}
    // gen_range(low..=high) on f64: UniformFloat::new_inclusive + sample (rand 0.8.5).
    double gen_range_inclusive(double low, double high);
    Vec3 random_vec(double l, double r) {            // Vec3::random, vec.rs:48-55
        double x = gen_range(l, r), y = gen_range(l, r), z = gen_range(l, r);
        return Vec3(x, y, z);
    }
};
Vec3 random_in_unit_xz_disk(HostRng &rng);           // vec.rs:98-106

// ---------------------------------------------------------------- AABB -----
struct AABB {                                        // hittable/bvh/aabb.rs:5-46
    Point3 min, max;
    AABB() {}
    AABB(Point3 a, Point3 b) : min(a), max(b) {}
    static AABB surrounding_box(const AABB &b0, const AABB &b1);
};

class Flattener;

// ------------------------------------------------------------- textures ----
struct Texture {                                     // texture/mod.rs:10-12
    virtual ~Texture() {}
    virtual uint32_t flatten(Flattener &f) const = 0;
};
using TexturePtr = std::shared_ptr<const Texture>;

struct SolidColor : Texture {                        // texture/mod.rs:14-29
    Color color_value;
    explicit SolidColor(Color c) : color_value(c) {}
    uint32_t flatten(Flattener &f) const override;
};

struct CheckerTexture : Texture {                    // texture/mod.rs:31-60
    TexturePtr odd, even;
    CheckerTexture(Color c1, Color c2);              // CheckerTexture::new(c1, c2)
    CheckerTexture(TexturePtr o, TexturePtr e) : odd(std::move(o)), even(std::move(e)) {}
    uint32_t flatten(Flattener &f) const override;
};

struct Perlin {                                      // texture/perlin.rs:8-48
    rt_perlin tab;
    explicit Perlin(HostRng &rng);                   // Perlin::new()
};

struct NoiseTexture : Texture {                      // texture/mod.rs:62-79
    std::shared_ptr<const Perlin> noise;
    double scale;
    NoiseTexture(double scale_, HostRng &rng);       // NoiseTexture::new(scale)
    uint32_t flatten(Flattener &f) const override;
};

struct ImageTexture : Texture {                      // texture/mod.rs:81-139
    uint32_t width = 0, height = 0;
    std::vector<uint8_t> pixel_color;                // RGB8, bottom-up rows (mod.rs:94-99)
    // ImageTexture::new(filename), texture/mod.rs:89-107: a baseline JPEG (decoded by host/jpeg.cpp — the
    // reference's `image::open`) or a binary PPM (P6, maxval 255).
    explicit ImageTexture(const std::string &filename);
    // Top-down RGB8 rows as an image decoder yields them; flipped here like mod.rs:94-99.
    ImageTexture(uint32_t w, uint32_t h, const uint8_t *rgb_top_down);
    uint32_t flatten(Flattener &f) const override;
};

// ------------------------------------------------------------ materials ----
struct Material {                                    // material/mod.rs:15-25
    virtual ~Material() {}
    virtual uint32_t flatten(Flattener &f) const = 0;
};
using MaterialPtr = std::shared_ptr<const Material>;

struct Lambertian : Material {                       // material/mod.rs:27-66
    TexturePtr albedo;
    explicit Lambertian(Color a) : albedo(std::make_shared<SolidColor>(a)) {}          // Lambertian::new
    explicit Lambertian(TexturePtr t) : albedo(std::move(t)) {}                        // Lambertian::new_arc
    uint32_t flatten(Flattener &f) const override;
};
struct Metal : Material {                            // material/mod.rs:68-97
    Color albedo;
    double fuzz;
    Metal(Color a, double f) : albedo(a), fuzz(f < 1.0 ? f : 1.0) {}
    uint32_t flatten(Flattener &f) const override;
};
struct Dielectric : Material {                       // material/mod.rs:99-148
    double ir;
    explicit Dielectric(double index_of_refraction) : ir(index_of_refraction) {}
    uint32_t flatten(Flattener &f) const override;
};
struct DiffuseLight : Material {                     // material/mod.rs:150-181
    TexturePtr emit;
    explicit DiffuseLight(Color c) : emit(std::make_shared<SolidColor>(c)) {}
    explicit DiffuseLight(TexturePtr t) : emit(std::move(t)) {}
    uint32_t flatten(Flattener &f) const override;
};
struct Isotropic : Material {                        // material/mod.rs:183-214
    TexturePtr albedo;
    explicit Isotropic(Color c) : albedo(std::make_shared<SolidColor>(c)) {}
    explicit Isotropic(TexturePtr t) : albedo(std::move(t)) {}
    uint32_t flatten(Flattener &f) const override;
};

// ------------------------------------------------------------ hittables ----
struct Hittable {                                    // hittable/mod.rs:59-68
    virtual ~Hittable() {}
    virtual std::optional<AABB> bounding_box(double time0, double time1) const = 0;
    // Emits this object into the pools and returns its ref; `flip` = an odd
    // number of FlipFace wrappers whose effect reaches this object unchanged.
    virtual uint32_t flatten(Flattener &f, bool flip) const = 0;
};
using HittablePtr = std::shared_ptr<const Hittable>;

template <class T, class... A>
std::shared_ptr<T> make(A &&...a) { return std::make_shared<T>(std::forward<A>(a)...); }

struct HittableList : Hittable {                     // hittable/mod.rs:70-133
    std::vector<HittablePtr> objects;
    void add(HittablePtr object) { objects.push_back(std::move(object)); }
    std::optional<AABB> bounding_box(double t0, double t1) const override;
    uint32_t flatten(Flattener &f, bool flip) const override;
};

struct Sphere : Hittable {                           // hittable/sphere.rs:12-91
    Point3 center; double radius; MaterialPtr mat_ptr;
    Sphere(Point3 c, double r, MaterialPtr m) : center(c), radius(r), mat_ptr(std::move(m)) {}
    std::optional<AABB> bounding_box(double, double) const override;
    uint32_t flatten(Flattener &f, bool flip) const override;
};

struct MovingSphere : Hittable {                     // hittable/sphere.rs:93-178
    Point3 center0, center1; double time0, time1, radius; MaterialPtr mat_ptr;
    MovingSphere(Point3 c0, Point3 c1, double t0, double t1, double r, MaterialPtr m)
        : center0(c0), center1(c1), time0(t0), time1(t1), radius(r), mat_ptr(std::move(m)) {}
    Point3 center(double time) const;
    std::optional<AABB> bounding_box(double t0, double t1) const override;
    uint32_t flatten(Flattener &f, bool flip) const override;
};

struct Rect : Hittable {                             // hittable/aarect.rs
    uint32_t axis; double a0, a1, b0, b1, k; MaterialPtr mp;
    Rect(uint32_t ax, double a0_, double a1_, double b0_, double b1_, double k_, MaterialPtr m)
        : axis(ax), a0(a0_), a1(a1_), b0(b0_), b1(b1_), k(k_), mp(std::move(m)) {}
    std::optional<AABB> bounding_box(double, double) const override;
    uint32_t flatten(Flattener &f, bool flip) const override;
};
struct XYRect : Rect {                               // XYRect::new(x0,x1,y0,y1,k,mp)
    XYRect(double x0, double x1, double y0, double y1, double k, MaterialPtr m) : Rect(RT_RECT_XY, x0, x1, y0, y1, k, std::move(m)) {}
};
struct XZRect : Rect {                               // XZRect::new(x0,x1,z0,z1,k,mp)
    XZRect(double x0, double x1, double z0, double z1, double k, MaterialPtr m) : Rect(RT_RECT_XZ, x0, x1, z0, z1, k, std::move(m)) {}
};
struct YZRect : Rect {                               // YZRect::new(y0,y1,z0,z1,k,mp)
    YZRect(double y0, double y1, double z0, double z1, double k, MaterialPtr m) : Rect(RT_RECT_YZ, y0, y1, z0, z1, k, std::move(m)) {}
};

struct Boxes : Hittable {                            // hittable/boxes.rs:12-83
    Point3 min, max; MaterialPtr ptr;
    Boxes(Point3 p0, Point3 p1, MaterialPtr m) : min(p0), max(p1), ptr(std::move(m)) {}
    std::optional<AABB> bounding_box(double, double) const override;
    uint32_t flatten(Flattener &f, bool flip) const override;
};

struct Triangle : Hittable {                         // hittable/triangle.rs:11-93
    Point3 a, b, c; MaterialPtr mp;
    Triangle(Point3 x, Point3 y, Point3 z, MaterialPtr m) : a(x), b(y), c(z), mp(std::move(m)) {}
    std::optional<AABB> bounding_box(double, double) const override;
    uint32_t flatten(Flattener &f, bool flip) const override;
};

struct Ring : Hittable {                             // hittable/ring.rs:11-63
    double r, t; MaterialPtr mat; double dis_min, dis_max;
    Ring(double r_, double t_, MaterialPtr m) : r(r_), t(t_), mat(std::move(m)), dis_min((r_ - t_) * (r_ - t_)), dis_max((r_ + t_) * (r_ + t_)) {}
    std::optional<AABB> bounding_box(double, double) const override;
    uint32_t flatten(Flattener &f, bool flip) const override;
};

struct ConstantMedium : Hittable {                   // hittable/constantmedium.rs:14-84
    HittablePtr boundary; std::shared_ptr<const Isotropic> phase_function; double neg_inv_density;
    ConstantMedium(HittablePtr b, double d, Color c)                                   // ConstantMedium::new
        : boundary(std::move(b)), phase_function(std::make_shared<Isotropic>(c)), neg_inv_density(-1.0 / d) {}
    ConstantMedium(HittablePtr b, double d, TexturePtr a)                              // ConstantMedium::new_arc
        : boundary(std::move(b)), phase_function(std::make_shared<Isotropic>(std::move(a))), neg_inv_density(-1.0 / d) {}
    std::optional<AABB> bounding_box(double t0, double t1) const override;
    uint32_t flatten(Flattener &f, bool flip) const override;
};

struct Translate : Hittable {                        // hittable/mod.rs:135-175
    HittablePtr ptr; Vec3 offset;
    Translate(HittablePtr p, Vec3 displacement) : ptr(std::move(p)), offset(displacement) {}
    std::optional<AABB> bounding_box(double t0, double t1) const override;
    uint32_t flatten(Flattener &f, bool flip) const override;
};

struct RotateY : Hittable {                          // hittable/mod.rs:177-265
    HittablePtr ptr; double sin_theta, cos_theta; std::optional<AABB> aabbox;
    RotateY(HittablePtr p, double angle_degrees);
    std::optional<AABB> bounding_box(double, double) const override { return aabbox; }
    uint32_t flatten(Flattener &f, bool flip) const override;
};

struct Zoom : Hittable {                             // hittable/mod.rs:294-331
    double rate; HittablePtr ptr;
    Zoom(HittablePtr p, double rate_) : rate(rate_), ptr(std::move(p)) {}
    std::optional<AABB> bounding_box(double t0, double t1) const override;
    uint32_t flatten(Flattener &f, bool flip) const override;
};

struct FlipFace : Hittable {                         // hittable/mod.rs:267-292
    HittablePtr ptr;
    explicit FlipFace(HittablePtr p) : ptr(std::move(p)) {}
    std::optional<AABB> bounding_box(double t0, double t1) const override { return ptr->bounding_box(t0, t1); }
    uint32_t flatten(Flattener &f, bool flip) const override;
};

struct BvhNode : Hittable {                          // hittable/bvh/mod.rs:11-105
    AABB aabbox; HittablePtr left, right;
    // BvhNode::new_list / new_vec. The split axis comes from `rng` (bvh/mod.rs:35).
    static std::shared_ptr<BvhNode> new_list(const HittableList &list, double time0, double time1, HostRng &rng);
    static std::shared_ptr<BvhNode> new_vec(std::vector<HittablePtr> objects, double time0, double time1, HostRng &rng);
    std::optional<AABB> bounding_box(double, double) const override { return aabbox; }
    uint32_t flatten(Flattener &f, bool flip) const override;
};

// --------------------------------------------------------------- camera ----
struct Camera {                                      // basic/camera.rs:8-62
    rt_camera c;
    Camera(Point3 lookfrom, Point3 lookat, Vec3 vup, double vfov, double aspect_ratio,
           double aperture, double focus_dist, double time0, double time1);
};

// ------------------------------------------------------------ flattening ----
// Owns the pools behind an rt_scene_desc.
class Flattener {
  public:
    std::vector<rt_bvh_node> nodes;
    std::vector<rt_sphere> spheres;
    std::vector<rt_moving_sphere> moving_spheres;
    std::vector<rt_rect> rects;
    std::vector<rt_box> boxes;
    std::vector<rt_triangle> triangles;
    std::vector<rt_ring> rings;
    std::vector<rt_medium> media;
    std::vector<rt_xform> xforms;
    std::vector<rt_list> lists;
    std::vector<uint32_t> list_items;
    std::vector<uint32_t> lights;
    std::vector<rt_material> materials;
    std::vector<rt_texture> textures;
    std::vector<rt_image> images;
    std::vector<uint8_t> image_data;
    std::vector<rt_perlin> perlins;
    uint32_t root = 0;

    uint32_t material(const MaterialPtr &m);
    uint32_t texture(const TexturePtr &t);
    uint32_t perlin(const std::shared_ptr<const Perlin> &p);
    // Memoised Hittable::flatten keyed by (object, flip).
    uint32_t hittable(const HittablePtr &h, bool flip);

    // world → root, lights → light refs (FlipFace on a light is ignored like the
    // reference: pdf_value/random are not forwarded by wrappers, mod.rs:62-67).
    void set_world(const HittablePtr &world) { root = hittable(world, false); }
    void set_lights(const HittableList &lights_list);
    rt_scene_desc desc() const;

  private:
    std::map<const void *, uint32_t> mat_ids_, tex_ids_, perlin_ids_;
    std::map<std::pair<const void *, bool>, uint32_t> hit_ids_;
};

// ------------------------------------------------------------- scenes ------
// scene.rs:22-571. Fns returning only a world in the reference return an empty
// light list here, except where a build decision is noted.
struct SceneOut {
    HittableList world;
    HittableList lights;
};
struct SceneAssets {            // where ImageTexture::new finds its files; empty → procedural stand-in
    std::string dir;
};
SceneOut random_scene(HostRng &rng);                                  // scene.rs:22-84
SceneOut random_scene_n(HostRng &rng, int half_grid);                 // same rule on a (2k+1)^2 grid (scaling runs)
SceneOut two_spheres(HostRng &rng);                                   // scene.rs:87-105
SceneOut two_perlin_spheres(HostRng &rng);                            // scene.rs:108-124
SceneOut earth(HostRng &rng, const SceneAssets &assets);              // scene.rs:127-140
SceneOut simple_light(HostRng &rng);                                  // scene.rs:143-162
SceneOut cornell_box(HostRng &rng);                                   // scene.rs:165-196
SceneOut cornell_smoke(HostRng &rng);                                 // scene.rs:199-257
SceneOut final_scene(HostRng &rng, const SceneAssets &assets);        // scene.rs:260-362
SceneOut wwscene(HostRng &rng, const SceneAssets &assets, int shuttle_subdiv);   // scene.rs:468-571

// Procedural RGB8 stand-in (top-down rows) used when an asset file is absent.
std::vector<uint8_t> procedural_planet_rgb8(uint32_t w, uint32_t h, uint32_t variant);

// host/jpeg.cpp: baseline JPEG -> RGB8 (rows top-down), and RGB8 -> baseline JPEG at an IJG quality (main.rs:213-221 uses 100).
bool jpeg_decode_rgb8(const uint8_t *data, size_t size, uint32_t &width, uint32_t &height, std::vector<uint8_t> &rgb, std::string &err);
bool jpeg_encode_rgb8(const uint8_t *rgb, uint32_t width, uint32_t height, int quality, std::vector<uint8_t> &out);
// The decode step of ImageTexture::new: JPEG or binary PPM file -> RGB8 rows, top-down. Throws Error.
void load_image_file(const std::string &filename, uint32_t &width, uint32_t &height, std::vector<uint8_t> &rgb_top_down);

// main.rs:93-99: Fisher-Yates shuffle of the row ids.
std::vector<uint32_t> shuffled_rows(uint32_t image_height, HostRng &rng);

} // namespace rt2022
#endif
