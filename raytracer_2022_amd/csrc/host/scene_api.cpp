// scene_api.cpp — host scene objects, BVH construction and flattening.
// See scene_api.hpp for the mapping to the reference's types.
#include "scene_api.hpp"

#include <iterator>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <fstream>

namespace rt2022 {

// ------------------------------------------------------------------ RNG ----
double HostRng::gen_range_inclusive(double low, double high) {
    // rand 0.8.5 UniformFloat::new_inclusive + Distribution::sample.
    const double max_rand = rtm::u2d((0xFFFFFFFFFFFFFFFFull >> 12) | 0x3FF0000000000000ull) - 1.0;
    double scale = (high - low) / max_rand;
    while (scale * max_rand + low > high) scale = rtm::u2d(rtm::d2u(scale) - 1);
    double value1_2 = rtm::u2d((next_u64() >> 12) | 0x3FF0000000000000ull);
    double value0_1 = value1_2 - 1.0;
    return value0_1 * scale + low;
}

Vec3 random_in_unit_xz_disk(HostRng &rng) {
    for (;;) {
        double x = rng.gen_range(-1.0, 1.0);
        double z = rng.gen_range(-1.0, 1.0);
        Vec3 p(x, 0.0, z);
        if (p.length() < 1.0) return p;
    }
}

// ---------------------------------------------------------------- AABB -----
AABB AABB::surrounding_box(const AABB &b0, const AABB &b1) {          // aabb.rs:34-46
    Point3 small(rtm::fmin_(b0.min.x, b1.min.x), rtm::fmin_(b0.min.y, b1.min.y), rtm::fmin_(b0.min.z, b1.min.z));
    Point3 large(rtm::fmax_(b0.max.x, b1.max.x), rtm::fmax_(b0.max.y, b1.max.y), rtm::fmax_(b0.max.z, b1.max.z));
    return AABB(small, large);
}

static void put3(double d[3], Vec3 v) { d[0] = v.x; d[1] = v.y; d[2] = v.z; }

// ------------------------------------------------------------- textures ----
uint32_t SolidColor::flatten(Flattener &f) const {
    rt_texture t{};
    t.kind = RT_TEX_SOLID;
    put3(t.color, color_value);
    f.textures.push_back(t);
    return (uint32_t)f.textures.size() - 1;
}

CheckerTexture::CheckerTexture(Color c1, Color c2)
    : odd(std::make_shared<SolidColor>(c1)), even(std::make_shared<SolidColor>(c2)) {}

uint32_t CheckerTexture::flatten(Flattener &f) const {
    uint32_t o = f.texture(odd), e = f.texture(even);
    rt_texture t{};
    t.kind = RT_TEX_CHECKER;
    t.a = o;
    t.b = e;
    f.textures.push_back(t);
    return (uint32_t)f.textures.size() - 1;
}

Perlin::Perlin(HostRng &rng) {                                          // perlin.rs:17-48
    for (int i = 0; i < 256; i++) {
        Vec3 v = rtm::to_unit(rng.random_vec(-1.0, 1.0));
        tab.randvec[i][0] = v.x; tab.randvec[i][1] = v.y; tab.randvec[i][2] = v.z;
    }
    int32_t *perms[3] = {tab.perm_x, tab.perm_y, tab.perm_z};
    for (int a = 0; a < 3; a++) {
        int32_t *p = perms[a];
        for (int i = 0; i < 256; i++) p[i] = i;
        for (int i = 255; i >= 0; i--) {                                 // permute(), perlin.rs:41-48
            uint64_t target = rng.gen_index((uint64_t)i + 1);
            std::swap(p[i], p[target]);
        }
    }
}

NoiseTexture::NoiseTexture(double scale_, HostRng &rng) : noise(std::make_shared<Perlin>(rng)), scale(scale_) {}

uint32_t NoiseTexture::flatten(Flattener &f) const {
    rt_texture t{};
    t.kind = RT_TEX_NOISE;
    t.a = f.perlin(noise);
    t.scale = scale;
    f.textures.push_back(t);
    return (uint32_t)f.textures.size() - 1;
}

ImageTexture::ImageTexture(uint32_t w, uint32_t h, const uint8_t *rgb_top_down) : width(w), height(h) {
    pixel_color.resize((size_t)w * h * 3);
    for (uint32_t y = 0; y < h; y++)                                     // mod.rs:94-99: row y of storage = image row h-1-y
        std::memcpy(&pixel_color[(size_t)y * w * 3], rgb_top_down + (size_t)(h - 1 - y) * w * 3, (size_t)w * 3);
}

static void ppm_skip(std::istream &in) {
    for (;;) {
        int c = in.peek();
        if (c == '#') { std::string line; std::getline(in, line); }
        else if (c == ' ' || c == '\n' || c == '\r' || c == '\t') in.get();
        else break;
    }
}

void load_image_file(const std::string &filename, uint32_t &w, uint32_t &h, std::vector<uint8_t> &top) {
    std::ifstream in(filename, std::ios::binary);
    if (!in) throw Error(RT_ERR_INVALID, "ImageTexture::new: cannot open " + filename);   // image::open(..).unwrap()
    int b0 = in.get(), b1 = in.peek();
    in.seekg(0);
    if (b0 == 0xFF && b1 == 0xD8) {                                      // JPEG: texture/mod.rs:90-93
        std::vector<uint8_t> bytes((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
        std::string err;
        if (!jpeg_decode_rgb8(bytes.data(), bytes.size(), w, h, top, err)) throw Error(RT_ERR_INVALID, "ImageTexture::new: " + filename + ": " + err);
        return;
    }
    std::string magic;
    in >> magic;
    if (magic != "P6") throw Error(RT_ERR_INVALID, "ImageTexture::new: " + filename + " is neither a JPEG nor a binary PPM (P6)");
    uint32_t maxv = 0;
    w = h = 0;
    ppm_skip(in); in >> w; ppm_skip(in); in >> h; ppm_skip(in); in >> maxv;
    in.get();
    if (!in || w == 0 || h == 0 || maxv != 255) throw Error(RT_ERR_INVALID, "ImageTexture::new: bad PPM header in " + filename);
    top.resize((size_t)w * h * 3);
    in.read((char *)top.data(), (std::streamsize)top.size());
    if ((size_t)in.gcount() != top.size()) throw Error(RT_ERR_INVALID, "ImageTexture::new: truncated PPM " + filename);
}

ImageTexture::ImageTexture(const std::string &filename) {
    uint32_t w = 0, h = 0;
    std::vector<uint8_t> top;
    load_image_file(filename, w, h, top);
    *this = ImageTexture(w, h, top.data());
}

uint32_t ImageTexture::flatten(Flattener &f) const {
    rt_image im{};
    im.width = width;
    im.height = height;
    im.offset = f.image_data.size();
    f.image_data.insert(f.image_data.end(), pixel_color.begin(), pixel_color.end());
    f.images.push_back(im);
    rt_texture t{};
    t.kind = RT_TEX_IMAGE;
    t.a = (uint32_t)f.images.size() - 1;
    f.textures.push_back(t);
    return (uint32_t)f.textures.size() - 1;
}

// ------------------------------------------------------------ materials ----
static uint32_t push_mat(Flattener &f, uint32_t kind, uint32_t tex, Color albedo, double param) {
    rt_material m{};
    m.kind = kind;
    m.tex = tex;
    put3(m.albedo, albedo);
    m.param = param;
    f.materials.push_back(m);
    return (uint32_t)f.materials.size() - 1;
}
uint32_t Lambertian::flatten(Flattener &f) const { return push_mat(f, RT_MAT_LAMBERTIAN, f.texture(albedo), Color(), 0.0); }
uint32_t Metal::flatten(Flattener &f) const { return push_mat(f, RT_MAT_METAL, 0, albedo, fuzz); }
uint32_t Dielectric::flatten(Flattener &f) const { return push_mat(f, RT_MAT_DIELECTRIC, 0, Color(), ir); }
uint32_t DiffuseLight::flatten(Flattener &f) const { return push_mat(f, RT_MAT_DIFFUSE_LIGHT, f.texture(emit), Color(), 0.0); }
uint32_t Isotropic::flatten(Flattener &f) const { return push_mat(f, RT_MAT_ISOTROPIC, f.texture(albedo), Color(), 0.0); }

// ------------------------------------------------------------ hittables ----
static uint32_t with_flip(uint32_t ref, bool flip) { return flip ? (ref | RT_REF_FLIP) : ref; }

std::optional<AABB> HittableList::bounding_box(double t0, double t1) const {     // mod.rs:102-120
    if (objects.empty()) return std::nullopt;
    AABB out(Point3(rtm::INF, rtm::INF, rtm::INF), Point3(-rtm::INF, -rtm::INF, -rtm::INF));
    for (auto &o : objects) {
        auto b = o->bounding_box(t0, t1);
        if (!b) return std::nullopt;
        out = AABB::surrounding_box(out, *b);
    }
    return out;
}
uint32_t HittableList::flatten(Flattener &f, bool flip) const {
    std::vector<uint32_t> items;
    for (auto &o : objects) items.push_back(f.hittable(o, flip));
    rt_list l{};
    l.first = (uint32_t)f.list_items.size();
    l.count = (uint32_t)items.size();
    f.list_items.insert(f.list_items.end(), items.begin(), items.end());
    f.lists.push_back(l);
    return RT_MAKE_REF(RT_KIND_LIST, f.lists.size() - 1);
}

std::optional<AABB> Sphere::bounding_box(double, double) const {                 // sphere.rs:68-73
    Vec3 r(radius, radius, radius);
    return AABB(center - r, center + r);
}
uint32_t Sphere::flatten(Flattener &f, bool flip) const {
    rt_sphere s{};
    put3(s.center, center);
    s.radius = radius;
    s.mat = f.material(mat_ptr);
    f.spheres.push_back(s);
    return with_flip(RT_MAKE_REF(RT_KIND_SPHERE, f.spheres.size() - 1), flip);
}

Point3 MovingSphere::center(double time) const {                                   // sphere.rs:124-127
    return center0 + (center1 - center0) * ((time - time0) / (time1 - time0));
}
std::optional<AABB> MovingSphere::bounding_box(double t0, double t1) const {     // sphere.rs:167-177
    Vec3 r(radius, radius, radius);
    AABB b0(center(t0) - r, center(t0) + r), b1(center(t1) - r, center(t1) + r);
    return AABB::surrounding_box(b0, b1);
}
uint32_t MovingSphere::flatten(Flattener &f, bool flip) const {
    rt_moving_sphere s{};
    put3(s.center0, center0);
    put3(s.center1, center1);
    s.time0 = time0; s.time1 = time1; s.radius = radius;
    s.mat = f.material(mat_ptr);
    f.moving_spheres.push_back(s);
    return with_flip(RT_MAKE_REF(RT_KIND_MOVING_SPHERE, f.moving_spheres.size() - 1), flip);
}

std::optional<AABB> Rect::bounding_box(double, double) const {                   // aarect.rs:40-45,123-128,206-211
    switch (axis) {
        case RT_RECT_XY: return AABB(Point3(a0, b0, k - 0.0001), Point3(a1, b1, k + 0.0001));
        case RT_RECT_XZ: return AABB(Point3(a0, k - 0.0001, b0), Point3(a1, k + 0.0001, b1));
        default:         return AABB(Point3(k - 0.0001, a0, b0), Point3(k + 0.0001, a1, b1));
    }
}
uint32_t Rect::flatten(Flattener &f, bool flip) const {
    rt_rect r{};
    r.a0 = a0; r.a1 = a1; r.b0 = b0; r.b1 = b1; r.k = k;
    r.axis = axis;
    r.mat = f.material(mp);
    f.rects.push_back(r);
    return with_flip(RT_MAKE_REF(RT_KIND_RECT, f.rects.size() - 1), flip);
}

std::optional<AABB> Boxes::bounding_box(double, double) const { return AABB(min, max); }   // boxes.rs:77-79
uint32_t Boxes::flatten(Flattener &f, bool flip) const {
    rt_box b{};
    put3(b.p0, min);
    put3(b.p1, max);
    b.mat = f.material(ptr);
    f.boxes.push_back(b);
    return with_flip(RT_MAKE_REF(RT_KIND_BOX, f.boxes.size() - 1), flip);
}

std::optional<AABB> Triangle::bounding_box(double, double) const {               // triangle.rs:79-92
    return AABB(Point3(rtm::fmin_(a.x, rtm::fmin_(b.x, c.x)), rtm::fmin_(a.y, rtm::fmin_(b.y, c.y)), rtm::fmin_(a.z, rtm::fmin_(b.z, c.z))),
                Point3(rtm::fmax_(a.x, rtm::fmax_(b.x, c.x)), rtm::fmax_(a.y, rtm::fmax_(b.y, c.y)), rtm::fmax_(a.z, rtm::fmax_(b.z, c.z))));
}
uint32_t Triangle::flatten(Flattener &f, bool flip) const {
    rt_triangle t{};
    put3(t.a, a); put3(t.b, b); put3(t.c, c);
    t.mat = f.material(mp);
    f.triangles.push_back(t);
    return with_flip(RT_MAKE_REF(RT_KIND_TRIANGLE, f.triangles.size() - 1), flip);
}

std::optional<AABB> Ring::bounding_box(double, double) const {                   // ring.rs:55-62
    double thickness = 0.0001;
    double rr = r + t;
    return AABB(Point3(-rr, -thickness, -rr), Point3(rr, thickness, rr));
}
uint32_t Ring::flatten(Flattener &f, bool flip) const {
    rt_ring g{};
    g.r = r; g.t = t; g.dis_min = dis_min; g.dis_max = dis_max;
    g.mat = f.material(mat);
    f.rings.push_back(g);
    return with_flip(RT_MAKE_REF(RT_KIND_RING, f.rings.size() - 1), flip);
}

std::optional<AABB> ConstantMedium::bounding_box(double t0, double t1) const { return boundary->bounding_box(t0, t1); }
uint32_t ConstantMedium::flatten(Flattener &f, bool flip) const {
    rt_medium m{};
    m.boundary = f.hittable(boundary, false);
    m.mat = f.material(phase_function);
    m.neg_inv_density = neg_inv_density;
    f.media.push_back(m);
    return with_flip(RT_MAKE_REF(RT_KIND_MEDIUM, f.media.size() - 1), flip);
}

std::optional<AABB> Translate::bounding_box(double t0, double t1) const {        // mod.rs:154-163
    auto b = ptr->bounding_box(t0, t1);
    if (!b) return std::nullopt;
    return AABB(b->min + offset, b->max + offset);
}
static uint32_t push_xform(Flattener &f, uint32_t kind, uint32_t child, double p0, double p1, double p2, bool flip) {
    rt_xform x{};
    x.kind = kind;
    x.child = child;
    x.p[0] = p0; x.p[1] = p1; x.p[2] = p2;
    f.xforms.push_back(x);
    return with_flip(RT_MAKE_REF(kind, f.xforms.size() - 1), flip);
}
uint32_t Translate::flatten(Flattener &f, bool flip) const {
    return push_xform(f, RT_KIND_TRANSLATE, f.hittable(ptr, false), offset.x, offset.y, offset.z, flip);
}

RotateY::RotateY(HittablePtr p, double angle) : ptr(std::move(p)) {               // mod.rs:188-228
    double radians = angle * (rtm::PI / 180.0);                                    // f64::to_radians
    sin_theta = rtm::sin_(radians);
    cos_theta = rtm::cos_(radians);
    auto ob = ptr->bounding_box(0.0, 1.0);
    if (!ob) { aabbox = std::nullopt; return; }
    Point3 mn(rtm::INF, rtm::INF, rtm::INF), mx(-rtm::INF, -rtm::INF, -rtm::INF);
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < 2; j++)
            for (int k = 0; k < 2; k++) {
                double x = (double)i * ob->max.x + (double)(1 - i) * ob->min.x;
                double y = (double)j * ob->max.y + (double)(1 - j) * ob->min.y;
                double z = (double)k * ob->max.z + (double)(1 - k) * ob->min.z;
                double newx = cos_theta * x + sin_theta * z;
                double newz = -sin_theta * x + cos_theta * z;
                Vec3 tester(newx, y, newz);
                for (int c = 0; c < 3; c++) {
                    mn.set(c, rtm::fmin_(mn[c], tester[c]));
                    mx.set(c, rtm::fmax_(mx[c], tester[c]));
                }
            }
    aabbox = AABB(mn, mx);
}
uint32_t RotateY::flatten(Flattener &f, bool flip) const {
    return push_xform(f, RT_KIND_ROTATE_Y, f.hittable(ptr, false), sin_theta, cos_theta, 0.0, flip);
}

std::optional<AABB> Zoom::bounding_box(double t0, double t1) const {             // mod.rs:310-319
    auto b = ptr->bounding_box(t0, t1);
    if (!b) return std::nullopt;
    return AABB(b->min * rate, b->max * rate);
}
uint32_t Zoom::flatten(Flattener &f, bool flip) const {
    return push_xform(f, RT_KIND_ZOOM, f.hittable(ptr, false), rate, 0.0, 0.0, flip);
}

uint32_t FlipFace::flatten(Flattener &f, bool flip) const { return f.hittable(ptr, !flip); }   // mod.rs:281-288

// ------------------------------------------------------------------ BVH ----
static int box_compare(const HittablePtr &a, const HittablePtr &b, int axis) {    // bvh/mod.rs:19-29
    auto ba = a->bounding_box(0.0, 0.0), bb = b->bounding_box(0.0, 0.0);
    if (!ba || !bb) throw Error(RT_ERR_INVALID, "BvhNode::box_compare: object without bounding box");
    double x = ba->min[axis], y = bb->min[axis];
    if (x < y) return -1;
    if (x > y) return 1;
    return 0;
}

std::shared_ptr<BvhNode> BvhNode::new_list(const HittableList &list, double time0, double time1, HostRng &rng) {
    return new_vec(list.objects, time0, time1, rng);
}

std::shared_ptr<BvhNode> BvhNode::new_vec(std::vector<HittablePtr> objects, double time0, double time1, HostRng &rng) {
    int axis = (int)rng.gen_index(3);                                               // bvh/mod.rs:35
    size_t span = objects.size();
    HittablePtr left, right;
    if (span == 0) throw Error(RT_ERR_INVALID, "BvhNode::new_vec: Get empty vec");
    if (span == 1) {
        left = right = objects.back();
    } else if (span == 2) {
        HittablePtr obj0 = objects.back(); objects.pop_back();                      // pop order, bvh/mod.rs:49-50
        HittablePtr obj1 = objects.back(); objects.pop_back();
        if (box_compare(obj0, obj1, axis) < 0) { left = obj0; right = obj1; }
        else { left = obj1; right = obj0; }
    } else {
        // sort_by is a stable sort; keys are cached (bounding_box(0,0).min[axis]).
        std::vector<std::pair<double, HittablePtr>> keyed;
        keyed.reserve(span);
        for (auto &o : objects) {
            auto b = o->bounding_box(0.0, 0.0);
            if (!b) throw Error(RT_ERR_INVALID, "BvhNode::box_compare: object without bounding box");
            keyed.emplace_back(b->min[axis], o);
        }
        std::stable_sort(keyed.begin(), keyed.end(), [](const auto &p, const auto &q) { return p.first < q.first; });
        size_t mid = span / 2;                                                      // split_off(span / 2)
        std::vector<HittablePtr> lv, rv;
        for (size_t i = 0; i < mid; i++) lv.push_back(keyed[i].second);
        for (size_t i = mid; i < span; i++) rv.push_back(keyed[i].second);
        left = new_vec(std::move(lv), time0, time1, rng);
        right = new_vec(std::move(rv), time0, time1, rng);
    }
    auto lb = left->bounding_box(time0, time1), rb = right->bounding_box(time0, time1);
    if (!lb || !rb) throw Error(RT_ERR_INVALID, "BvhNode::new_vec: No bounding box in bvh_node constructor.");
    auto node = std::make_shared<BvhNode>();
    node->aabbox = AABB::surrounding_box(*lb, *rb);
    node->left = left;
    node->right = right;
    return node;
}

uint32_t BvhNode::flatten(Flattener &f, bool flip) const {
    size_t idx = f.nodes.size();
    f.nodes.push_back(rt_bvh_node{});
    uint32_t l = f.hittable(left, flip);
    uint32_t r = (right.get() == left.get()) ? l : f.hittable(right, flip);
    rt_bvh_node n{};
    put3(n.bmin, aabbox.min);
    put3(n.bmax, aabbox.max);
    n.left = l;
    n.right = r;
    f.nodes[idx] = n;
    return RT_MAKE_REF(RT_KIND_NODE, idx);
}

// --------------------------------------------------------------- camera ----
Camera::Camera(Point3 lookfrom, Point3 lookat, Vec3 vup, double vfov, double aspect_ratio,
               double aperture, double focus_dist, double time0, double time1) {      // camera.rs:24-62
    double theta = vfov * (rtm::PI / 180.0);
    double half = theta / 2.0;
    double h = rtm::sin_(half) / rtm::cos_(half);                                     // tan
    double viewport_height = 2.0 * h;
    double viewport_width = aspect_ratio * viewport_height;
    Vec3 w = rtm::to_unit(lookfrom - lookat);
    Vec3 u = rtm::to_unit(rtm::cross(vup, w));
    Vec3 v = rtm::cross(w, u);
    Vec3 origin = lookfrom;
    Vec3 horizontal = u * viewport_width * focus_dist;
    Vec3 vertical = v * viewport_height * focus_dist;
    Vec3 llc = origin - horizontal / 2.0 - vertical / 2.0 - w * focus_dist;
    put3(c.origin, origin);
    put3(c.lower_left_corner, llc);
    put3(c.horizontal, horizontal);
    put3(c.vertical, vertical);
    put3(c.u, u); put3(c.v, v); put3(c.w, w);
    c.lens_radius = aperture / 2.0;
    c.time0 = time0;
    c.time1 = time1;
}

// ------------------------------------------------------------ flattening ----
uint32_t Flattener::material(const MaterialPtr &m) {
    auto it = mat_ids_.find(m.get());
    if (it != mat_ids_.end()) return it->second;
    uint32_t id = m->flatten(*this);
    mat_ids_[m.get()] = id;
    return id;
}
uint32_t Flattener::texture(const TexturePtr &t) {
    auto it = tex_ids_.find(t.get());
    if (it != tex_ids_.end()) return it->second;
    uint32_t id = t->flatten(*this);
    tex_ids_[t.get()] = id;
    return id;
}
uint32_t Flattener::perlin(const std::shared_ptr<const Perlin> &p) {
    auto it = perlin_ids_.find(p.get());
    if (it != perlin_ids_.end()) return it->second;
    perlins.push_back(p->tab);
    uint32_t id = (uint32_t)perlins.size() - 1;
    perlin_ids_[p.get()] = id;
    return id;
}
uint32_t Flattener::hittable(const HittablePtr &h, bool flip) {
    // Leaves / movers / media share one pool entry between flip states: the
    // flip lives in the ref. Nodes and lists are re-emitted with the flip
    // pushed down to their children.
    auto key = std::make_pair((const void *)h.get(), flip);
    auto it = hit_ids_.find(key);
    if (it != hit_ids_.end()) return it->second;
    auto other = hit_ids_.find(std::make_pair((const void *)h.get(), !flip));
    uint32_t ref;
    if (other != hit_ids_.end() && RT_REF_KIND(other->second) != RT_KIND_NODE && RT_REF_KIND(other->second) != RT_KIND_LIST)
        ref = other->second ^ RT_REF_FLIP;
    else
        ref = h->flatten(*this, flip);
    hit_ids_[key] = ref;
    return ref;
}
void Flattener::set_lights(const HittableList &lights_list) {
    lights.clear();
    for (auto &o : lights_list.objects) lights.push_back(hittable(o, false) & ~RT_REF_FLIP);
}

rt_scene_desc Flattener::desc() const {
    rt_scene_desc d{};
    d.abi_version = RT2022_ABI_VERSION;
    d.root = root;
#define RT_POOL(n, p, v) d.n = (uint32_t)v.size(); d.p = v.empty() ? nullptr : v.data()
    RT_POOL(n_nodes, nodes, nodes);
    RT_POOL(n_spheres, spheres, spheres);
    RT_POOL(n_moving_spheres, moving_spheres, moving_spheres);
    RT_POOL(n_rects, rects, rects);
    RT_POOL(n_boxes, boxes, boxes);
    RT_POOL(n_triangles, triangles, triangles);
    RT_POOL(n_rings, rings, rings);
    RT_POOL(n_media, media, media);
    RT_POOL(n_xforms, xforms, xforms);
    RT_POOL(n_lists, lists, lists);
    RT_POOL(n_list_items, list_items, list_items);
    RT_POOL(n_lights, lights, lights);
    RT_POOL(n_materials, materials, materials);
    RT_POOL(n_textures, textures, textures);
    RT_POOL(n_images, images, images);
    RT_POOL(n_perlins, perlins, perlins);
#undef RT_POOL
    d.image_data_bytes = image_data.size();
    d.image_data = image_data.empty() ? nullptr : image_data.data();
    return d;
}

std::vector<uint32_t> shuffled_rows(uint32_t image_height, HostRng &rng) {          // main.rs:93-99
    std::vector<uint32_t> id(image_height);
    for (uint32_t i = 0; i < image_height; i++) {
        id[i] = i;
        uint32_t target = rng.gen_index_u32(i + 1);
        std::swap(id[i], id[target]);
    }
    return id;
}

} // namespace rt2022
