// host_cabi.cpp — extern "C" surface of the host scene layer (include/rt2022_host.h)
// and the host half of include/rt2022.h (rt_write_color, rt_last_error).
#include <cstdio>
#include <cstring>
#include <string>

#include "../../../include/rt2022_host.h"
#include "rt_error.hpp"
#include "scene_api.hpp"

using namespace rt2022;

namespace rt2022 {
thread_local std::string g_last_error;
void set_error(const std::string &m) { g_last_error = m; }
} // namespace rt2022

struct rtb_scene {
    Flattener flat;
    rt_scene_desc desc;
    std::string name;
};

namespace {
struct ViewSpec { double from[3], at[3], vup[3], vfov, aperture, focus, bg[3]; };

ViewSpec view_for(const std::string &name) {
    // v1 only carries the wwscene camera (main.rs:43-51); the
    // other scenes take the RTIOW book values the README names as their basis.
    if (name == "wwscene") return {{0, 15, -150}, {35, 0, 0}, {1, 5, 0}, 40, 0.0, 10, {0, 0, 0}};
    if (name == "cornell_box" || name == "cornell_smoke") return {{278, 278, -800}, {278, 278, 0}, {0, 1, 0}, 40, 0.0, 10, {0, 0, 0}};
    if (name == "final_scene") return {{478, 278, -600}, {278, 278, 0}, {0, 1, 0}, 40, 0.0, 10, {0, 0, 0}};
    if (name == "simple_light") return {{26, 3, 6}, {0, 2, 0}, {0, 1, 0}, 20, 0.0, 10, {0, 0, 0}};
    if (name == "random_scene") return {{13, 2, 3}, {0, 0, 0}, {0, 1, 0}, 20, 0.1, 10, {0.7, 0.8, 1.0}};
    return {{13, 2, 3}, {0, 0, 0}, {0, 1, 0}, 20, 0.0, 10, {0.7, 0.8, 1.0}};
}

template <class F>
int guarded(F &&f) {
    try {
        return f();
    } catch (const Error &e) {
        set_error(e.what());
        return e.code;
    } catch (const std::exception &e) {
        set_error(e.what());
        return RT_ERR_INVALID;
    }
}

// A leaf stand-in for rtb_bvh_build: fixed box, fixed ref.
struct BoxedLeaf : Hittable {
    AABB box; uint32_t ref;
    BoxedLeaf(AABB b, uint32_t r) : box(b), ref(r) {}
    std::optional<AABB> bounding_box(double, double) const override { return box; }
    uint32_t flatten(Flattener &, bool flip) const override { return flip ? (ref ^ RT_REF_FLIP) : ref; }
};
} // namespace

extern "C" {

const char *rt_last_error(void) { return g_last_error.c_str(); }
const char *rtb_last_error(void) { return g_last_error.c_str(); }
uint32_t rt_abi_version(void) { return RT2022_ABI_VERSION; }

void rt_write_color(const double rgb_sum[3], int32_t spp, uint8_t out_rgb[3]) {      // main.rs:280-299
    for (int i = 0; i < 3; i++) {
        double c = rgb_sum[i];
        if (c != c) c = 0.0;
        double v = rtm::floor_(rtm::clamp_(rtm::sqrt_(c / (double)spp), 0.0, 0.999) * 255.999);
        out_rgb[i] = (uint8_t)v;
    }
}

int rtb_scene_build(const char *name, uint64_t seed, const char *assets_dir, int32_t param, rtb_scene **out) {
    return guarded([&]() -> int {
        if (!name || !out) throw Error(RT_ERR_INVALID, "rtb_scene_build: null argument");
        std::string n(name);
        HostRng rng(seed);
        SceneAssets assets{assets_dir ? std::string(assets_dir) : std::string()};
        SceneOut s;
        if (n == "random_scene") s = param > 0 ? random_scene_n(rng, param) : random_scene(rng);
        else if (n == "two_spheres") s = two_spheres(rng);
        else if (n == "two_perlin_spheres") s = two_perlin_spheres(rng);
        else if (n == "earth") s = earth(rng, assets);
        else if (n == "simple_light") s = simple_light(rng);
        else if (n == "cornell_box") s = cornell_box(rng);
        else if (n == "cornell_smoke") s = cornell_smoke(rng);
        else if (n == "final_scene") s = final_scene(rng, assets);
        else if (n == "wwscene") s = wwscene(rng, assets, param);
        else throw Error(RT_ERR_INVALID, "rtb_scene_build: unknown scene '" + n + "'");
        auto world = BvhNode::new_list(s.world, 0.0, 1.0, rng);                       // main.rs:90 (time0 = 0, time1 = 1)
        auto sc = new rtb_scene();
        sc->name = n;
        sc->flat.set_world(world);
        sc->flat.set_lights(s.lights);
        sc->desc = sc->flat.desc();
        *out = sc;
        return RT_OK;
    });
}

void rtb_scene_free(rtb_scene *scene) { delete scene; }
const rt_scene_desc *rtb_scene_desc(const rtb_scene *scene) { return scene ? &scene->desc : nullptr; }

int rtb_scene_default_view(const rtb_scene *scene, double aspect_ratio, rt_camera *cam, double background[3]) {
    return guarded([&]() -> int {
        if (!scene || !cam) throw Error(RT_ERR_INVALID, "rtb_scene_default_view: null argument");
        ViewSpec v = view_for(scene->name);
        Camera c(Point3(v.from[0], v.from[1], v.from[2]), Point3(v.at[0], v.at[1], v.at[2]), Vec3(v.vup[0], v.vup[1], v.vup[2]),
                 v.vfov, aspect_ratio, v.aperture, v.focus, 0.0, 1.0);
        *cam = c.c;
        if (background) { background[0] = v.bg[0]; background[1] = v.bg[1]; background[2] = v.bg[2]; }
        return RT_OK;
    });
}

int rtb_camera_new(const double lookfrom[3], const double lookat[3], const double vup[3], double vfov,
                   double aspect_ratio, double aperture, double focus_dist, double time0, double time1,
                   rt_camera *out) {
    return guarded([&]() -> int {
        if (!lookfrom || !lookat || !vup || !out) throw Error(RT_ERR_INVALID, "rtb_camera_new: null argument");
        Camera c(Point3(lookfrom[0], lookfrom[1], lookfrom[2]), Point3(lookat[0], lookat[1], lookat[2]),
                 Vec3(vup[0], vup[1], vup[2]), vfov, aspect_ratio, aperture, focus_dist, time0, time1);
        *out = c.c;
        return RT_OK;
    });
}

int rtb_shuffled_rows(uint32_t image_height, uint64_t seed, uint32_t *out_rows) {
    return guarded([&]() -> int {
        if (!out_rows && image_height) throw Error(RT_ERR_INVALID, "rtb_shuffled_rows: null output");
        HostRng rng(seed);
        auto rows = shuffled_rows(image_height, rng);
        std::memcpy(out_rows, rows.data(), rows.size() * sizeof(uint32_t));
        return RT_OK;
    });
}

int rtb_bvh_build(const uint32_t *leaf_refs, const double *boxes6, uint32_t n, uint64_t seed,
                  rt_bvh_node *out_nodes, uint32_t max_nodes) {
    return guarded([&]() -> int {
        if (!leaf_refs || !boxes6 || !out_nodes) throw Error(RT_ERR_INVALID, "rtb_bvh_build: null argument");
        HittableList list;
        for (uint32_t i = 0; i < n; i++) {
            const double *b = boxes6 + 6 * (size_t)i;
            list.add(make<BoxedLeaf>(AABB(Point3(b[0], b[1], b[2]), Point3(b[3], b[4], b[5])), leaf_refs[i]));
        }
        HostRng rng(seed);
        auto root = BvhNode::new_list(list, 0.0, 1.0, rng);
        Flattener f;
        f.set_world(root);
        if (f.nodes.size() > max_nodes) throw Error(RT_ERR_INVALID, "rtb_bvh_build: output too small");
        std::memcpy(out_nodes, f.nodes.data(), f.nodes.size() * sizeof(rt_bvh_node));
        return (int)f.nodes.size();
    });
}

int rtb_fill_image(const double *rgb_sum, const uint32_t *row_ids, uint32_t n_rows, uint32_t width,
                   uint32_t height, int32_t spp, uint8_t *rgb8) {
    return guarded([&]() -> int {
        if (!rgb_sum || !row_ids || !rgb8) throw Error(RT_ERR_INVALID, "rtb_fill_image: null argument");
        for (uint32_t yi = 0; yi < n_rows; yi++) {
            uint32_t y = row_ids[yi];
            if (y >= height) throw Error(RT_ERR_INVALID, "rtb_fill_image: row id out of range");
            for (uint32_t x = 0; x < width; x++)                                       // img(x, H-1-y), main.rs:197
                rt_write_color(rgb_sum + ((size_t)yi * width + x) * 3, spp, rgb8 + ((size_t)(height - y - 1) * width + x) * 3);
        }
        return RT_OK;
    });
}

// sizeof of every ABI structure, in the order of _ffi.ABI_STRUCTS (layout check for bindings).
int rtb_abi_sizes(uint32_t *out, uint32_t n) {
    const uint32_t sizes[] = {sizeof(rt_bvh_node), sizeof(rt_sphere), sizeof(rt_moving_sphere), sizeof(rt_rect), sizeof(rt_box),
                              sizeof(rt_triangle), sizeof(rt_ring), sizeof(rt_medium), sizeof(rt_xform), sizeof(rt_list),
                              sizeof(rt_material), sizeof(rt_texture), sizeof(rt_image), sizeof(rt_perlin), sizeof(rt_scene_desc),
                              sizeof(rt_camera), sizeof(rt_params), sizeof(rt_stats)};
    const uint32_t count = sizeof(sizes) / sizeof(sizes[0]);
    for (uint32_t i = 0; i < n && i < count; i++) out[i] = sizes[i];
    return (int)count;
}

int rtb_write_ppm(const char *path, const uint8_t *rgb8, uint32_t width, uint32_t height) {
    return guarded([&]() -> int {
        FILE *f = path ? std::fopen(path, "wb") : nullptr;
        if (!f) throw Error(RT_ERR_INVALID, std::string("rtb_write_ppm: cannot create ") + (path ? path : "(null)"));
        std::fprintf(f, "P6\n%u %u\n255\n", width, height);
        std::fwrite(rgb8, 1, (size_t)width * height * 3, f);
        std::fclose(f);
        return RT_OK;
    });
}

int rtb_image_load(const char *path, uint32_t *width, uint32_t *height, uint8_t *out_rgb8, uint64_t capacity) {
    return guarded([&]() -> int {
        if (!path || !width || !height) throw Error(RT_ERR_INVALID, "rtb_image_load: null argument");
        std::vector<uint8_t> top;
        load_image_file(path, *width, *height, top);
        if (out_rgb8) {
            if (capacity < top.size()) throw Error(RT_ERR_INVALID, "rtb_image_load: output buffer too small");
            std::memcpy(out_rgb8, top.data(), top.size());
        }
        return RT_OK;
    });
}

int rtb_write_jpeg(const char *path, const uint8_t *rgb8, uint32_t width, uint32_t height, int32_t quality) {
    return guarded([&]() -> int {
        std::vector<uint8_t> bytes;
        if (!rgb8 || !jpeg_encode_rgb8(rgb8, width, height, quality, bytes)) throw Error(RT_ERR_INVALID, "rtb_write_jpeg: bad image");
        FILE *f = path ? std::fopen(path, "wb") : nullptr;
        if (!f) throw Error(RT_ERR_INVALID, std::string("rtb_write_jpeg: cannot create ") + (path ? path : "(null)"));   // File::create(..).unwrap(), main.rs:214
        std::fwrite(bytes.data(), 1, bytes.size(), f);
        std::fclose(f);
        return RT_OK;
    });
}

} // extern "C"
