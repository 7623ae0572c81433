// scenes.cpp — the reference's scene builders (raytracer/src/scene.rs:22-571)
// restated over the host API of scene_api.hpp with a seeded stream in place of
// rand::thread_rng(); draws are consumed in the reference's order.
//
// Build decisions where v1 of the reference is silent:
//  * fns that return only a world there cannot be rendered by its own
//    ray_color (empty light list panics, hittable/mod.rs:130). Here sky-lit
//    scenes return lights = ∅ (cosine-only mode of the integrator) and scenes lit
//    by one emissive rect return that rect un-flipped, as cornell_box does
//    (scene.rs:170-193).
//  * image textures come from <assets>/<stem>.jpg (the reference's own files, decoded by host/jpeg.cpp) or
//    <stem>.ppm or, when neither is there, from a procedural stand-in of the same size.
#include "scene_api.hpp"

#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>

namespace rt2022 {

std::vector<uint8_t> procedural_planet_rgb8(uint32_t w, uint32_t h, uint32_t variant) {
    // Integer-only banded/blotchy pattern: deterministic on every host.
    std::vector<uint8_t> img((size_t)w * h * 3);
    for (uint32_t y = 0; y < h; y++)
        for (uint32_t x = 0; x < w; x++) {
            uint32_t cx = x * 64 / w, cy = y * 32 / h;
            uint64_t hsh = rtm::mix64(((uint64_t)variant << 40) ^ ((uint64_t)cx << 20) ^ cy);
            uint32_t band = (y * 16 / h + variant) & 15;
            uint8_t r = (uint8_t)(40 + (hsh & 0x7F) + band * 4);
            uint8_t g = (uint8_t)(60 + ((hsh >> 8) & 0x7F) + band * 2);
            uint8_t b = (uint8_t)(90 + ((hsh >> 16) & 0x7F));
            uint8_t *p = &img[((size_t)y * w + x) * 3];
            if (variant == 0 && ((hsh >> 24) & 3) != 0) { p[0] = 20; p[1] = (uint8_t)(40 + band * 3); p[2] = (uint8_t)(120 + (hsh & 0x3F)); }
            else { p[0] = r; p[1] = g; p[2] = b; }
        }
    return img;
}

static std::shared_ptr<ImageTexture> load_image(const SceneAssets &assets, const char *stem, uint32_t w, uint32_t h, uint32_t variant) {
    if (!assets.dir.empty())
        for (const char *ext : {".jpg", ".ppm"}) {                      // the reference opens source/<stem>.jpg (scene.rs:128,331,479-497)
            std::string path = assets.dir + "/" + stem + ext;
            std::ifstream probe(path, std::ios::binary);
            if (probe.good()) return std::make_shared<ImageTexture>(path);
        }
    auto px = procedural_planet_rgb8(w, h, variant);
    return std::make_shared<ImageTexture>(w, h, px.data());
}

SceneOut random_scene_n(HostRng &rng, int half_grid) {                                // scene.rs:22-84
    SceneOut s;
    auto checker = make<CheckerTexture>(Color(0.2, 0.3, 0.1), Color(0.9, 0.9, 0.9));
    s.world.add(make<Sphere>(Point3(0.0, -1000.0, 0.0), 1000.0, make<Lambertian>(TexturePtr(checker))));
    for (int a = -half_grid; a <= half_grid; a++) {
        for (int b = -half_grid; b <= half_grid; b++) {
            double choose_mat = rng.gen_f64();
            double cx = (double)a + 0.9 * rng.gen_f64();
            double cz = (double)b + 0.9 * rng.gen_f64();
            Point3 center(cx, 0.2, cz);
            if ((center - Point3(4.0, 0.2, 0.0)).length() > 0.9) {
                if (choose_mat < 0.80) {
                    Color albedo = rng.random_vec(0.0, 1.0);
                    Point3 center2 = center + Vec3(0.0, rng.gen_range(0.0, 0.5), 0.0);
                    s.world.add(make<MovingSphere>(center, center2, 0.0, 1.0, 0.2, make<Lambertian>(albedo)));
                } else if (choose_mat < 0.95) {
                    Color albedo = rng.random_vec(0.5, 1.0);
                    double fuzz = rng.gen_range(0.0, 0.5);
                    s.world.add(make<Sphere>(center, 0.2, make<Metal>(albedo, fuzz)));
                } else {
                    s.world.add(make<Sphere>(center, 0.2, make<Dielectric>(1.5)));
                }
            }
        }
    }
    s.world.add(make<Sphere>(Point3(0.0, 1.0, 0.0), 1.0, make<Dielectric>(1.5)));
    s.world.add(make<Sphere>(Point3(-4.0, 1.0, 0.0), 1.0, make<Lambertian>(Color(0.4, 0.2, 0.1))));
    s.world.add(make<Sphere>(Point3(4.0, 1.0, 0.0), 1.0, make<Metal>(Color(0.7, 0.6, 0.5), 0.0)));
    return s;
}
SceneOut random_scene(HostRng &rng) { return random_scene_n(rng, 11); }

SceneOut two_spheres(HostRng &) {                                                     // scene.rs:87-105
    SceneOut s;
    TexturePtr checker = make<CheckerTexture>(Color(0.2, 0.3, 0.1), Color(0.9, 0.9, 0.9));
    s.world.add(make<Sphere>(Point3(0.0, -10.0, 0.0), 10.0, make<Lambertian>(checker)));
    s.world.add(make<Sphere>(Point3(0.0, 10.0, 0.0), 10.0, make<Lambertian>(checker)));
    return s;
}

SceneOut two_perlin_spheres(HostRng &rng) {                                           // scene.rs:108-124
    SceneOut s;
    TexturePtr pertext = make<NoiseTexture>(4.0, rng);
    s.world.add(make<Sphere>(Point3(0.0, -1000.0, 0.0), 1000.0, make<Lambertian>(pertext)));
    s.world.add(make<Sphere>(Point3(0.0, 2.0, 0.0), 2.0, make<Lambertian>(pertext)));
    return s;
}

SceneOut earth(HostRng &, const SceneAssets &assets) {                                // scene.rs:127-140
    SceneOut s;
    TexturePtr earth_texture = load_image(assets, "earthmap", 1024, 512, 0);
    s.world.add(make<Sphere>(Point3(0.0, 0.0, 0.0), 2.0, make<Lambertian>(earth_texture)));
    return s;
}

SceneOut simple_light(HostRng &rng) {                                                 // scene.rs:143-162
    SceneOut s;
    TexturePtr pertext = make<NoiseTexture>(4.0, rng);
    s.world.add(make<Sphere>(Point3(0.0, -1000.0, 0.0), 1000.0, make<Lambertian>(pertext)));
    s.world.add(make<Sphere>(Point3(0.0, 2.0, 0.0), 2.0, make<Lambertian>(pertext)));
    auto difflight = make<DiffuseLight>(Color(4.0, 4.0, 4.0));
    auto rect = make<XYRect>(3.0, 5.0, 1.0, 3.0, -2.0, difflight);
    s.world.add(rect);
    s.lights.add(rect);                                                               // build decision
    return s;
}

SceneOut cornell_box(HostRng &) {                                                     // scene.rs:165-196
    SceneOut s;
    auto light_strong = make<DiffuseLight>(Color(60.0, 60.0, 60.0));
    auto light_top = make<XZRect>(213.0, 343.0, 127.0, 232.0, 554.0, light_strong);
    s.world.add(make<FlipFace>(light_top));
    auto red = make<Lambertian>(Color(0.65, 0.05, 0.05));
    auto white = make<Lambertian>(Color(0.73, 0.73, 0.73));
    auto green = make<Lambertian>(Color(0.12, 0.45, 0.15));
    s.world.add(make<YZRect>(0.0, 555.0, 0.0, 555.0, 555.0, red));
    s.world.add(make<YZRect>(0.0, 555.0, 0.0, 555.0, 0.0, green));
    s.world.add(make<XZRect>(0.0, 555.0, 0.0, 555.0, 0.0, white));
    s.world.add(make<XZRect>(0.0, 555.0, 0.0, 555.0, 555.0, white));
    s.world.add(make<XYRect>(0.0, 555.0, 0.0, 555.0, 555.0, white));
    s.lights.add(light_top);
    return s;
}

SceneOut cornell_smoke(HostRng &) {                                                   // scene.rs:199-257
    SceneOut s;
    auto red = make<Lambertian>(Color(0.65, 0.05, 0.05));
    auto white = make<Lambertian>(Color(0.73, 0.73, 0.73));
    auto green = make<Lambertian>(Color(0.12, 0.45, 0.15));
    auto light = make<DiffuseLight>(Color(7.0, 7.0, 7.0));
    s.world.add(make<YZRect>(0.0, 555.0, 0.0, 555.0, 555.0, green));
    s.world.add(make<YZRect>(0.0, 555.0, 0.0, 555.0, 0.0, red));
    auto light_rect = make<XZRect>(113.0, 443.0, 127.0, 432.0, 554.0, light);
    s.world.add(make<FlipFace>(light_rect));
    s.world.add(make<XZRect>(0.0, 555.0, 0.0, 555.0, 555.0, white));
    s.world.add(make<XZRect>(0.0, 555.0, 0.0, 555.0, 0.0, white));
    s.world.add(make<XYRect>(0.0, 555.0, 0.0, 555.0, 555.0, white));

    HittablePtr box1 = make<Boxes>(Point3(0.0, 0.0, 0.0), Point3(165.0, 330.0, 165.0), white);
    box1 = make<RotateY>(box1, 15.0);
    box1 = make<Translate>(box1, Vec3(265.0, 0.0, 295.0));
    s.world.add(make<ConstantMedium>(box1, 0.01, Color(0.0, 0.0, 0.0)));

    HittablePtr box2 = make<Boxes>(Point3(0.0, 0.0, 0.0), Point3(165.0, 165.0, 165.0), white);
    box2 = make<RotateY>(box2, -18.0);
    box2 = make<Translate>(box2, Vec3(130.0, 0.0, 65.0));
    s.world.add(make<ConstantMedium>(box2, 0.01, Color(1.0, 1.0, 1.0)));
    s.lights.add(light_rect);                                                         // build decision
    return s;
}

SceneOut final_scene(HostRng &rng, const SceneAssets &assets) {                       // scene.rs:260-362
    SceneOut s;
    HittableList box1;
    auto ground = make<Lambertian>(Color(0.48, 0.83, 0.53));
    const int boxes_per_side = 20;
    for (int i = 0; i < boxes_per_side; i++) {
        for (int j = 0; j < boxes_per_side; j++) {
            double w = 100.0;
            double x0 = -1000.0 + (double)i * w;
            double z0 = -1000.0 + (double)j * w;
            double y0 = 0.0;
            double x1 = x0 + w;
            double y1 = rng.gen_range(1.0, 101.0);
            double z1 = z0 + w;
            box1.add(make<Boxes>(Point3(x0, y0, z0), Point3(x1, y1, z1), ground));
        }
    }
    s.world.add(BvhNode::new_list(box1, 0.0, 1.0, rng));

    auto light = make<DiffuseLight>(Color(7.0, 7.0, 7.0));
    auto light_rect = make<XZRect>(123.0, 423.0, 147.0, 412.0, 554.0, light);
    s.world.add(make<FlipFace>(light_rect));

    Point3 center1(400.0, 400.0, 200.0);
    Point3 center2 = center1 + Vec3(25.0, 0.0, 0.0);
    auto moving_sphere_material = make<Lambertian>(Color(0.7, 0.3, 0.1));
    s.world.add(make<MovingSphere>(center1, center2, 0.0, 1.0, 50.0, moving_sphere_material));

    s.world.add(make<Sphere>(Point3(260.0, 150.0, 45.0), 50.0, make<Dielectric>(1.5)));
    s.world.add(make<Sphere>(Point3(0.0, 150.0, 145.0), 50.0, make<Metal>(Color(0.8, 0.8, 0.9), 1.0)));

    HittablePtr boundary = make<Sphere>(Point3(360.0, 150.0, 145.0), 70.0, make<Dielectric>(1.5));
    s.world.add(boundary);
    s.world.add(make<ConstantMedium>(boundary, 0.2, Color(0.2, 0.4, 0.9)));

    boundary = make<Sphere>(Point3(0.0, 0.0, 0.0), 5000.0, make<Dielectric>(1.5));
    s.world.add(make<ConstantMedium>(boundary, 0.0001, Color(1.0, 1.0, 1.0)));

    TexturePtr emat_tex = load_image(assets, "earthmap", 1024, 512, 0);
    s.world.add(make<Sphere>(Point3(400.0, 200.0, 400.0), 100.0, make<Lambertian>(emat_tex)));

    TexturePtr pertext = make<NoiseTexture>(0.1, rng);
    s.world.add(make<Sphere>(Point3(220.0, 280.0, 300.0), 80.0, make<Lambertian>(pertext)));

    HittableList box2;
    auto white = make<Lambertian>(Color(0.73, 0.73, 0.73));
    const int ns = 1000;
    for (int i = 0; i < ns; i++) box2.add(make<Sphere>(rng.random_vec(0.0, 165.0), 10.0, white));
    s.world.add(make<Translate>(make<RotateY>(BvhNode::new_list(box2, 0.0, 1.0, rng), 15.0), Vec3(-100.0, 270.0, 395.0)));

    s.lights.add(light_rect);                                                         // build decision
    return s;
}

// ---------------------------------------------------------------- meshes ----
namespace {
struct Mesh {
    std::vector<Point3> vertices;
    std::vector<uint32_t> indices;
};

// tobj 3.2.2 semantics as used at scene.rs:368-375: positions parsed as f32 and
// widened (scene.rs:386-388), faces fan-triangulated, only position indices
// matter for Triangle::new. Negative (relative) indices supported.
bool load_obj(const std::string &path, Mesh &m) {
    std::ifstream in(path);
    if (!in) return false;
    std::string line;
    while (std::getline(in, line)) {
        if (line.size() > 1 && line[0] == 'v' && (line[1] == ' ' || line[1] == '\t')) {
            const char *p = line.c_str() + 1;
            char *end;
            float x = std::strtof(p, &end); p = end;
            float y = std::strtof(p, &end); p = end;
            float z = std::strtof(p, &end);
            m.vertices.push_back(Point3((double)x, (double)y, (double)z));
        } else if (line.size() > 1 && line[0] == 'f' && (line[1] == ' ' || line[1] == '\t')) {
            std::istringstream ss(line.substr(1));
            std::string tok;
            std::vector<uint32_t> idx;
            while (ss >> tok) {
                long v = std::strtol(tok.c_str(), nullptr, 10);
                if (v < 0) v = (long)m.vertices.size() + v + 1;
                idx.push_back((uint32_t)(v - 1));
            }
            for (size_t k = 1; k + 1 < idx.size(); k++) {
                m.indices.push_back(idx[0]); m.indices.push_back(idx[k]); m.indices.push_back(idx[k + 1]);
            }
        }
    }
    return !m.indices.empty();
}

// Stand-in for the reference's OBJ models when no file is given: a closed
// "fuselage + wings" surface of revolution with ~n_tris triangles, bounds close to
// Shuttle.obj's ([-0.33,0,-0.48]..[0.34,0.49,0.55]).
Mesh synthetic_shuttle(uint32_t n_tris) {
    Mesh m;
    uint32_t rings = 8, seg = 8;
    while ((uint64_t)rings * seg * 2 < n_tris) { if (rings <= seg * 2) rings *= 2; else seg *= 2; }
    for (uint32_t i = 0; i <= rings; i++) {
        double t = (double)i / (double)rings;                   // along z
        double z = -0.475 + 1.02 * t;
        double prof = 4.0 * t * (1.0 - t);                      // 0 at both ends
        double rad = 0.02 + 0.20 * prof;
        for (uint32_t j = 0; j < seg; j++) {
            double a = (double)j / (double)seg * 2.0 * rtm::PI;
            double sx, cx;
            rtm::sincos_(a, sx, cx);
            double wing = 1.0 + 0.6 * prof * cx * cx;           // flatten into wings along x
            float x = (float)(rad * wing * cx), y = (float)(0.245 + rad * sx), zz = (float)z;
            m.vertices.push_back(Point3((double)x, (double)y, (double)zz));
        }
    }
    for (uint32_t i = 0; i < rings; i++)
        for (uint32_t j = 0; j < seg; j++) {
            uint32_t a = i * seg + j, b = i * seg + (j + 1) % seg, c = (i + 1) * seg + j, d = (i + 1) * seg + (j + 1) % seg;
            m.indices.insert(m.indices.end(), {a, b, c});
            m.indices.insert(m.indices.end(), {b, d, c});
        }
    return m;
}

// One level of midpoint subdivision (x4 triangles); midpoints rounded through f32
// like every other OBJ position.
Mesh subdivide(const Mesh &in) {
    Mesh out;
    out.vertices = in.vertices;
    std::map<std::pair<uint32_t, uint32_t>, uint32_t> mid;
    auto midpoint = [&](uint32_t a, uint32_t b) {
        auto key = std::make_pair(std::min(a, b), std::max(a, b));
        auto it = mid.find(key);
        if (it != mid.end()) return it->second;
        Point3 p = (in.vertices[a] + in.vertices[b]) * 0.5;
        out.vertices.push_back(Point3((double)(float)p.x, (double)(float)p.y, (double)(float)p.z));
        uint32_t id = (uint32_t)out.vertices.size() - 1;
        mid[key] = id;
        return id;
    };
    for (size_t t = 0; t + 2 < in.indices.size(); t += 3) {
        uint32_t a = in.indices[t], b = in.indices[t + 1], c = in.indices[t + 2];
        uint32_t ab = midpoint(a, b), bc = midpoint(b, c), ca = midpoint(c, a);
        out.indices.insert(out.indices.end(), {a, ab, ca});
        out.indices.insert(out.indices.end(), {ab, b, bc});
        out.indices.insert(out.indices.end(), {ca, bc, c});
        out.indices.insert(out.indices.end(), {ab, bc, ca});
    }
    return out;
}

// get_shuttle / get_ship, scene.rs:364-465.
void add_model(HittableList &world, const Mesh &mesh, double zoom, double angle, Vec3 offset, HostRng &rng) {
    HittableList object;
    auto mat = make<Lambertian>(Color(0.78, 0.78, 0.78));
    for (size_t v = 0; v + 2 < mesh.indices.size(); v += 3)
        object.add(make<Triangle>(mesh.vertices[mesh.indices[v]], mesh.vertices[mesh.indices[v + 1]], mesh.vertices[mesh.indices[v + 2]], mat));
    HittablePtr o = BvhNode::new_list(object, 0.0, 1.0, rng);
    o = make<Zoom>(o, zoom);
    o = make<RotateY>(o, angle);
    o = make<Translate>(o, offset);
    world.add(o);
}
} // namespace

SceneOut wwscene(HostRng &rng, const SceneAssets &assets, int shuttle_subdiv) {         // scene.rs:468-571
    SceneOut s;
    auto light_strong = make<DiffuseLight>(Color(130.0, 130.0, 130.0));
    auto light_sphere = make<Sphere>(Point3(800.0, 700.0, -800.0), 70.0, light_strong);
    s.world.add(light_sphere);
    s.lights.add(light_sphere);

    s.world.add(make<Sphere>(Point3(0.0, 0.0, 0.0), 43.0, make<Lambertian>(TexturePtr(load_image(assets, "Saturn", 1280, 640, 1)))));
    s.world.add(make<Sphere>(Point3(150.0, 20.0, 150.0), 26.0, make<Lambertian>(TexturePtr(load_image(assets, "Jupiter", 1024, 512, 2)))));
    s.world.add(make<Sphere>(Point3(480.0, 25.0, 500.0), 25.0, make<Lambertian>(TexturePtr(load_image(assets, "Mars", 800, 383, 3)))));

    for (int i = 0; i < 40; i++) {                                                      // ring stars, scene.rs:507-514
        Vec3 pos = rtm::to_unit(random_in_unit_xz_disk(rng)) * (100.0 + rng.gen_range_inclusive(-15.0, 15.0));
        pos += Vec3(0.0, 0.0, rng.gen_range_inclusive(-1.0, 1.0));
        Color albedo = rng.random_vec(0.5, 1.0);
        double fuzz = rng.gen_range(0.0, 0.5);
        double radius = rng.gen_range_inclusive(0.3, 0.5);
        s.world.add(make<Sphere>(pos, radius, make<Metal>(albedo, fuzz)));
    }
    for (int i = 0; i < 40; i++) {                                                      // scene.rs:515-520
        Vec3 pos = rtm::to_unit(random_in_unit_xz_disk(rng)) * (100.0 + rng.gen_range_inclusive(-15.0, 15.0));
        pos += Vec3(0.0, 0.0, rng.gen_range_inclusive(-1.0, 1.0));
        double radius = rng.gen_range_inclusive(0.3, 0.6);
        s.world.add(make<Sphere>(pos, radius, make<Dielectric>(1.5)));
    }

    const int CNT = 20;                                                                 // rings, scene.rs:523-543
    const size_t delta = 2;
    const size_t weight[CNT] = {2, 3, 2, 3, 4, 3, 2, 2, 3, 2, 3, 4, 3, 6, 4, 5, 3, 3, 4, 3};
    size_t now = 80;
    for (int k = 0; k < CNT; k++) {
        for (size_t i = now * weight[k]; i < (now + delta) * weight[k]; i++) {
            double thickness = weight[k] <= 4 ? rng.gen_range(0.009, 0.01) : rng.gen_range(0.007, 0.008);
            s.world.add(make<Ring>((double)i / (double)weight[k], thickness, make<Lambertian>(Color(0.78, 0.78, 0.78))));
        }
        now += delta;
    }

    for (int i = 0; i <= 100; i++) {                                                    // stars, scene.rs:546-564
        Color scolor = (i % 2 == 0) ? Color(1.0, 1.0, 1.0) : Color(1.0, 1.0, 0.0);
        double x = rng.gen_range_inclusive(-500.0, 500.0);
        double y = rng.gen_range_inclusive(-500.0, 500.0);
        double z = rng.gen_range_inclusive(100.0, 400.0);
        double radius = rng.gen_range_inclusive(0.3, 0.45);
        s.world.add(make<Sphere>(Point3(x, y, z), radius, make<DiffuseLight>(scolor)));
    }

    // Imported objects. Shuttle.obj when present under assets.dir, else the
    // synthetic stand-in; Ship.obj is absent from the reference (.MISSING_LARGE_BLOBS)
    // and is represented by the same mesh with the ship's placement.
    Mesh mesh;
    if (assets.dir.empty() || !load_obj(assets.dir + "/Shuttle.obj", mesh)) mesh = synthetic_shuttle(13079);
    for (int i = 0; i < shuttle_subdiv; i++) mesh = subdivide(mesh);
    add_model(s.world, mesh, 13.5, 56.0, Vec3(40.88, 1.3, -85.59), rng);              // get_shuttle, scene.rs:408-412
    Mesh ship = synthetic_shuttle(2048);
    add_model(s.world, ship, 13.5 * 0.56 * 10.0, 153.0, Vec3(15.0, 2.0, -116.0), rng); // get_ship placement, scene.rs:459-463
    return s;
}

} // namespace rt2022
