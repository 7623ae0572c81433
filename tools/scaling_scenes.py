"""Scene-size scaling on one GPU (SURVEY.md §8d "scaling scenes"): the random_scene rule on a (2k+1)^2 grid,
N ~ 488 / 1e4 / 1e5 / 1e6 spheres, 1200x800, one bench-style line per size (rays from a counter pass)."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytracer_2022_amd as rt

W, H = 1200, 800
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 32
for k in (11, 50, 158, 500):
    t0 = time.time()
    s = rt.HostScene("random_scene", seed=2022, param=k)
    t_build = time.time() - t0
    d = s.desc
    cam, bg = s.default_view(W / H)
    rows = rt.shuffled_rows(H, 2022)
    dev = rt.DeviceScene(d)
    info = dev.info()
    pc = rt.make_params(W, H, 2, 50, bg, seed=2022, spp_chunk=1)
    _, st = dev.render(cam, pc, rows, want_stats=True)
    p = rt.make_params(W, H, spp, 50, bg, seed=2022, spp_chunk=1)
    dev.render(cam, p, rows)                      # warm-up (pool allocation)
    t0 = time.time()
    dev.render(cam, p, rows)
    dt = time.time() - t0
    rays = st.rays * spp / 2.0
    nodes_per_ray = st.node_visits / st.rays
    print(json.dumps({"spheres": int(d.n_spheres + d.n_moving_spheres), "nodes": int(d.n_nodes), "bvh_MB": round(d.n_nodes * 64 / 1e6, 1),
                      "stack_need": info["stack_need"], "host_build_s": round(t_build, 2), "spp": spp, "ms": round(dt * 1e3, 1),
                      "Mrays_per_s": round(rays / dt / 1e6, 1), "node_visits_per_ray": round(nodes_per_ray, 1)}), flush=True)
