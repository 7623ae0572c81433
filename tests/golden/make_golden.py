#!/usr/bin/env python3
"""Generates the golden vectors under tests/golden/ with the CPU oracle.

The reference cannot produce vectors (Rust, no toolchain here; unseeded RNG; no tests or
fixtures of its own — SURVEY.md §4, §8c), so these are REGRESSION PINS made by the oracle
itself after it passed its known-answer tests (tests/test_oracle_kat.py): they freeze the
seeded scenes, the RNG keying and every arithmetic detail, and both the oracle and the
HIP path must keep reproducing them bit for bit.

    python tests/golden/make_golden.py        # rewrites golden_*.npz and golden_index.json
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

CASES = [
    # name, scene, param, W, H, spp, depth, chunk, frames, seed
    ("cornell_box", "cornell_box", 0, 24, 24, 4, 50, 0, 1, 2022),
    ("final_scene", "final_scene", 0, 24, 24, 2, 50, 0, 1, 2022),
    ("final_scene_chunked_2frames", "final_scene", 0, 16, 12, 5, 50, 2, 2, 7),
    ("random_scene", "random_scene", 0, 30, 20, 2, 50, 0, 1, 2022),
    ("cornell_smoke", "cornell_smoke", 0, 20, 20, 2, 50, 0, 1, 2022),
    ("two_perlin_spheres", "two_perlin_spheres", 0, 24, 16, 2, 50, 0, 1, 2022),
    ("simple_light", "simple_light", 0, 24, 16, 3, 50, 0, 1, 2022),
    ("earth", "earth", 0, 24, 16, 2, 50, 0, 1, 2022),
    ("two_spheres_depth3", "two_spheres", 0, 24, 16, 2, 3, 0, 1, 2022),
    ("wwscene", "wwscene", 0, 32, 18, 1, 50, 0, 1, 2022),
]


def render_case(case, O, rt):
    name, scene, param, W, H, spp, depth, chunk, frames, seed = case
    s = rt.HostScene(scene, seed=seed, param=param)
    cam, bg = s.default_view(W / H)
    p = rt.make_params(W, H, spp, depth, bg, seed=seed, n_frames=frames, spp_chunk=chunk)
    rows = rt.shuffled_rows(H * frames, seed + 1)
    out, st = O.render_cpu(s.desc, cam, p, rows, n_threads=4, want_stats=True)
    return s, cam, p, rows, out, st


def main():
    import raytracer_2022_amd as rt
    from oracle import oracle_ffi as O
    index = {}
    for case in CASES:
        name = case[0]
        s, cam, p, rows, out, st = render_case(case, O, rt)
        np.savez_compressed(os.path.join(HERE, "golden_%s.npz" % name), rgb_sum=out, rows=rows,
                            rgb8=O.write_color(out, case[5]))
        index[name] = {"scene": case[1], "param": case[2], "width": case[3], "height": case[4], "spp": case[5],
                       "max_depth": case[6], "spp_chunk": case[7], "n_frames": case[8], "seed": case[9],
                       "counters": st.as_dict()}
        print(name, st.as_dict()["rays"], "rays")
    with open(os.path.join(HERE, "golden_index.json"), "w") as f:
        json.dump(index, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
