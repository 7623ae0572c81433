#!/usr/bin/env python3
"""What the last ulp of sin / cos / acos / atan2 / log does to a pixel (CPU only; build container or GPU box host).

The product and the oracle share rt_math.h's fdlibm restatements so that HIP == oracle bit for bit. The Rust reference
calls the platform libm (f64::sin, cos, acos, atan2, ln: sphere.rs:30-34, constantmedium.rs:61, texture/mod.rs:52,77,
pdf.rs:15-18, vec.rs:112-115), which differs from those restatements by <= 1 ulp on 3-5 % of arguments. This script
renders the same scenes with the oracle built both ways — `make -C oracle` and `make -C oracle libm` (-DRTO_LIBM: the
render path's five transcendentals are glibc's, nothing shared with the product) — and reports how far the pixels move,
next to north_star's 1e-4 relative per channel.

It pins nothing: both sides are the repo's own restatement; the reference cannot be run (no Rust toolchain) and is
unseeded. PARITY UNPINNED. What it does show is the size of the one effect every parity test of the repo is blind
to by construction (the common-mode transcendentals).

    python3 tools/libm_sensitivity.py [--size 96] [--spp 64] > profiles/r3_libm_sensitivity.txt
"""
import argparse
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, HERE)

SCENES = ["final_scene", "cornell_box", "random_scene", "two_perlin_spheres", "earth"]


def compare(scene_name, size, spp, seed=2022, threads=8, assets=None):
    """Render `scene_name` at size x size x spp with both oracle builds; dict of difference statistics."""
    import raytracer_2022_amd as rt
    from oracle import oracle_ffi as O
    scene = rt.HostScene(scene_name, seed=seed, assets_dir=assets)
    cam, bg = scene.default_view(1.0)
    rows = rt.shuffled_rows(size, seed)
    out = {"scene": scene_name, "size": size, "spp": spp}
    for label, n in (("frame", spp), ("paths", 1)):
        p = rt.make_params(size, size, n, 50, bg, seed=seed)
        a, sa = O.render_cpu(scene.desc, cam, p, rows, n_threads=threads, want_stats=True)
        b, sb = O.render_cpu(scene.desc, cam, p, rows, n_threads=threads, want_stats=True, libm=True)
        nan_a, nan_b = np.isnan(a), np.isnan(b)
        ok = ~(nan_a | nan_b)
        rel = np.zeros_like(a)
        denom = np.maximum(np.abs(a), np.abs(b))
        nz = ok & (denom > 0)
        rel[nz] = np.abs(a[nz] - b[nz]) / denom[nz]
        d = {
            "values": int(a.size), "nan_mismatch": int((nan_a != nan_b).sum()),
            "identical": float((a[ok] == b[ok]).mean()),
            "within_1e-12": float((rel[ok] <= 1e-12).mean()), "within_1e-9": float((rel[ok] <= 1e-9).mean()),
            "within_1e-4": float((rel[ok] <= 1e-4).mean()), "max_rel": float(rel[ok].max()),
            "rays": (int(sa.rays), int(sb.rays)), "node_visits": (int(sa.node_visits), int(sb.node_visits)),
            "rng_draws": (int(sa.rng_draws), int(sb.rng_draws)),
            "u8_differ": int((rt.write_color(a, n) != rt.write_color(b, n)).sum()),
        }
        if label == "paths":
            # one sample per pixel: a pixel IS a path; a path that took another branch somewhere (a root accepted on one
            # side of a threshold and rejected on the other, one more turn of a rejection loop) moves by far more than
            # rounding does
            px = rel.reshape(-1, 3).max(axis=1)
            d["paths_moved_more_than_1e-9"] = int((px > 1e-9).sum())
        out[label] = d
    return out


def report(results, fh=sys.stdout):
    w = fh.write
    w("# tools/libm_sensitivity.py: oracle with rt_math.h's sin/cos/acos/atan2/log (what HIP shares) vs the same oracle with glibc's (-DRTO_LIBM;\n")
    w("# what the Rust reference links). Relative difference per channel of the f64 pixel sums, against north_star's 1e-4. PARITY UNPINNED:\n")
    w("# both sides are the repo's restatement; this measures the common-mode blind spot of the parity tests, it pins no bit to the reference.\n")
    for r in results:
        f, p = r["frame"], r["paths"]
        w("%s %dx%dx%d spp: identical %.4f | <=1e-12 %.4f | <=1e-9 %.4f | <=1e-4 %.6f | max rel %.3e | u8 channels that differ %d of %d | NaN mismatches %d\n"
          % (r["scene"], r["size"], r["size"], r["spp"], f["identical"], f["within_1e-12"], f["within_1e-9"], f["within_1e-4"], f["max_rel"],
             f["u8_differ"], f["values"], f["nan_mismatch"]))
        w("    counters rt_math / libm: rays %d / %d, node visits %d / %d, RNG words %d / %d\n"
          % (f["rays"] + f["node_visits"] + f["rng_draws"]))
        w("    1 spp (a pixel = a path): %d of %d paths moved by more than 1e-9 (took another branch); rays %d / %d; max rel %.3e\n"
          % (p["paths_moved_more_than_1e-9"], p["values"] // 3, p["rays"][0], p["rays"][1], p["max_rel"]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=96)
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--scenes", default=",".join(SCENES))
    args = ap.parse_args()
    assets = os.path.join(HERE, "assets")
    res = [compare(s, args.size, args.spp, threads=args.threads, assets=assets if os.path.isdir(assets) else None) for s in args.scenes.split(",")]
    report(res)


if __name__ == "__main__":
    main()
