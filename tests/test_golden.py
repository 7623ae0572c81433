"""The oracle against the committed golden vectors (tests/golden/, regression pins made by
make_golden.py) — bit for bit: pixel sums, u8 pixels, ray / node / primitive counters."""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
INDEX = json.load(open(os.path.join(HERE, "golden", "golden_index.json")))


def load_case(name, rt):
    c = INDEX[name]
    g = np.load(os.path.join(HERE, "golden", "golden_%s.npz" % name))
    s = rt.HostScene(c["scene"], seed=c["seed"], param=c["param"])
    cam, bg = s.default_view(c["width"] / c["height"])
    p = rt.make_params(c["width"], c["height"], c["spp"], c["max_depth"], bg, seed=c["seed"], n_frames=c["n_frames"],
                       spp_chunk=c["spp_chunk"])
    return c, g, s, cam, p


@pytest.mark.parametrize("name", sorted(INDEX))
def test_oracle_reproduces_golden(rt, O, name):
    c, g, s, cam, p = load_case(name, rt)
    out, st = O.render_cpu(s.desc, cam, p, g["rows"], n_threads=2, want_stats=True)
    assert st.as_dict() == c["counters"]
    assert np.array_equal(out.view(np.uint64), g["rgb_sum"].view(np.uint64)), "f64 sums are not bit-identical"
    assert np.array_equal(O.write_color(out, c["spp"]), g["rgb8"])
    assert np.array_equal(rt.write_color(out, c["spp"]), g["rgb8"])      # the product's write_color agrees


def test_oracle_thread_count_does_not_change_results(rt, O):
    c, g, s, cam, p = load_case("final_scene", rt)
    a = O.render_cpu(s.desc, cam, p, g["rows"], n_threads=1)
    b = O.render_cpu(s.desc, cam, p, g["rows"], n_threads=5)
    assert np.array_equal(a.view(np.uint64), b.view(np.uint64))


def test_one_sample_per_item_is_the_running_sum(rt, O):
    """spp_chunk = 1 (0 + L0 + L1 + ...) has the bits of spp_chunk = 0, the reference's loop (main.rs:144-151)."""
    for scene in ("cornell_box", "final_scene"):
        s = rt.HostScene(scene, seed=4)
        cam, bg = s.default_view(1.5)
        rows = np.arange(20, dtype=np.uint32)
        outs = [O.render_cpu(s.desc, cam, rt.make_params(30, 20, 6, 50, bg, seed=4, spp_chunk=k), rows, n_threads=4) for k in (0, 1, 4)]
        assert np.array_equal(outs[0].view(np.uint64), outs[1].view(np.uint64)), scene
        assert np.allclose(outs[0], outs[2], rtol=1e-12, atol=0, equal_nan=True), scene


def test_rows_are_independent(rt, O):
    """RNG keyed by (seed, frame, pixel, sample): any row subset gives the same pixels (multi-GPU invariance)."""
    c, g, s, cam, p = load_case("cornell_box", rt)
    rows = g["rows"]
    part = O.render_cpu(s.desc, cam, p, rows[5:9], n_threads=1)
    assert np.array_equal(part.view(np.uint64), g["rgb_sum"][5:9].view(np.uint64))
