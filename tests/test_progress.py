"""rt_params.progress_cb — the per-worker progress of the reference's render threads (main.rs:102-127,154-155: one indicatif
bar per thread, advanced as its rows finish). Speed only: the callback never changes a bit of the result."""
import numpy as np
import pytest


@pytest.mark.gpu
def test_progress_is_monotonic_and_ends_at_the_total(rt):
    W, H, spp = 160, 120, 24
    scene = rt.HostScene("cornell_box", seed=7)
    cam, bg = scene.default_view(W / H)
    p = rt.make_params(W, H, spp, 50, bg, seed=7, spp_chunk=1)
    rows = rt.shuffled_rows(H, 7)
    dev = rt.DeviceScene(scene.desc)
    plain = dev.render(cam, p, rows)
    seen = []
    out = dev.render(cam, p, rows, progress=lambda worker, done, total: seen.append((worker, done, total)))
    assert np.array_equal(out, plain, equal_nan=True)
    assert seen, "the callback was never called"
    total = W * H * spp
    assert all(w == 0 and t == total for w, _, t in seen)
    done = [d for _, d, _ in seen]
    assert done == sorted(done) and done[-1] == total and all(0 < d <= total for d in done)
    assert done.count(total) == 1                     # exactly one final call
    # a chunked job reports in camera paths too, not in work items
    p4 = rt.make_params(W, H, spp, 50, bg, seed=7, spp_chunk=5)
    seen.clear()
    dev.render(cam, p4, rows, progress=lambda worker, done, total: seen.append((worker, done, total)))
    assert seen[-1][1] == total == seen[-1][2]


@pytest.mark.gpu
def test_progress_of_a_device_set_names_its_workers(rt):
    import torch
    n = max(1, min(torch.cuda.device_count(), 4))
    W, H, spp = 96, 64, 8
    scene = rt.HostScene("two_spheres", seed=3)
    cam, bg = scene.default_view(W / H)
    p = rt.make_params(W, H, spp, 50, bg, seed=3, spp_chunk=1)
    rows = rt.shuffled_rows(H, 3)
    sset = rt.DeviceSceneSet(scene.desc, (1 << n) - 1)
    seen = []
    sset.render(cam, p, rows, progress=lambda worker, done, total: seen.append((worker, done, total)))
    workers = sorted({w for w, _, _ in seen})
    assert workers == list(range(n))
    for k in range(n):                                # every worker ends at ITS share of the paths
        last = [s for s in seen if s[0] == k][-1]
        assert last[1] == last[2] == len(rows[k::n]) * W * spp
