"""examples/main.cpp — the reference's main() (main.rs:28-231) over the C ABI, in C++: the five stage banners, the per-worker
progress bar of main.rs:102-127,154-155 fed by rt_params.progress_cb, the elapsed time and the image file."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_example_driver_prints_banners_progress_and_writes_the_image(tmp_path, rt):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "examples")])
    out = tmp_path / "out.ppm"
    p = subprocess.run([os.path.join(ROOT, "examples", "render"), "cornell_box", "96", "96", "32", str(out)], cwd=ROOT,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    for banner in ("[1/5] Initlizing...", "[2/5] Rendering on the GPU...", "[3/5] Collecting Results...", "[4/5] Generating Image...",
                   "[5/5] Outping Image...", "All Work Done.", "Elapsed Time:"):
        assert banner in p.stdout, banner
    assert "GPU 0 [" in p.stderr and "100.0 %" in p.stderr           # the bar reached the end
    img = rt.load_image(str(out))
    assert img.shape == (96, 96, 3) and img.max() == 255 and 5 < img.mean() < 250      # the ceiling light saturates; the box is lit
