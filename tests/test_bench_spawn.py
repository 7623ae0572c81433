"""`python3 bench.py --gpus N` without torch.distributed.run around it: the process becomes the launcher of N rank
children (the reference's main() spawning and joining its workers, main.rs:109-183). What is checked here, on CPU:
the launcher starts N ranks with a consistent rendezvous, has touched no GPU API when it does so (no torch, no
librt2022.so, no HIP runtime in its address space), relays the ranks' outcome as its own exit code, and the explicit
torch.distributed.run form is still accepted. The N > 1 render itself is covered by tests/test_multi.py."""
import json
import os
import subprocess
import sys

import pytest

from conftest import has_gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def run_bench(args, env=None, timeout=600):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + args, env=e, capture_output=True, text=True, timeout=timeout)


@pytest.mark.skipif(has_gpu(), reason="the no-GPU exit path is what this test drives")
def test_launcher_spawns_ranks_before_touching_the_gpu(tmp_path):
    report = tmp_path / "parent.json"
    p = run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0", "--spp", "1"], {"RT2022_BENCH_SPAWN_REPORT": str(report)})
    # both ranks came up, found no GPU, said so and exited 3; the launcher passes that on
    assert p.returncode == 3, p.stderr[-2000:]
    assert p.stderr.count("no GPU visible") == 2, p.stderr[-2000:]
    assert "launch with torch.distributed.run" not in p.stderr
    seen = json.loads(report.read_text())
    assert "torch" not in seen["modules"] and "raytracer_2022_amd" not in seen["modules"]
    for so in seen["shared_objects"]:
        assert "librt2022" not in so and "libamdhip64" not in so and "libhsa-runtime" not in so and "libtorch" not in so, so


def test_launcher_is_not_used_under_an_explicit_launcher():
    # WORLD_SIZE set by torch.distributed.run but contradicting --gpus is still refused, not re-spawned
    p = run_bench(["--gpus", "4", "--steps", "1"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode == 2
    assert "WORLD_SIZE=2" in p.stderr


def test_single_gpu_form_does_not_spawn(tmp_path):
    report = tmp_path / "parent.json"
    p = run_bench(["--gpus", "1", "--steps", "1", "--warmup", "0", "--spp", "1", "--no-pmc", "--no-cpu-baseline"],
                  {"RT2022_BENCH_SPAWN_REPORT": str(report)})
    assert not report.exists()
    if not has_gpu():
        assert p.returncode == 3


@pytest.mark.gpu
@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_rehearsal_two_ranks_sharing_the_card(scaling):
    """The whole N = 2 bench path on a one-GPU box: the launcher, two ranks over gloo sharing the card, one JSON line.
    (With two GPUs visible the same command runs over RCCL; either way the line is checked, not its speed.)"""
    import torch
    env = {} if torch.cuda.device_count() >= 2 else {"RT2022_BENCH_BACKEND": "gloo"}
    p = run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0", "--spp", "8", "--scaling", scaling, "--no-cpu-baseline"], env, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == scaling and d["value"] > 0
    assert d["config"]["frames"] == (2 if scaling == "weak" else 1)
    if env:
        assert "REHEARSAL" in d["data"]
    # weak: every rank renders one frame's worth of rows; strong: the two ranks split one frame
    assert d["config"]["rows_per_gpu"] == (800 if scaling == "weak" else 400)
