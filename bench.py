#!/usr/bin/env python3
"""bench.py — Mrays/s of the path-tracing hot loop on MI355X.

One step = one pass of the hot path over one frame's worth of rows per GPU: the book-2 final scene at 800x800x1000 spp
(BASELINE.json's metric config; --config c1|c2|c4|c5|s1e4|s1e5|s1e6 for the others) with the scene, camera and row list already
resident in HBM. Steps are handed to the library up to four frames per rt_render_device call (an n-frame film strip: the path pool
then drains once per call, not once per frame; every frame is bit-identical to the frame rendered alone). At N GPUs the default job
is an N-frame strip per step whose rows are dealt cyclically to the ranks (weak scaling, raytracer_2022_amd/film.py); `--scaling
strong` splits ONE frame's rows over the ranks instead (BASELINE configs 4 and 5). The only exchange is the gather of the row
buffers. `python3 bench.py --gpus N` starts its own N rank processes (children, before this process touches the GPU); under
torch.distributed.run it takes its rank from the environment.

Prints ONE JSON line on rank 0 (contract in the task statement), with
  roofline      the dominant kernel (wf_trace). bound / achieved / peak / frac = the resource the counters of THIS run show closest to
                its ceiling — the measured busy share of the vector issue slots, or the HBM bytes of FETCH_SIZE / WRITE_SIZE — with the
                others ranked beside it (limiter); SURVEY.md §8(d)'s contractual figure (algorithmic bytes per launch / mean launch
                duration from HIP events / 8 TB/s) under contract_sec8d; every figure recomputable from the fields beside it.
                `value` is the PLAIN path: the timed steps carry no events and the library cuts its pool into its own number of groups
                (two: one group's shade pass beside another's traversal pass); the per-kernel times come from ONE more call of the same
                size made right after the timed region with RT_FLAG_KERNEL_TIMES (one group: a launch's duration is its own) —
                `instrumented_call`
  cpu_baseline  the CPU oracle on a bounded sample of the same workload, at the reference's 8 threads (main.rs:40) and at all usable
                cores (c1: also the whole frame)
The PMC figures come from rocprofv3 passes this script runs on itself before it touches the GPU (N=1 only; `--no-pmc` skips them,
and the traffic then falls back to the committed profile of the same workload, marked as such).
"""
import argparse
import csv
import ctypes as C
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

CONFIGS = {
    # name: (scene, width, height, spp, description)
    # BASELINE.json configs[0]: the reference's own CPU-runnable case; here the GPU figure beside the oracle run IN FULL (cpu_baseline.full_frame)
    "c1": ("random_scene", 400, 225, 100, "book-1 final scene (random spheres) 400x225x100spp depth 50"),
    "c3": ("final_scene", 800, 800, 1000, "book-2 final scene 800x800x1000spp depth 50"),
    "c2": ("random_scene", 1200, 800, 500, "book-1 final scene (random spheres) 1200x800x500spp depth 50"),
    "c4": ("cornell_box", 600, 600, 1000, "book-3 Cornell box, MixturePdf, 600x600x1000spp depth 50"),
    # 8-GPU config of BASELINE.json; the mesh is assets/Shuttle.obj subdivided three times (837 056 triangles)
    "c5": ("wwscene", 1920, 1080, 2000, "OBJ mesh scene (Shuttle.obj x 3 subdivisions = 0.84M triangles) + planet textures 1920x1080x2000spp depth 50"),
    # north_star's synthetic random-sphere table (SURVEY.md §8d "scaling scenes"): the random_scene rule on a (2k+1)^2 grid
    "s1e4": ("random_scene", 1200, 800, 160, "synthetic random spheres, (2k+1)^2 grid with k=50: 1.0e4 spheres, 1200x800x160spp depth 50"),
    "s1e5": ("random_scene", 1200, 800, 160, "synthetic random spheres, (2k+1)^2 grid with k=158: 1.0e5 spheres, 1200x800x160spp depth 50"),
    "s1e6": ("random_scene", 1200, 800, 160, "synthetic random spheres, (2k+1)^2 grid with k=500: 1.0e6 spheres (67 MB of BVH nodes: beyond the 32 MiB of L2), 1200x800x160spp depth 50"),
}
SCENE_PARAM = {"c5": 3, "s1e4": 50, "s1e5": 158, "s1e6": 500}

# Algorithmic bytes per unit, SURVEY.md §8(d).
BYTES_NODE = 64
BYTES_PRIM = {"sphere": 40, "moving_sphere": 80, "rect": 48, "box": 56, "triangle": 80, "ring": 32,
              "medium": 24, "translate": 56, "rotate_y": 56, "zoom": 56, "list": 8, "node": 0}
BYTES_RAY_READ = 128       # per ray segment: 128 B of state read ...
BYTES_RAY_WRITE = 128      # ... and 128 B written
BYTES_PIXEL = 24
# Peaks, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0      # HBM3E, spec (6 290 measured by a float4 copy)
L2_PEAK_GBS = 34500.0      # aggregate of the eight 4 MiB L2s
LDS_PEAK_GBS = 256 * 256 * 2.4  # 256 CUs x 256 B per clock (conflict-free ds_read_b128) x 2.4 GHz = 157 TB/s
BYTES_NODE_LDS = 56        # what a node step reads from the LDS node table: 48 B of box + 8 B of child refs
FP64_VECTOR_TFLOPS = 78.6  # = 39.3 T f64 lane-instructions/s (an FMA counts two flops)
N_SIMDS = 256 * 4
L2_BYTES = 32 << 20


def algorithmic_bytes(st, n_pixels):
    """bytes = 64 N_node + sum_type size N_type + 256 N_ray + 24 N_pixel, split by the kernel that
    moves them: the traversal kernel fetches nodes and primitives, reads each ray and writes its winner (half of the
    per-ray state traffic); the shading kernel has the other half and the pixels."""
    from raytracer_2022_amd import _ffi as F
    prim = sum(BYTES_PRIM[F.KIND_NAMES[k]] * st["prim_tests"][k] for k in range(F.RT_KIND_COUNT))
    traversal = BYTES_NODE * st["node_visits"] + prim
    trace = traversal + BYTES_RAY_READ * st["rays"]
    shade = BYTES_RAY_WRITE * st["rays"] + BYTES_PIXEL * n_pixels
    return {"traversal": traversal, "wf_trace": trace, "wf_shade": shade, "total": trace + shade}


def scene_bytes(desc):
    from raytracer_2022_amd import _ffi as F
    d = desc
    return (d.n_nodes * C.sizeof(F.rt_bvh_node) + d.n_spheres * C.sizeof(F.rt_sphere) + d.n_moving_spheres * C.sizeof(F.rt_moving_sphere)
            + d.n_rects * C.sizeof(F.rt_rect) + d.n_boxes * C.sizeof(F.rt_box) + d.n_triangles * C.sizeof(F.rt_triangle)
            + d.n_rings * C.sizeof(F.rt_ring) + d.n_media * C.sizeof(F.rt_medium) + d.n_xforms * C.sizeof(F.rt_xform))


def usable_cores():
    """CPUs this process may really use: affinity mask capped by the cgroup CPU quota (the GPU box
    shows 256 logical CPUs but grants a 16-CPU share)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except Exception:
        pass
    return max(1, n)


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def committed_traffic(config, spp, spp_chunk):
    """HBM bytes per step from the committed rocprofv3 PMC passes (profiles/*hbm_traffic.json), when they were
    taken on exactly this workload; otherwise None. The fallback when the in-run PMC passes are off or fail."""
    best = None
    for name in sorted(glob.glob(os.path.join(HERE, "profiles", "*hbm_traffic.json"))):
        try:
            for t in json.load(open(name))["runs"]:
                if t.get("config") == config and t.get("spp") == spp and t.get("spp_chunk") == spp_chunk:
                    best = (t, os.path.relpath(name, HERE))
        except Exception:
            pass
    return best


# ---------------------------------------------------------------------------------------------
# In-run PMC passes: this script under rocprofv3, in `--inner-frame` mode (one frame through the
# library's host-buffer entry point, no torch), once per counter set — FETCH_SIZE and WRITE_SIZE
# cannot share a pass (TCC slots), and the counter passes never carry a trace (MI355X_MICROARCH.md).
# ---------------------------------------------------------------------------------------------
PMC_PASSES = [
    ["FETCH_SIZE", "GRBM_GUI_ACTIVE"],
    ["WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum"],
    # issue side (8 SQ slots): see valu_busy()
    ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU2", "SQ_THREAD_CYCLES_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_BUSY_CYCLES", "SQ_INSTS_SALU", "SQ_WAIT_INST_ANY"],
    # instruction mix by the counters (8 SQ slots)
    ["SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64", "SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_INT64",
     "SQ_INSTS_VALU_CVT", "SQ_INSTS_LDS"],
]
N_SHADER_ENGINES = 32      # 4 per XCD x 8 XCDs: rocprofv3 sums the SQ counters over them


def valu_busy(c):
    """MEASURED busy fraction of the vector pipes of one kernel, from counters of ONE pass (no wall clock, no assumed frequency).
    Normalisation, checked on kernels that saturate the pipes by construction (tools/valu_calib.sh, profiles/r3_valu_calibration.txt):
      cycles      = SQ_BUSY_CYCLES / 32        cycles with a wave present, summed by rocprofv3 over the 32 shader engines
      issue slots = 1024 SIMDs x cycles / 4    a SIMD issues vector instructions in quad-cycles: one 64-bit instruction, or up to
                                               two 32-bit ones from different waves (SQ_ACTIVE_INST_VALU2 = quad-cycles where two issued)
      busy        = (SQ_INSTS_VALU - SQ_ACTIVE_INST_VALU2) / issue slots      = share of the slots in which the pipe issued at all
    (SQ_ACTIVE_INST_VALU, which VERDICT r2 asked for, reads exactly SQ_INSTS_VALU on gfx950 — one unit per instruction whatever
    its width — so it cannot serve as a cycle count by itself; the calibration file shows both.)"""
    if not c.get("SQ_BUSY_CYCLES") or not c.get("SQ_INSTS_VALU"):
        return None
    cycles = c["SQ_BUSY_CYCLES"] / N_SHADER_ENGINES
    slots = N_SIMDS * cycles / 4.0
    iv, v2 = c["SQ_INSTS_VALU"], c.get("SQ_ACTIVE_INST_VALU2", 0.0)
    util = c.get("SQ_THREAD_CYCLES_VALU", 0.0) / (iv * 64.0)
    out = {"busy": round((iv - v2) / slots, 4), "instructions_per_slot": round(iv / slots, 4), "dual_issue_slots": round(v2 / slots, 4),
           "lane_utilisation": round(util, 4), "useful": round((iv - v2) / slots * util, 4),
           "wave_instructions_per_step": int(iv), "kernel_cycles_per_step": int(cycles),
           "wait_any_over_wave_cycles": round(c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"], 4) if c.get("SQ_WAVE_CYCLES") else None,
           "issue_stall_over_wave_cycles": round(c.get("SQ_WAIT_INST_ANY", 0.0) / c["SQ_WAVE_CYCLES"], 4) if c.get("SQ_WAVE_CYCLES") else None,
           "salu_per_valu": round(c.get("SQ_INSTS_SALU", 0.0) / iv, 3)}
    f64 = sum(c.get(k, 0.0) for k in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64"))
    if f64:
        out["mix"] = {"f64_add_mul_fma_trans": round(f64 / iv, 4), "int32": round(c.get("SQ_INSTS_VALU_INT32", 0.0) / iv, 4),
                      "int64": round(c.get("SQ_INSTS_VALU_INT64", 0.0) / iv, 4), "cvt": round(c.get("SQ_INSTS_VALU_CVT", 0.0) / iv, 4),
                      "lds_per_valu": round(c.get("SQ_INSTS_LDS", 0.0) / iv, 4),
                      "note": "share of SQ_INSTS_VALU by the per-type counters (compares, selects, moves, min/max are in none of them)"}
    return out


def run_pmc(args, spp, spp_chunk, budget_s):
    """{kernel: {counter: total over one frame}} or (None, reason). Runs before the parent touches the GPU."""
    exe = shutil.which("rocprofv3")
    if not exe:
        return None, "rocprofv3 not on PATH"
    out = {}
    t_all = time.time()
    tmp = tempfile.mkdtemp(prefix="rt2022_pmc_", dir=os.environ.get("TMPDIR", "/tmp"))
    try:
        for i, counters in enumerate(PMC_PASSES):
            left = budget_s - (time.time() - t_all)
            if left < 20:
                return None, "PMC time budget spent after %d passes" % i
            d = os.path.join(tmp, "p%d" % i)
            cmd = [exe, "--pmc"] + counters + ["--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__),
                                               "--inner-frame", "--config", args.config, "--spp", str(spp), "--spp-chunk", str(spp_chunk),
                                               "--seed", str(args.seed), "--assets", args.assets]
            env = dict(os.environ, TMPDIR=os.environ.get("TMPDIR", "/tmp"))
            p = subprocess.Popen(cmd, cwd=tmp, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, start_new_session=True)
            try:
                log, _ = p.communicate(timeout=left)
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(p.pid, 9)           # (the group this call started, nothing else)
                except Exception:
                    pass
                p.wait()
                return None, "PMC pass %d timed out" % i
            if p.returncode != 0:
                return None, "PMC pass %d failed: %s" % (i, log.decode(errors="replace")[-300:].replace("\n", " | "))
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if not files:
                return None, "PMC pass %d wrote no counter_collection.csv" % i
            for f in files:
                with open(f) as fh:
                    for row in csv.DictReader(fh):
                        k = row.get("Kernel_Name", "")
                        name = "wf_trace" if "wf_trace" in k else "wf_shade" if "wf_shade" in k else "chunk_sum" if "chunk_sum" in k else None
                        if name is None:
                            continue
                        slot = out.setdefault(name, {})
                        slot[row["Counter_Name"]] = slot.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
                        if i == 0 and row["Counter_Name"] == counters[0]:
                            slot["_dispatches"] = slot.get("_dispatches", 0) + 1
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return out, "rocprofv3 --pmc, %d passes of one frame each, %.0f s" % (len(PMC_PASSES), time.time() - t_all)


def hbm_bytes(c):
    """MI355X_MICROARCH.md §HBM: rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB; on gfx950 FETCH_SIZE counts 64 B
    per 128-B request, so the read side is doubled; WRITE_SIZE is exact for 16-B-per-lane stores."""
    rd = c.get("FETCH_SIZE", 0.0) * 1024.0 * 2.0
    wr = c.get("WRITE_SIZE", 0.0) * 1024.0
    return rd, wr


def inner_frame(args):
    """One frame of the workload through rt_render (host buffers): the dispatches the profiler counts."""
    import raytracer_2022_amd as rt
    scene_name, W, H, spp, _ = CONFIGS[args.config]
    spp = args.spp or spp
    assets = args.assets if os.path.isdir(args.assets) else None
    scene = rt.HostScene(scene_name, seed=args.seed, assets_dir=assets, param=SCENE_PARAM.get(args.config, 0))
    cam, bg = scene.default_view(W / H)
    params = rt.make_params(W, H, spp, 50, bg, seed=args.seed, spp_chunk=args.spp_chunk)
    rows = rt.shuffled_rows(H, args.seed)
    dev = rt.DeviceScene(scene.desc)
    # (one group of pool segments, like the instrumented call: the counters of a dispatch are then that kernel's alone — with the
    # library's default of two groups a traversal launch shares the SIMDs with the other group's shade launch)
    dev.set_tuning(18 | (1 << 8) | (2 << 12) | (8 << 16) | (2 << 20) | (1 << 24))
    out = dev.render(cam, params, rows)
    assert np.isfinite(out).any()


def cpu_full_frame(scene, cam, params, height, seed):
    """BASELINE.json configs[0] (book-1 final scene 400x225x100): the reference's own CPU-runnable case — the oracle renders
    the WHOLE frame, at the reference's 8 threads and at all usable cores."""
    from oracle import oracle_ffi as O
    from raytracer_2022_amd import _ffi as F, shuffled_rows
    rows = shuffled_rows(height, seed)
    p = F.rt_params.from_buffer_copy(params)
    p.spp_chunk, p.n_frames = 0, 1
    out = {}
    for label, n in (("t8", 8), ("all", usable_cores())):
        if label == "all" and n == 8:
            out["all"] = dict(out["t8"])
            continue
        t0 = time.time()
        _, st = O.render_cpu(scene.desc, cam, p, rows, n_threads=n, want_stats=True)
        dt = time.time() - t0
        out[label] = {"value": round(st.rays / dt / 1e6, 3), "unit": "Mrays/s", "cores": n, "ms_per_frame": round(dt * 1e3, 1), "rays": int(st.rays)}
    return out


def cpu_baseline(scene, cam, params, height, seed, target_s):
    """Time the oracle (the reference's threading scheme, main.rs:93-116) on a bounded sample of the same workload:
    at all usable cores, and at the reference's THREAD_NUMBER = 8 (main.rs:40)."""
    from oracle import oracle_ffi as O
    from raytracer_2022_amd import _ffi as F, shuffled_rows
    cores = usable_cores()
    rows = shuffled_rows(height, seed)
    p = F.rt_params.from_buffer_copy(params)
    p.spp, p.spp_chunk, p.n_frames = 1, 0, 1
    sub = rows[: max(cores, height // 16)]
    t0 = time.time()
    _, st = O.render_cpu(scene.desc, cam, p, sub, n_threads=cores, want_stats=True)
    dt = max(time.time() - t0, 1e-3)
    rate = st.rays / dt
    rays_per_row_spp = st.rays / len(sub)

    def sample(n_threads, seconds):
        want_rays = rate * seconds * n_threads / cores
        spp = int(max(1, min(params.spp, want_rays / (rays_per_row_spp * height))))
        n_rows = int(max(n_threads, min(height, want_rays / (rays_per_row_spp * spp))))
        p.spp = spp
        t0 = time.time()
        _, s2 = O.render_cpu(scene.desc, cam, p, rows[:n_rows], n_threads=n_threads, want_stats=True)
        dt = time.time() - t0
        return {"value": round(s2.rays / dt / 1e6, 3), "cores": n_threads, "rays": int(s2.rays), "seconds": round(dt, 2),
                "sample": "%d shuffled rows x %d px x %d spp of the same scene/camera, %d threads in the reference's "
                          "contiguous-section scheme (main.rs:109-116), %.1f s" % (n_rows, params.width, spp, n_threads, dt)}
    full = sample(cores, target_s * 0.6)
    out = {"value": full["value"], "unit": "Mrays/s", "cores": cores, "kind": "port", "sample": full["sample"],
           "rays": full["rays"], "seconds": full["seconds"], "cpu_model": cpu_model(),
           "logical_cpus_visible": os.cpu_count(), "cores_note": "cores = affinity mask capped by the cgroup CPU quota"}
    if cores > 8:
        t8 = sample(8, target_s * 0.4)
        out["t8"] = {"value": t8["value"], "unit": "Mrays/s", "cores": 8, "sample": t8["sample"],
                     "note": "the reference's hard-coded THREAD_NUMBER (main.rs:40)"}
    else:
        out["t8"] = {"value": full["value"] if cores == 8 else None, "unit": "Mrays/s", "cores": 8,
                     "note": "only %d cores usable here" % cores}
    return out


def spawn_ranks(n):
    """Start `n` copies of this command as rank processes (children, never an exec), one per GPU, with the rendezvous
    variables torch.distributed reads; relay rank 0's stdout (the JSON line) and return the worst exit code.
    Nothing in this process has imported torch, loaded librt2022.so or made a HIP call when the children start
    (RT2022_BENCH_SPAWN_REPORT=<file> writes what is loaded at that moment — tests/test_bench_spawn.py)."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:      # a free rendezvous port on the loopback interface
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    report = os.environ.get("RT2022_BENCH_SPAWN_REPORT")
    if report:
        with open("/proc/self/maps") as fh:
            libs = sorted({line.split()[-1] for line in fh if ".so" in line})
        with open(report, "w") as fh:
            json.dump({"modules": sorted(sys.modules), "shared_objects": libs}, fh)
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.pop("RT2022_BENCH_SPAWN_REPORT", None)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    worst = 0
    failed_at = None
    out0 = b""
    try:
        live = set(range(n))
        while live:
            for r in sorted(live):
                if r == 0:
                    try:
                        out0, _ = procs[0].communicate(timeout=0.2)
                    except subprocess.TimeoutExpired:
                        continue
                elif procs[r].poll() is None:
                    continue
                live.discard(r)
                rc = procs[r].returncode
                if rc != 0:
                    worst = rc if worst == 0 or abs(rc) > abs(worst) else worst
                    if failed_at is None:
                        failed_at = time.time()
            # a rank that failed leaves the others waiting at a collective: give them a moment, then end them (these PIDs only)
            if failed_at is not None and live and time.time() - failed_at > 20.0:
                for r in live:
                    procs[r].kill()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
                p.wait()
    sys.stdout.write(out0.decode(errors="replace"))
    sys.stdout.flush()
    if worst < 0:                           # killed by a signal
        worst = 128 - worst
    return worst


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="c3", choices=sorted(CONFIGS))
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: one frame per GPU (an N-frame strip); strong: ONE frame, its rows dealt over the GPUs")
    ap.add_argument("--spp", type=int, default=0, help="override samples per pixel (a reduced-spp run is NOT the headline number)")
    ap.add_argument("--spp-chunk", type=int, default=-1,
                    help="samples per work item; default: 1 (the reference's summation order) unless the per-sample partial sums "
                         "would exceed 100 GB of HBM, then the smallest chunk that fits")
    ap.add_argument("--seed", type=int, default=2022)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--no-gather", action="store_true")
    ap.add_argument("--quorum", type=int, default=0, help="lanes that must want a node step for the fast path, 1..64 (0: the library's default, 18)")
    ap.add_argument("--segs", type=int, default=0, help="pool size: segments of 4096 path slots per resident traversal workgroup, 1..8 (0: the library's default, 8)")
    ap.add_argument("--groups", type=int, default=0, help="groups of pool segments passing independently on streams of their own (0: the library's default)")
    ap.add_argument("--no-pmc", action="store_true", help="skip the in-run rocprofv3 counter passes")
    ap.add_argument("--no-plain", action="store_true", help="(accepted for older scripts; no effect: the timed region IS the plain path now)")
    ap.add_argument("--partial-ring", type=int, default=0,
                    help="planes of the library's partial-sum ring: 0 = its own choice (a ring when all spp planes of a call would exceed 40 % of the device's memory), -1 never, n force")
    ap.add_argument("--frames-per-call", type=int, default=0,
                    help="steps (frames per GPU) handed to the library per rt_render_device call; default: min(4, steps, what keeps the call's partial sums under 100 GB)")
    ap.add_argument("--pmc-seconds", type=float, default=240.0, help="time budget of the counter passes")
    ap.add_argument("--assets", default=os.path.join(HERE, "assets"))
    ap.add_argument("--inner-frame", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.inner_frame:
        inner_frame(args)
        return

    # `python3 bench.py --gpus N` by itself (no torch.distributed.run around it): this process becomes the launcher —
    # the reference's main() spawning and joining its workers, main.rs:109-183 — BEFORE anything here has touched the GPU.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE=%d; launch with torch.distributed.run --nproc-per-node %d"
                  % (args.gpus, world, args.gpus), file=sys.stderr)
        sys.exit(2)

    scene_name, W, H, spp, desc_text = CONFIGS[args.config]
    if args.spp > 0:
        spp = args.spp
    # One STEP = one frame per GPU (weak scaling; strong: one frame over all GPUs). Steps are handed to the library a few frames at a
    # time — ONE rt_render_device call renders `fpc` frames' worth of rows, an fpc-frame film strip whose frames differ only in their
    # RNG key (rt2022.h: row id g = frame g / height) — because the engine drains its path pool at the end of every call: the last
    # third of a frame's passes run on a thinning pool (tools/passlog.py: 7.5 % of the traversal time of a one-frame call is that
    # tail), and inside one call the next frame's paths fill the slots the previous frame's stragglers leave free.
    # Every frame of a strip is bit-identical to the same frame rendered alone (tests/test_gpu_parity.py). Limits: the per-sample
    # partial sums of a call (24 B x spp x pixels x fpc) stay under 100 GB, and fpc <= 4.
    frames_per_step = world if args.scaling == "weak" else 1                 # over all GPUs
    rows_per_gpu_step = H * frames_per_step // world
    per_sample = rows_per_gpu_step * W * 24.0                                # bytes of partial sums per sample index and step on one GPU
    if args.frames_per_call > 0:
        fpc = args.frames_per_call
    else:
        fpc = int(max(1, min(4, max(args.steps, 1), 100e9 // max(spp * per_sample, 1.0))))
    if args.spp_chunk < 0:
        args.spp_chunk = max(1, int(-(-spp * per_sample * fpc // 100e9)))

    # Counter passes first: child processes, started before this process has touched the GPU.
    pmc, pmc_note = (None, "N > 1" if world > 1 else "--no-pmc")
    if world == 1 and not args.no_pmc:
        pmc, pmc_note = run_pmc(args, spp, args.spp_chunk, args.pmc_seconds)

    import torch
    import torch.distributed as dist
    import raytracer_2022_amd as rt
    from raytracer_2022_amd import _ffi as F, film

    if not torch.cuda.is_available():
        print("bench.py: no GPU visible — the hot path has no CPU fallback", file=sys.stderr)
        sys.exit(3)
    # RT2022_BENCH_BACKEND=gloo: rehearsal of the N > 1 path on a box with fewer GPUs than ranks (ranks share the cards,
    # collectives go over gloo on host copies). Never a measurement: the line says so.
    backend = os.environ.get("RT2022_BENCH_BACKEND", "nccl")
    rehearsal = world > 1 and backend != "nccl"
    dev_index = local_rank % torch.cuda.device_count() if rehearsal else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group(backend, rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    coll = (lambda t: t.cpu()) if rehearsal else (lambda t: t)       # what a collective is handed

    assets = args.assets if os.path.isdir(args.assets) else None
    scene = rt.HostScene(scene_name, seed=args.seed, assets_dir=assets, param=SCENE_PARAM.get(args.config, 0))
    cam, bg = scene.default_view(W / H)
    dscene = rt.DeviceScene(scene.desc)
    if args.partial_ring:
        dscene.set_partial_ring(args.partial_ring)
    if args.groups or args.segs or args.quorum:
        dscene.set_tuning((args.quorum or 18) | (1 << 8) | (2 << 12) | ((args.segs or 8) << 16) | (2 << 20) | (args.groups << 24))
    stream = torch.cuda.current_stream().cuda_stream

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    class Call:
        """One rt_render_device call over `k` steps' worth of rows: its row list, buffers, parameters and (counter pass) its ray counts."""
        def __init__(self, k):
            self.k = k
            self.n_frames = frames_per_step * k
            self.params = rt.make_params(W, H, spp, 50, bg, seed=args.seed, n_frames=self.n_frames, spp_chunk=args.spp_chunk)
            self.params.flags |= F.RT_FLAG_KERNEL_TIMES
            self.plain = F.rt_params.from_buffer_copy(self.params)
            self.plain.flags &= ~F.RT_FLAG_KERNEL_TIMES
            rows = film.rank_rows(H, self.n_frames, args.seed, rank, world)
            self.n_rows = len(rows)
            self.d_rows = torch.from_numpy(rows.view(np.int32)).to(dev)
            self.d_out = torch.empty((self.n_rows, W, 3), dtype=torch.float64, device=dev)
            # (strong scaling deals the rows over `world` ranks: the shares differ by at most one row; gather needs equal shapes)
            max_rows = -(-H * self.n_frames // world)
            self.d_pad = torch.zeros((max_rows, W, 3), dtype=torch.float64, device=dev) if max_rows != self.n_rows else None
            self.gather_list = [torch.empty((max_rows, W, 3), dtype=torch.float64, device="cpu" if rehearsal else dev) for _ in range(world)] \
                if (world > 1 and rank == 0 and not args.no_gather) else None
            # counter pass (untimed, deterministic): rays / node visits / primitive tests of this call
            pc = F.rt_params.from_buffer_copy(self.params)
            pc.flags = F.RT_FLAG_COUNTERS
            st = F.rt_stats()
            dscene.render_device(cam, pc, self.d_rows.data_ptr(), self.n_rows, self.d_out.data_ptr(), stream, st)
            dscene.wait(stream)
            self.counts = st.as_dict()

        def run(self, stats=None, plain=False):
            dscene.render_device(cam, self.plain if plain else self.params, self.d_rows.data_ptr(), self.n_rows, self.d_out.data_ptr(), stream, stats)
            if world > 1 and not args.no_gather:
                src = self.d_out
                if self.d_pad is not None:
                    self.d_pad[:self.n_rows].copy_(self.d_out)
                    src = self.d_pad
                dist.gather(coll(src), self.gather_list, dst=0)

    def split(n_steps):          # n_steps as calls of fpc steps (+ one shorter call for the remainder)
        return [fpc] * (n_steps // fpc) + ([n_steps % fpc] if n_steps % fpc else [])
    calls = {}
    for k in sorted(set(split(args.steps) + split(args.warmup) + [min(fpc, max(args.steps, 1))]), reverse=True):
        calls[k] = Call(k)
    main_call = calls[min(fpc, max(args.steps, 1))]

    # The timed region is the PLAIN path — the calls a caller would make: no RT_FLAG_KERNEL_TIMES, the library's own choice of how many
    # groups of pool segments pass independently (two, on streams of their own, for every scene but the meshes: one group's shade pass
    # runs beside another's traversal pass). The per-kernel figures the roofline needs come from ONE more call of the same size made
    # right after it WITH the flag (HIP events around every pass on the launch stream; the flag keeps the pool in one group, so that
    # a launch's duration is its own): `instrumented_call` in the line.
    for k in split(args.warmup):
        calls[k].run(plain=True)
    barrier()
    timed_counts = None
    t0 = time.perf_counter()
    for k in split(args.steps):
        s = F.rt_stats()
        calls[k].run(s, plain=True)
        dscene.wait(stream)
    barrier()
    elapsed = time.perf_counter() - t0
    # what the timed region did, from the counter passes of its calls
    counts = None
    for k in split(args.steps):
        c = calls[k].counts
        if counts is None:
            counts = {key: (list(v) if isinstance(v, (list, tuple)) else v) for key, v in c.items()}
        else:
            for key, v in c.items():
                if isinstance(v, (list, tuple)):
                    counts[key] = [x + y for x, y in zip(counts[key], v)]
                elif isinstance(v, (int, float)):
                    counts[key] = counts[key] + v
    if counts is None:
        counts = dict(main_call.counts)
    n_steps_counted = max(args.steps, 1) if args.steps else main_call.k

    # The instrumented call (see above): per-kernel device time and pass pairs of main_call.k steps.
    n_plain = 0 if not args.steps else main_call.k
    elapsed_plain = 0.0
    kernel_ms, trace_ms, shade_ms, passes = 0.0, 0.0, 0.0, 0
    if n_plain:
        barrier()
        t0 = time.perf_counter()
        s = F.rt_stats()
        main_call.run(s, plain=False)
        dscene.wait(stream)          # fills s.ms / s.trace_ms / s.shade_ms from the HIP events on `stream`
        barrier()
        elapsed_plain = time.perf_counter() - t0
        kernel_ms, trace_ms, shade_ms, passes = s.ms, s.trace_ms, s.shade_ms, s.passes

    t = torch.tensor([elapsed, elapsed_plain], dtype=torch.float64, device=dev)
    cnt = torch.tensor([counts["rays"], counts["paths"], main_call.counts["rays"]], dtype=torch.float64, device=dev)
    if world > 1:
        t, cnt = coll(t), coll(cnt)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
    elapsed, elapsed_plain = float(t[0].item()), float(t[1].item())
    total_rays, total_paths, plain_rays = float(cnt[0].item()), float(cnt[1].item()), float(cnt[2].item())     # of the whole timed region, all ranks

    if rank == 0:
        ms_per_step = elapsed / max(args.steps, 1) * 1e3
        value = total_rays / elapsed / 1e6 if args.steps else 0.0
        # per STEP figures of rank 0 (the roofline is about one GPU's kernels): totals of the timed region / steps
        for key, v in list(counts.items()):
            counts[key] = [x / n_steps_counted for x in v] if isinstance(v, list) else (v / n_steps_counted if isinstance(v, (int, float)) else v)
        # (per-kernel figures: of the instrumented call, per step of it)
        k_ms = kernel_ms / n_plain if n_plain else float("nan")
        tr_ms = trace_ms / n_plain if n_plain else float("nan")
        sh_ms = shade_ms / n_plain if n_plain else float("nan")
        n_pass = passes / n_plain if n_plain else 0          # pass pairs per step (a call of fpc steps shares its passes among them)
        n_rows = main_call.n_rows // main_call.k
        ab = algorithmic_bytes(counts, n_rows * W)
        gbs = lambda nbytes, ms: nbytes / (ms * 1e-3) / 1e9 if ms and ms > 0 else None
        rnd = lambda x, n=2: None if x is None else round(x, n)
        frac = lambda x, peak: None if x is None else round(x / peak, 4)
        achieved = gbs(ab["wf_trace"], tr_ms)
        sbytes = scene_bytes(scene.desc)
        # Which traversal variant ran: with the node table in LDS the node bytes never reach L1 / L2 (primitives still do).
        tv = dscene.trace_variant()
        nodes_in_lds = tv["nodes_in_lds"] >= scene.desc.n_nodes and tv["nodes_in_lds"] > 0
        # (the sphere-only instance tests node boxes in single precision: 3 x 8 B of box + 8 B of child refs per visit)
        node_lds = 32 if tv.get("f32_slabs") else BYTES_NODE_LDS
        lds_bytes = node_lds * counts["node_visits"] if nodes_in_lds else 0
        cache_bytes = ab["traversal"] - (BYTES_NODE * counts["node_visits"] if nodes_in_lds else 0)
        if tv.get("spheres_in_lds"):       # (small sphere-only scenes: both sphere pools are LDS tables too, 36 / 80 B per test)
            k_s, k_m = F.KIND_NAMES.index("sphere"), F.KIND_NAMES.index("moving_sphere")
            lds_bytes += 36 * counts["prim_tests"][k_s] + 80 * counts["prim_tests"][k_m]
            cache_bytes -= BYTES_PRIM["sphere"] * counts["prim_tests"][k_s] + BYTES_PRIM["moving_sphere"] * counts["prim_tests"][k_m]

        l2_bytes = cache_bytes - ((BYTES_NODE - 32) * counts["node_visits"] if (tv.get("f32_slabs") and not nodes_in_lds) else 0)
        launch = lambda nbytes: int(nbytes / n_pass) if n_pass else None
        # SURVEY.md §8(d), the contract's figure: ALGORITHMIC bytes of the dominant kernel / its device time / 8 TB/s. It prices every
        # node and primitive byte against HBM although the node table (and, for small sphere scenes, the primitives) is read from LDS
        # and the rest from L1 / L2 — on a cache-resident scene it exceeds 1 and says nothing physical. It is
        # kept, under its own name; the top-level bound / achieved / frac are what the counters of this run MEASURED.
        sec8d = {
            "bound": "hbm", "achieved": rnd(achieved), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": frac(achieved, HBM_PEAK_GBS),
            "algorithmic_bytes_per_launch": launch(ab["wf_trace"]), "algorithmic_bytes_per_step": {k: int(v) for k, v in ab.items()},
            "served_from_lds_per_launch": launch(ab["wf_trace"] - cache_bytes - BYTES_RAY_READ * counts["rays"]),
            "achieved_without_lds_served": rnd(gbs(cache_bytes + BYTES_RAY_READ * counts["rays"], tr_ms)),
            "frac_without_lds_served": frac(gbs(cache_bytes + BYTES_RAY_READ * counts["rays"], tr_ms), HBM_PEAK_GBS),
            "whole_frame": {"achieved": rnd(gbs(ab["total"], k_ms)), "frac": frac(gbs(ab["total"], k_ms), HBM_PEAK_GBS)},
            "traversal_only": {"bytes": int(ab["traversal"]), "achieved": rnd(gbs(ab["traversal"], tr_ms)), "frac": frac(gbs(ab["traversal"], tr_ms), HBM_PEAK_GBS)},
            "note": "algorithmic bytes (64 B per node visit, the record size per primitive test, 128 B per ray) over the kernel's device time; a frac above 1 "
                    "prices bytes that LDS / L1 / L2 served against HBM bandwidth — nominal, not a measurement of the memory system",
        }
        roof = {
            "bound": None, "achieved": None, "peak": None, "unit": None, "frac": None,
            "kernel": "wf_trace (BVH traversal + Hittable::hit, pt_wavefront.hip): %.0f %% of the frame's device time" % (100.0 * tr_ms / k_ms if k_ms else 0),
            "traffic": None, "traffic_unit": "HBM bytes per wf_trace launch (FETCH_SIZE x 2 + WRITE_SIZE, KiB -> bytes)",
            "launches_per_step": n_pass, "launch_ms": rnd(tr_ms / n_pass if n_pass else None, 4),
            "device_ms_per_step": {"all": rnd(k_ms, 3), "wf_trace": rnd(tr_ms, 3), "wf_shade": rnd(sh_ms, 3)},
            "contract_sec8d": sec8d,
            # the same traversal bytes against the level that really serves them when the scene fits in cache
            # (a sphere scene too large for LDS fetches 32-byte single-precision node records: l2_bytes counts those, §8(d)'s figures keep their 64 B)
            "l2": {"achieved": rnd(gbs(l2_bytes, tr_ms)), "peak": L2_PEAK_GBS, "unit": "GB/s",
                   "frac": frac(gbs(l2_bytes, tr_ms), L2_PEAK_GBS), "bytes_per_step": int(l2_bytes),
                   "what": "primitive records (the node records are read from LDS)" if nodes_in_lds else
                           "32-byte single-precision node records and primitive records" if tv.get("f32_slabs") else "node and primitive records",
                   "scene_bytes": int(sbytes), "fits_aggregate_l2": bool(sbytes <= L2_BYTES)},
            "trace_variant": tv,
            "pmc": pmc_note,
        }
        limiter = []
        if pmc and "wf_trace" in pmc:
            c = pmc["wf_trace"]
            rd, wr = hbm_bytes(c)
            launches = max(1, int(c.get("_dispatches", n_pass or 1)))
            if rd + wr > 0:
                roof["traffic"] = int((rd + wr) / launches)
                roof["traffic_per_step"] = {"wf_trace": int(rd + wr), "read": int(rd), "write": int(wr)}
                if "wf_shade" in pmc:
                    r2, w2 = hbm_bytes(pmc["wf_shade"])
                    roof["traffic_per_step"]["wf_shade"] = int(r2 + w2)
                    roof["traffic_per_step"]["wf_shade_read"] = int(r2)
                    roof["traffic_per_step"]["wf_shade_write"] = int(w2)
                # Calibration of this access pattern (profiles/r2_traffic_calibration.txt, tools/traffic_calib.sh): a lane's
                # 64-byte ray fetch from a random 128-byte record is one 64-byte request and FETCH_SIZE reads it exactly
                # (ratio 1.000); only whole-line requests are counted at half. wf_trace's reads are of the first kind, so
                # the prescribed x2 doubles them: the calibrated figure takes FETCH_SIZE as it is.
                roof["traffic_calibrated"] = {"wf_trace_per_launch": int((rd / 2.0 + wr) / launches), "wf_trace_per_step": int(rd / 2.0 + wr),
                                              "bytes_per_ray": rnd((rd / 2.0 + wr) / max(counts["rays"], 1), 1),
                                              "note": "reads = FETCH_SIZE x 1 (64-byte requests, measured ratio 1.000), writes = WRITE_SIZE (ratio 1.000); "
                                                      "`traffic` above keeps the guide's x2 and is an upper bound"}
                hb = gbs(rd + wr, tr_ms)
                roof["hbm_counter"] = {"achieved": rnd(hb), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": frac(hb, HBM_PEAK_GBS),
                                       "bytes_per_ray": rnd((rd + wr) / max(counts["rays"], 1), 1),
                                       "over_algorithmic": rnd((rd + wr) / max(ab["wf_trace"], 1), 3),
                                       "l2_hit_rate": rnd(c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), 4) if c.get("TCC_HIT_sum") else None}
                limiter.append(("hbm", roof["hbm_counter"]["frac"]))
            vb = valu_busy(c)
            if vb:
                vb["what"] = ("share of the SIMDs' vector issue slots (quad-cycles) in which wf_trace issued: (SQ_INSTS_VALU - SQ_ACTIVE_INST_VALU2) / "
                              "(1024 x SQ_BUSY_CYCLES / 32 / 4); `useful` = busy x lane utilisation; normalisation checked on saturating kernels, "
                              "profiles/r3_valu_calibration.txt")
                roof["valu"] = vb
                limiter.append(("valu", vb["busy"]))
            if "wf_shade" in pmc and pmc["wf_shade"].get("SQ_INSTS_VALU"):        # the other kernel of the frame, same figures
                c2 = pmc["wf_shade"]
                r2, w2 = hbm_bytes(c2)
                roof["wf_shade"] = {
                    "device_ms_per_step": rnd(sh_ms, 3), "launch_ms": rnd(sh_ms / n_pass if n_pass else None, 4),
                    "algorithmic_achieved": rnd(gbs(ab["wf_shade"], sh_ms)), "hbm_counter_achieved": rnd(gbs(r2 + w2, sh_ms)), "unit": "GB/s",
                    "hbm_counter_frac": frac(gbs(r2 + w2, sh_ms), HBM_PEAK_GBS), "valu": valu_busy(c2),
                }
        elif world == 1:
            tc = committed_traffic(args.config, spp, args.spp_chunk)
            if tc:
                t_, src = tc
                tb = t_.get("by_kernel", {}).get("wf_trace")
                if tb and n_pass:
                    roof["traffic"] = int((tb["read"] + tb["write"]) / n_pass)
                    roof["traffic_source"] = "committed profile %s (not measured by this run: %s)" % (src, pmc_note)
        if roof["l2"]["frac"] is not None and roof["l2"]["fits_aggregate_l2"]:
            limiter.append(("l2", roof["l2"]["frac"]))
        if nodes_in_lds:
            roof["lds"] = {"achieved": rnd(gbs(lds_bytes, tr_ms)), "peak": round(LDS_PEAK_GBS, 1), "unit": "GB/s", "frac": frac(gbs(lds_bytes, tr_ms), LDS_PEAK_GBS),
                           "bytes_per_step": int(lds_bytes),
                           "what": "%d B per node visit from the workgroup's node table (%s at random records: bank conflicts "
                                   "make the usable rate a third to a quarter of the conflict-free peak)%s"
                                   % (node_lds, "4 x ds_read2_b32: float boxes, single-precision slab test" if tv.get("f32_slabs") else "3 x ds_read_b128 + ds_read_b64",
                                      "; the sphere pools too" if tv.get("spheres_in_lds") else "")}
            limiter.append(("lds", roof["lds"]["frac"]))
        measured = pmc and "wf_trace" in pmc and roof.get("valu") and roof.get("hbm_counter")
        if measured:
            # bound = the resource the counters of THIS run show closest to its ceiling, and achieved / peak / frac are that resource's
            limiter.sort(key=lambda kv: -(kv[1] or 0))
            top = limiter[0][0]
            if top == "valu":
                roof.update(bound="valu", achieved=roof["valu"]["busy"], peak=1.0, unit="share of the vector issue slots (SIMD quad-cycles), measured", frac=roof["valu"]["busy"])
            else:
                src = roof["hbm_counter"] if top == "hbm" else roof[top]
                roof.update(bound=top, achieved=src["achieved"], peak=src["peak"], unit=src["unit"], frac=src["frac"])
            waits = roof["valu"].get("wait_any_over_wave_cycles") or 0.0
            regime = ("issue-bound: the vector pipes issue in %.0f %% of their slots" % (100 * roof["valu"]["busy"]) if roof["valu"]["busy"] >= 0.6
                      else "latency-bound: no resource is above %.0f %% of its ceiling and the waves spend %.0f %% of their cycles at s_waitcnt "
                           "(dependent node / primitive fetches served by %s)" % (100 * (limiter[0][1] or 0), 100 * waits, "L2 and beyond" if not nodes_in_lds else "LDS, L1 and L2")
                      if waits >= 0.5 else "mixed: vector issue %.0f %%, waits %.0f %% of wave cycles" % (100 * roof["valu"]["busy"], 100 * waits))
            roof["limiter"] = {"resource": top, "frac": limiter[0][1], "ranked": limiter, "regime": regime,
                               "note": "bound = this ranking's first entry; hbm and valu are counter measurements, l2 and lds are the algorithmic bytes those levels serve over their peaks"}
        else:
            # no counters in this run (N > 1, --no-pmc, or a pass failed): nothing measured to name a bound with. The contract's figure
            # stands in, with the LDS-served bytes taken out; see contract_sec8d for the nominal one.
            roof.update(bound="hbm", achieved=sec8d["achieved_without_lds_served"], peak=HBM_PEAK_GBS, unit="GB/s", frac=sec8d["frac_without_lds_served"])
            roof["limiter"] = {"resource": None, "note": "not measured in this run (%s): top-level figures are SURVEY §8(d)'s algorithmic bytes without the LDS-served ones — "
                                                         "cache-served bytes are still priced against HBM" % pmc_note}
        note = ["scene %.2f MB (%s the 32 MiB of aggregate L2)" % (sbytes / 1e6, "fits" if sbytes <= L2_BYTES else "exceeds")]
        if tv.get("f32_slabs") and not nodes_in_lds:
            note.append("traversal variant: node boxes tested in single precision on 32-byte records from L2 / HBM%s (double-precision record for the steps the two tests could "
                        "differ on) — half of the 64 B per node visit that SURVEY §8(d)'s figures keep counting"
                        % (", the first %d of them in LDS" % tv["nodes_in_lds"] if tv["nodes_in_lds"] else ""))
        if nodes_in_lds:
            note.append("traversal variant: %d-thread workgroups, all %d node records in LDS%s" % (tv["workgroup_threads"], tv["nodes_in_lds"],
                        " as floats (single-precision slab test, double-precision second opinion where the two could differ)" if tv.get("f32_slabs") else ""))
        if roof.get("hbm_counter"):
            note.append("measured HBM traffic of wf_trace %.0f B per ray segment = %.2f of the 8 TB/s peak, against SURVEY §8(d)'s nominal %.2f: "
                        "the node and primitive bytes are served by %s" % (roof["hbm_counter"]["bytes_per_ray"], roof["hbm_counter"]["frac"], sec8d["frac"] or 0,
                                                                          "LDS and the caches" if nodes_in_lds else "the caches"))
        if roof.get("valu"):
            note.append("wf_trace's vector pipes issue in %.0f %% of their slots (measured) at %.0f %% lane utilisation; its waves wait %.0f %% of their cycles at s_waitcnt and %.0f %% for an issue slot"
                        % (100 * roof["valu"]["busy"], 100 * (roof["valu"]["lane_utilisation"] or 0), 100 * (roof["valu"]["wait_any_over_wave_cycles"] or 0),
                           100 * (roof["valu"]["issue_stall_over_wave_cycles"] or 0)))
        roof["note"] = "; ".join(note)

        out = {
            "metric": "Mrays/s (primary+secondary), book-2 final scene 800x800x1000spp" if args.config == "c3" and args.spp == 0
                      else "Mrays/s (primary+secondary)",
            "value": round(value, 2), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "ms_per_frame": round(ms_per_step, 3), "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic" if not rehearsal else
            "synthetic — REHEARSAL over %s with ranks sharing GPUs: not a measurement" % backend,
            "config": {"workload": desc_text if args.spp == 0 else desc_text + " [spp overridden to %d]" % spp,
                       "scene": scene_name, "width": W, "height": H, "spp": spp, "max_depth": 50,
                       "frames": frames_per_step, "rows_per_gpu": n_rows, "spp_chunk": int(s.spp_chunk) if args.steps else args.spp_chunk,
                       "pool_slots": int(s.pool_slots) if args.steps else None, "seed": args.seed,
                       "partial_sum_bytes_per_call": int(s.partial_bytes) if args.steps else None,
                       "steps_per_call": fpc, "calls": split(args.steps),
                       "steps_per_call_note": "a step is one frame per GPU; the library gets them %d at a time (one rt_render_device call = a %d-frame strip, frames keyed "
                                              "0..%d): its path pool drains once per call instead of once per frame; every frame's pixels are those of the frame rendered alone"
                                              % (fpc, fpc * frames_per_step, fpc * frames_per_step - 1),
                       "call_latency_ms": round(elapsed / max(len(split(args.steps)), 1) * 1e3, 3) if args.steps else None,
                       "sharding": ("rows of an N-frame strip dealt cyclically (one frame's worth per GPU)" if args.scaling == "weak"
                                    else "rows of ONE frame dealt cyclically over the GPUs") + "; gather of row buffers to rank 0",
                       "assets": ("assets/ (the reference's earthmap / planet JPEGs and Shuttle.obj)" if assets and os.path.exists(os.path.join(assets, "earthmap.jpg"))
                                  else "procedural stand-ins (no assets directory)"),
                       "triangles": int(scene.desc.n_triangles), "bvh_nodes": int(scene.desc.n_nodes)},
            "rays_per_step": int(round(total_rays / max(args.steps, 1))), "paths_per_step": int(round(total_paths / max(args.steps, 1))),
            "instrumented_call": ({"value": round(plain_rays / elapsed_plain / 1e6, 2), "unit": "Mrays/s", "steps": n_plain,
                                   "ms_per_step": round(elapsed_plain / n_plain * 1e3, 3),
                                   "note": "one more call of the same size WITH RT_FLAG_KERNEL_TIMES, run after the timed region: HIP events around every pass, the pool in one "
                                           "group (a launch's duration is then its own) — where roofline.device_ms_per_step / launch_ms / launches_per_step come from; "
                                           "`value` above is the timed region's: the plain calls, no events, the library's own choice of groups"}
                                  if n_plain and elapsed_plain > 0 else None),
            "roofline": roof,
            "counters_rank0": counts,
        }
        if not args.no_cpu_baseline and world == 1:
            try:
                p1 = rt.make_params(W, H, spp, 50, bg, seed=args.seed)
                out["cpu_baseline"] = cpu_baseline(scene, cam, p1, H, args.seed, args.cpu_seconds)
                if args.config == "c1":
                    out["cpu_baseline"]["full_frame"] = cpu_full_frame(scene, cam, p1, H, args.seed)
            except Exception as e:  # the oracle is a checker, never the product: report and go on
                out["cpu_baseline"] = {"value": None, "unit": "Mrays/s", "cores": usable_cores(), "kind": "port",
                                       "sample": "failed: %r" % (e,)}
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
