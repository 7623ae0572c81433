"""N>1 path on CPU: two gloo ranks shard the rows of a 2-frame strip exactly as bench.py does
(film.rank_rows), render their share (here with the oracle standing in for the GPU, which this
container does not have), gather to rank 0 and reassemble. The result must equal one process
rendering everything — rows are independent and the RNG is keyed per pixel."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent('''
    import os, sys
    import numpy as np
    import torch, torch.distributed as dist
    sys.path.insert(0, %(root)r)
    import raytracer_2022_amd as rt
    from raytracer_2022_amd import film
    from oracle import oracle_ffi as O
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    W, H, spp, seed = 20, 12, 2, 2022
    s = rt.HostScene("cornell_box", seed=seed)
    cam, bg = s.default_view(W / H)
    p = rt.make_params(W, H, spp, 50, bg, seed=seed, n_frames=world)
    rows = film.rank_rows(H, world, seed, rank, world)
    mine = torch.from_numpy(O.render_cpu(s.desc, cam, p, rows, n_threads=1))
    parts = [torch.empty_like(mine) for _ in range(world)] if rank == 0 else None
    dist.gather(mine, parts, dst=0)
    if rank == 0:
        all_rows = [film.rank_rows(H, world, seed, r, world) for r in range(world)]
        strip = film.assemble([t.numpy() for t in parts], all_rows, H, world, W)
        np.save(sys.argv[1], strip)
    dist.barrier()
    dist.destroy_process_group()
''')


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_two_rank_strip_equals_single_process(tmp_path, rt, O):
    from raytracer_2022_amd import film
    world = 2
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    out = tmp_path / "strip.npy"
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script), str(out)], env=env))
    for pr in procs:
        assert pr.wait(timeout=300) == 0
    strip = np.load(out)
    W, H, spp, seed = 20, 12, 2, 2022
    s = rt.HostScene("cornell_box", seed=seed)
    cam, bg = s.default_view(W / H)
    p = rt.make_params(W, H, spp, 50, bg, seed=seed, n_frames=world)
    rows = np.arange(H * world, dtype=np.uint32)
    single = O.render_cpu(s.desc, cam, p, rows, n_threads=2).reshape(world, H, W, 3)
    assert not np.isnan(strip).all()
    assert np.array_equal(strip.view(np.uint64), single.view(np.uint64))
    # the two frames differ (different RNG key), the sharding covers every row once
    assert not np.array_equal(single[0], single[1])
    assert sorted(np.concatenate([film.rank_rows(H, world, seed, r, world) for r in range(world)])) == list(range(H * world))


GPU_WORKER = textwrap.dedent('''
    import os, sys
    import numpy as np
    import torch, torch.distributed as dist
    sys.path.insert(0, %(root)r)
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    ndev = torch.cuda.device_count()                  # (counting devices does not initialise the GPU)
    real = ndev >= world                              # one GPU per rank: RCCL; otherwise the ranks share GPU 0 and gather over gloo
    dev_id = rank if real else 0
    torch.cuda.set_device(dev_id)
    dist.init_process_group("nccl" if real else "gloo", rank=rank, world_size=world,
                            **({"device_id": torch.device("cuda", dev_id)} if real else {}))
    import raytracer_2022_amd as rt
    from raytracer_2022_amd import film
    scaling, W, H, spp, seed = sys.argv[2], 40, 24, 3, 2022
    n_frames = world if scaling == "weak" else 1
    s = rt.HostScene("final_scene", seed=seed)
    cam, bg = s.default_view(W / H)
    p = rt.make_params(W, H, spp, 50, bg, seed=seed, n_frames=n_frames, spp_chunk=1)
    rows = film.rank_rows(H, n_frames, seed, rank, world)
    scene = rt.DeviceScene(s.desc)                    # the HIP path, on this rank's device
    d_rows = torch.from_numpy(rows.view(np.int32)).cuda()
    d_out = torch.empty((len(rows), W, 3), dtype=torch.float64, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    scene.render_device(cam, p, d_rows.data_ptr(), len(rows), d_out.data_ptr(), stream)
    scene.wait(stream)
    mine = d_out if real else d_out.cpu()
    parts = [torch.empty_like(mine) for _ in range(world)] if rank == 0 else None
    dist.gather(mine, parts, dst=0)
    if rank == 0:
        all_rows = [film.rank_rows(H, n_frames, seed, r, world) for r in range(world)]
        strip = film.assemble([t.cpu().numpy() for t in parts], all_rows, H, n_frames, W)
        np.save(sys.argv[1], strip)
    dist.barrier()
    dist.destroy_process_group()
''')


import pytest


@pytest.mark.gpu
@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_two_ranks_through_the_hip_path(tmp_path, rt, O, scaling):
    """bench.py's N > 1 path with the kernels doing the rendering: two ranks, film.rank_rows shares (an N-frame strip for
    weak scaling, ONE frame's rows for strong), gather to rank 0, film.assemble — against one oracle render of everything.
    With two GPUs visible each rank has its own and the gather is RCCL; on a one-GPU box the ranks share the card and
    gather over gloo, so the first multi-GPU run of the driver is not the first execution of this code either way.
    The children are started before this process touches the GPU through the library."""
    from raytracer_2022_amd import film
    world = 2
    script = tmp_path / "gpu_worker.py"
    script.write_text(GPU_WORKER % {"root": ROOT})
    out = tmp_path / "strip.npy"
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, str(script), str(out), scaling], env=env))
    for pr in procs:
        assert pr.wait(timeout=600) == 0
    strip = np.load(out)
    W, H, spp, seed = 40, 24, 3, 2022
    n_frames = world if scaling == "weak" else 1
    s = rt.HostScene("final_scene", seed=seed)
    cam, bg = s.default_view(W / H)
    p = rt.make_params(W, H, spp, 50, bg, seed=seed, n_frames=n_frames, spp_chunk=1)
    rows = np.arange(H * n_frames, dtype=np.uint32)
    single = O.render_cpu(s.desc, cam, p, rows, n_threads=4).reshape(n_frames, H, W, 3)
    assert strip.shape == single.shape and not np.isnan(strip).any()
    assert np.array_equal(strip.view(np.uint64), single.view(np.uint64))
    shares = [film.rank_rows(H, n_frames, seed, r, world) for r in range(world)]
    assert sorted(np.concatenate(shares)) == list(range(H * n_frames)) and abs(len(shares[0]) - len(shares[1])) <= 1
