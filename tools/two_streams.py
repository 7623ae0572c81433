"""Do two calls in flight (RT_FLAG_ASYNC, two streams of one scene, each with its own pool) fill each other's drain tails?

    python3 tools/two_streams.py [scene W H spp frames_per_call]

Renders 2 x F frames of the scene (a) as two consecutive synchronous calls of F frames on one stream and (b) as two asynchronous
calls of F frames on two streams at once; prints wall time and Mrays/s of both (rays from one counter pass), twice over.
"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import raytracer_2022_amd as rt
from raytracer_2022_amd import _ffi as F

name = sys.argv[1] if len(sys.argv) > 1 else "final_scene"
W = int(sys.argv[2]) if len(sys.argv) > 2 else 800
H = int(sys.argv[3]) if len(sys.argv) > 3 else 800
spp = int(sys.argv[4]) if len(sys.argv) > 4 else 1000
fpc = int(sys.argv[5]) if len(sys.argv) > 5 else 2
assets = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "assets")
s = rt.HostScene(name, seed=2022, assets_dir=assets if os.path.isdir(assets) else None)
cam, bg = s.default_view(W / H)
dev = rt.DeviceScene(s.desc)
rows = np.concatenate([rt.shuffled_rows(H, 2022) + f * H for f in range(fpc)]).astype(np.uint32)
d_rows = torch.from_numpy(rows.view(np.int32)).cuda()
outs = [torch.empty((len(rows), W, 3), dtype=torch.float64, device="cuda") for _ in range(2)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
p = rt.make_params(W, H, spp, 50, bg, seed=2022, spp_chunk=1)
p.n_frames = fpc
# rays of one call (counter pass at a few samples, scaled: rays per sample are what they are)
pc = rt.make_params(W, H, 4, 50, bg, seed=2022, spp_chunk=1)
pc.n_frames = fpc
pc.flags |= F.RT_FLAG_COUNTERS
st = F.rt_stats()
dev.render_device(cam, pc, d_rows.data_ptr(), len(rows), outs[0].data_ptr(), streams[0].cuda_stream, st)
dev.wait(streams[0].cuda_stream)
rays_per_call = st.rays / 4 * spp
for s_ in streams:                                             # warm both workspaces (pool allocation)
    dev.render_device(cam, pc, d_rows.data_ptr(), len(rows), outs[0].data_ptr(), s_.cuda_stream, None)
    dev.wait(s_.cuda_stream)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(2):
        dev.render_device(cam, p, d_rows.data_ptr(), len(rows), outs[k].data_ptr(), streams[0].cuda_stream, None)
        dev.wait(streams[0].cuda_stream)
    torch.cuda.synchronize(); t_seq = time.perf_counter() - t0
    a = [o.clone() for o in outs]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(2):
        dev.render_device(cam, p, d_rows.data_ptr(), len(rows), outs[k].data_ptr(), streams[k].cuda_stream, None, asynchronous=True)
    for k in range(2):
        dev.wait(streams[k].cuda_stream)
    torch.cuda.synchronize(); t_par = time.perf_counter() - t0
    same = all(torch.equal(x.view(torch.int64), y.view(torch.int64)) for x, y in zip(a, outs))
    print("%s %dx%dx%d, %d frames per call, 2 calls: one after the other %.1f ms (%.0f Mrays/s), both in flight %.1f ms (%.0f Mrays/s): %+.1f %%, same bits: %s"
          % (name, W, H, spp, fpc, t_seq * 1e3, 2 * rays_per_call / t_seq / 1e6, t_par * 1e3, 2 * rays_per_call / t_par / 1e6, (t_seq / t_par - 1) * 100, same), flush=True)
