"""A/B of the 1M path: eager stream launches vs one captured hipGraph replayed (torch.cuda.graph)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from collision_amd import hip
from collision_amd.collision import Collider
import bench

ctx = hip.Context()
n, cap = bench.N_SPHERES, bench.PAIR_CAPACITY
coords, radii = bench.uniform_scene(n)
c_t = torch.from_numpy(coords).cuda(); r_t = torch.from_numpy(radii).cuda()
n_t = torch.zeros(1, dtype=torch.int32, device="cuda")
p_t = torch.zeros((cap, 2), dtype=torch.int32, device="cuda")
bufs = [hip.Buffer.from_tensor(ctx, t) for t in (c_t, r_t, n_t, p_t)]
col = Collider(ctx, n, bench.NGROUPS if hasattr(bench, "NGROUPS") else 64, bench.GROUP_SIZE)
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    cq = hip.CommandQueue(ctx, stream=side.cuda_stream)
    step = lambda: col.get_collisions(cq, bufs[0], bufs[1], bufs[2], bufs[3], cap)
    for _ in range(5):
        step()
    side.synchronize()
    expect = int(n_t.item())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        step()
    graph.replay(); side.synchronize()
    assert int(n_t.item()) == expect

    def timed(fn, k=200):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(20):
            fn()
        side.synchronize()
        t0 = time.perf_counter()
        a.record(side)
        for _ in range(k):
            fn()
        b.record(side)
        side.synchronize()
        return a.elapsed_time(b) / k, (time.perf_counter() - t0) * 1e3 / k

    for rnd in range(3):
        print("round %d eager  : %.4f ms device, %.4f ms wall" % ((rnd,) + timed(step)))
        print("round %d replay : %.4f ms device, %.4f ms wall" % ((rnd,) + timed(graph.replay)))
print("pairs", expect)
