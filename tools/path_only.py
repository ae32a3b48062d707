"""Runs only the 1 M-sphere path (config 2) a few times, for rocprofv3 passes."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from collision_amd import hip
from collision_amd.collision import Collider

ctx = hip.Context()
cq = hip.CommandQueue(ctx)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
coords, radii = bench.uniform_scene(bench.N_SPHERES)
cb, rb = hip.Buffer(ctx, hostbuf=coords), hip.Buffer(ctx, hostbuf=radii)
nb, pb = hip.Buffer(ctx, 4), hip.Buffer(ctx, bench.PAIR_CAPACITY * 8)
col = Collider(ctx, bench.N_SPHERES, bench.NGROUPS, bench.GROUP_SIZE)
for _ in range(steps):
    col.get_collisions(cq, cb, rb, nb, pb, bench.PAIR_CAPACITY)
cq.finish()
print("pairs", int(hip.read_buffer(cq, nb, "uint32", 1)[0]))
