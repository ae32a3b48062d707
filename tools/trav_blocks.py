"""Traversal time against the leaf-block criterion k (col_debug_leaf_blocks): uniform config 2 and clustered config 3."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from collision_amd import hip
from collision_amd._lib import call, cdll
from collision_amd.collision import Collider
import bench
lib = cdll(); lib.col_debug_leaf_blocks.argtypes = [ctypes.c_float]
ctx = hip.Context(); cq = hip.CommandQueue(ctx)
n = 1000000
for name, (coords, radii), cap in (("uniform", bench.uniform_scene(n), 1 << 17), ("config3", bench.clustered_scene(n, 0.0152), 1 << 25)):
    cb, rb = hip.Buffer(ctx, hostbuf=coords), hip.Buffer(ctx, hostbuf=radii)
    nb, pb = hip.Buffer(ctx, 4), hip.Buffer(ctx, cap * 8)
    col = Collider(ctx, n, 64, 256)
    col.sort_plan = "lsd" if name == "config3" else "auto"
    for k, variant in ((3.0, 0), (3.0, 1024), (3.0, 0), (3.0, 1024)):
        lib.col_debug_leaf_blocks(ctypes.c_float(k))
        lib.col_debug_traverse(variant)
        def step():
            col.get_collisions(cq, cb, rb, nb, pb, cap)
        for _ in range(3): step()
        cq.finish()
        whole = bench.time_events(hip, cq, step, 10)
        def trav():
            call.col_traverse(cq.stream, pb.ptr, nb.ptr, cap, col._nodes_buf.ptr, col._bounds_buf.ptr, n, 4)
        trav(); cq.finish()
        t = bench.time_events(hip, cq, trav, 10)
        print("%-8s k=%-6g variant %3d whole %.4f ms  traverse %.4f ms  pairs %d" % (name, k, variant, whole, t, int(hip.read_buffer(cq, nb, np.uint32, 1)[0]) ))
lib.col_debug_leaf_blocks(ctypes.c_float(3.0)); lib.col_debug_traverse(0)
