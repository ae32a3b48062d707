"""Analysis only: how many steps the packet walk (csrc/bvh.hip phase 2) takes per packet on BASELINE config 2 / 3
scenes, and how many it would take with leaf blocks of at most B leaves.  Uses the oracle's tree.
    python tests/analysis/sim_packet_walk.py [uniform|config3] [n]"""
import ctypes as C
import subprocess
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import bench
import oracle

here = Path(__file__).resolve().parent
so = here / "_sim_walk.so"
subprocess.run(["gcc", "-O2", "-shared", "-fPIC", "-o", str(so), str(here / "sim_walk.c")], check=True)
lib = C.CDLL(str(so))
kind = sys.argv[1] if len(sys.argv) > 1 else "uniform"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
coords, radii = bench.clustered_scene(n, 0.0152) if kind == "config3" else bench.uniform_scene(n)
res = oracle.collide(coords, radii, capacity=0, want=True)
nodes, bnds = np.ascontiguousarray(res["nodes"]), np.ascontiguousarray(res["bounds"], dtype=np.float32)
print(kind, n, "pairs", res["count"])
for B in (0, 8, 16, 32, 64):
    out = (C.c_uint64 * 8)()
    lib.sim_walk(nodes.ctypes.data_as(C.c_void_p), bnds.ctypes.data_as(C.c_void_p), C.c_uint32(n), C.c_int(B), out)
    np_ = (n + 63) // 64
    print("B=%2d: steps/packet %.1f descents %.1f leaf steps %.1f blocks %.1f block leaves %.1f" %
          (B, out[0] / np_, out[1] / np_, out[2] / np_, out[4] / np_, out[5] / np_))

for B in (0, 16):
    for K in (1, 2, 3, 4):
        out = (C.c_uint64 * 8)()
        lib.sim_walk_ctx(nodes.ctypes.data_as(C.c_void_p), bnds.ctypes.data_as(C.c_void_p), C.c_uint32(n), C.c_int(B), C.c_int(K), out)
        np_ = (n + 63) // 64
        print("contexts %d B=%2d: rounds/packet %.1f fetches %.1f single rounds %.1f" % (K, B, out[0] / np_, out[1] / np_, out[2] / np_))
