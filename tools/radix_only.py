"""Runs only the config-5 radix microbench (64 Mi u32 keys + u32 ids), for rocprofv3 passes."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from collision_amd import hip

ctx = hip.Context()
cq = hip.CommandQueue(ctx)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
print(json.dumps(bench.radix_microbench(hip, ctx, cq, reps=reps)))
