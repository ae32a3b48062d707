"""Step time of the multi-GPU protocol with one rank over RCCL (every exchange forced): a lower
bound of the per-step protocol overhead (host syncs, torch op launches, collective launches)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29688")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1"); os.environ.setdefault("LOCAL_RANK", "0")
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
from collision_amd import hip
from collision_amd.multi import DistributedCollider
import bench
n = 1000000
coords, radii = bench.uniform_scene(n)
configs = (("morton", False), ("morton", True), ("hash", True))
if len(sys.argv) > 1:
    configs = ((sys.argv[1], True),)
for partition, exercise in configs:
    dc = DistributedCollider(hip.Context(0), dist, n, pair_capacity=1 << 19, partition=partition, exercise_single_rank=exercise)
    dc.set_local_spheres(coords, radii, np.arange(n, dtype=np.uint32))
    for _ in range(5):
        dc.step()
    dc.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        dc.step()
    dc.synchronize()
    ms = (time.perf_counter() - t0) / 30 * 1e3
    print("partition=%s exchanges=%s: %.3f ms/step, %d pairs" % (partition, exercise, ms, dc.local_pair_count()))
dist.destroy_process_group()
