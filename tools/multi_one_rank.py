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
STEPS = 100
coords, radii = bench.uniform_scene(n)
configs = (("morton", False), ("morton", True), ("hash", True))
if len(sys.argv) > 1:
    configs = ((sys.argv[1], True),)
pads = []
for partition, exercise in configs:
    if os.environ.get("PAD_KB"):               # (experiment: shift every later allocation)
        pads.append(torch.empty(int(os.environ["PAD_KB"]) * 1024, dtype=torch.uint8, device="cuda"))
    dc = DistributedCollider(hip.Context(0), dist, n, pair_capacity=1 << 19, partition=partition, exercise_single_rank=exercise)
    dc.set_local_spheres(coords, radii, np.arange(n, dtype=np.uint32))
    for _ in range(int(os.environ.get("WARM", "150"))):     # (long enough for GPU and host clocks)
        dc.step()
    dc.synchronize()
    waited = [0.0]
    inner = dc.engine.owned_count
    def timed_owned_count():
        t = time.perf_counter()
        m = inner()
        waited[0] += time.perf_counter() - t
        return m
    dc.engine.owned_count = timed_owned_count
    t0 = time.perf_counter()
    for _ in range(STEPS):
        dc.step()
    host_ms = (time.perf_counter() - t0) / STEPS * 1e3       # host time in step(): enqueueing + the poll
    dc.synchronize()
    ms = (time.perf_counter() - t0) / STEPS * 1e3
    print("   host: %.3f ms/step in step(), of which %.3f ms polling for the owned count" % (host_ms, waited[0] / STEPS * 1e3))
    c = dc.engine.collider
    print("partition=%s exchanges=%s: %.3f ms/step, %d pairs   [repeats %d, slots %s/%s, lsd calls left %s, retry %s, oversize %s]" % (
        partition, exercise, ms, dc.local_pair_count(), dc.repeats, dc.part_slot, dc.slot, getattr(c, "_lsd_calls_left", None),
        getattr(c, "_retry_after", None), c.oversize_bucket))
dist.destroy_process_group()
