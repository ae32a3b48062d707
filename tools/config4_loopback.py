"""BASELINE config 4 (world 8 x 2 M spheres, hash arrival) on ONE GPU: eight HIP engines in one process driven in
lockstep (collision_amd.multi.LoopbackWorld).  Prints, per rank: owned spheres, ghost queries, slot sizes and the
device time of every protocol segment (the kernels between two collectives, run to completion rank by rank), i.e.
the per-rank critical path of a step without the wire.  JSON on the last line.

    python tools/config4_loopback.py [morton|hash] [per_rank] [world]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from collision_amd import hip
from collision_amd.multi import LoopbackWorld, hash_owner

partition = sys.argv[1] if len(sys.argv) > 1 else "morton"
per = int(sys.argv[2]) if len(sys.argv) > 2 else 2000000
world = int(sys.argv[3]) if len(sys.argv) > 3 else 8
n = per * world
rng = np.random.RandomState(4)
coords = np.zeros((n, 4), np.float32)
coords[:, :3] = rng.random_sample((n, 3))
radii = np.full(n, 0.001 * (1e6 / n) ** (1.0 / 3.0), np.float32)
gids = np.arange(n, dtype=np.uint32)
owner = hash_owner(gids, world)
ctx = hip.Context(0)
lw = LoopbackWorld(ctx, world, [int((owner == r).sum()) for r in range(world)], group_size=256, pair_capacity=1 << 20,
                   partition=partition)
for r in range(world):
    mine = owner == r
    lw.set_local_spheres(r, coords[mine], radii[mine], gids[mine])
lw.step(); lw.synchronize()           # slots adapt
lw.step(); lw.synchronize()
runs = []
for _ in range(5):
    lw.step(timed=True)
    runs.append([[ms for _, ms in segs] for segs in lw.segments])
lw.synchronize()
labels = [name for name, _ in lw.segments[0]]
med = np.median(np.array(runs), axis=0)            # [rank, segment]
total_pairs = lw.global_pair_count()
print("%s partition, %d x %d spheres, %d pairs" % (partition, world, per, total_pairs))
print("segments end with: " + ", ".join(labels))
rows = []
for r, dc in enumerate(lw.ranks):
    rows.append({"rank": r, "owned": dc.n_owned, "ghosts": dc.stats.get("ghosts"), "halo_slot": dc.slot,
                 "partition_slot": dc.part_slot, "peers_in": len(dc.peers_in), "peers_out": len(dc.peers_out),
                 "segment_ms": [round(float(v), 4) for v in med[r]], "sum_ms": round(float(med[r].sum()), 4)})
    print("rank %d: owned %8d  ghosts %8d  halo slot %8d  part slot %7d  peers in/out %d/%d  segments %s  sum %.3f ms"
          % (r, dc.n_owned, dc.stats.get("ghosts", 0), dc.slot, dc.part_slot, len(dc.peers_in), len(dc.peers_out),
             " ".join("%.3f" % v for v in med[r]), med[r].sum()))
# coherent arrival: what a rank owns is its next input (nothing moves)
lw.adopt_owned(); lw.step(); lw.synchronize(); lw.adopt_owned(); lw.step(); lw.synchronize()
runs = []
for _ in range(5):
    lw.adopt_owned()
    lw.step(timed=True)
    runs.append([[ms for _, ms in segs] for segs in lw.segments])
lw.synchronize()
medc = np.median(np.array(runs), axis=0)
print("coherent arrival (adopt_owned): per-rank sum " + " ".join("%.3f" % v for v in medc.sum(axis=1)) +
      "  partition slot %d" % lw.ranks[0].part_slot)
print(json.dumps({"partition": partition, "world": world, "per_rank": per, "pairs": total_pairs, "segment_labels": labels,
                  "ranks": rows, "critical_path_ms_max_over_ranks": round(float(med.sum(axis=1).max()), 4),
                  "coherent_arrival_sum_ms": [round(float(v), 4) for v in medc.sum(axis=1)],
                  "coherent_partition_slot": lw.ranks[0].part_slot}))
