"""Multi-rank path (collision_amd/multi.py).

CPU: world_size 2 and 3 over gloo with a NumPy/oracle engine double -- the distributed protocol
(AABB all-gathers, splitters, repartition all-to-all, halo selection by the ownership rule, ghost
queries) must produce exactly the brute-force pair set, each pair once.
GPU: the same protocol with the real HIP engine, ranks sharing the one GPU of the test box.
"""
import json
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

from collision_amd.multi import handles, hash_owner

ROOT = Path(__file__).resolve().parent.parent


def test_ownership_rule_covers_every_pair_once():
    for world in range(1, 10):
        for r in range(world):
            assert not handles(r, r, world)
            for q in range(r + 1, world):
                assert handles(r, q, world) != handles(q, r, world)
        # balanced: every rank answers for about half of the others
        loads = [sum(handles(r, q, world) for q in range(world)) for r in range(world)]
        assert max(loads) - min(loads) <= 1


def test_hash_owner_is_balanced():
    own = hash_owner(np.arange(1 << 16, dtype=np.uint32), 8)
    counts = np.bincount(own.astype(np.int64), minlength=8)
    assert counts.min() > 0.9 * (1 << 13) and counts.max() < 1.1 * (1 << 13)


def _run(world, mode, partition, n, tmp_path, kind="uniform", port=29611, halo_slot=0, dtype="float32", part_slot=0,
         check="brute", adopt=""):
    out = tmp_path / ("result_%s_%s_%d.json" % (mode, partition, world))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(ROOT / "tests" / "dist_worker.py"),
           mode, partition, str(n), str(out), kind, str(halo_slot), dtype, str(part_slot), check, adopt]
    proc = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, proc.stdout[-2000:] + proc.stderr[-4000:]
    return json.loads(out.read_text())


@pytest.mark.parametrize("world,partition,kind", [(2, "morton", "uniform"), (2, "hash", "uniform"),
                                                  (3, "morton", "clustered")])
def test_gloo_cpu_world(tmp_path, world, partition, kind):
    res = _run(world, "cpu", partition, 1500, tmp_path, kind, port=29611 + world)
    assert res["ok"], res
    assert res["expected"] > 100


def test_gloo_cpu_world_float64(tmp_path):
    """f64 coordinates on the multi-rank path: 9-word transport records."""
    res = _run(3, "cpu", "morton", 1500, tmp_path, "clustered", port=29648, dtype="float64")
    assert res["ok"], res


def test_halo_slot_overflow_is_repaired(tmp_path):
    """The halo travels in fixed-size slots whose headers carry the list lengths (no count exchange, no host
    sync).  A slot that starts too small is seen at synchronize(): every rank grows it and the step is
    repeated, so the pairs read afterwards are still exactly the brute-force set."""
    res = _run(3, "cpu", "morton", 1500, tmp_path, "uniform", port=29641, halo_slot=8)
    assert res["ok"], res
    assert all(s["repeats"] >= 1 and s["halo_slot"] >= 4096 for s in res["stats"])


def test_partition_slot_overflow_loses_nothing(tmp_path):
    """The repartition travels in fixed-size slots too (one per other rank, the list length in the header, what a
    rank keeps does not travel).  What does not fit into a slot STAYS with the sending rank -- ownership decides
    load balance and halo size, never the pair set -- so the result is exact WITHOUT a repeat; synchronize() sees
    the headers and grows the slot for the steps that follow."""
    res = _run(3, "cpu", "morton", 1500, tmp_path, "clustered", port=29643, part_slot=16)
    assert res["ok"], res
    assert all(s["repeats"] == 0 and s["partition_overflows"] >= 1 and s["partition_slot_next"] > 16 for s in res["stats"])
    assert sum(s["owned"] for s in res["stats"]) == 1500


@pytest.mark.parametrize("partition", ["morton", "hash"])
def test_unequal_rank_sizes_agree_on_buffer_sizes(tmp_path, partition):
    """n_local differs from rank to rank (1281 spheres hash to 3 ranks: 440 / 432 / 409); the capacities and slot
    sizes of the fixed-size exchanges must come out equal on every rank (one all-reduce in the constructor)."""
    res = _run(3, "cpu", partition, 1281, tmp_path, "uniform", port=29671 + (partition == "hash"))
    assert res["ok"], res
    assert len({s["halo_slot"] for s in res["stats"]}) == 1 and len({s["partition_slot"] for s in res["stats"]}) == 1


def test_moving_spheres_with_adopted_ownership(tmp_path):
    """adopt_owned() + motion: every sphere moves a few radii per step for 12 steps, the repartition slots start
    from what a scene at rest needs (nearly nothing), so the lists outgrow them.  Pair set exact at every step, the
    owned sets stay a partition of the scene."""
    res = _run(3, "cpu", "morton", 1500, tmp_path, "clustered", port=29673, adopt="move")
    assert res["ok"], res["problems"]
    assert res["steps"] >= 10 and res["expected"] > 100
    assert any(s["partition_overflows"] >= 1 for s in res["stats"])      # the slots really were outgrown


def test_adopting_the_owned_spheres_keeps_the_result(tmp_path):
    """adopt_owned(): what a rank owns becomes its next input (nothing is copied).  The pair set stays exact, and
    once the slots have adapted nothing travels any more: every sphere is kept by its rank."""
    res = _run(3, "cpu", "morton", 1500, tmp_path, "clustered", port=29644, adopt="adopt")
    assert res["ok"], res
    assert all(s["partition_slot"] <= 4096 for s in res["stats"])


@pytest.mark.parametrize("partition,kind", [("hash", "uniform"), ("morton", "uniform"), ("morton", "clustered")])
def test_gloo_cpu_world8(tmp_path, partition, kind):
    """BASELINE config 4's rank count: world = 8 (every rank has 3-4 halo peers under the ownership rule;
    with the hash partition every rank's region is the whole scene).  Union of all ranks' pairs == brute
    force, each pair once."""
    res = _run(8, "cpu", partition, 4000, tmp_path, kind, port=29651 + (partition == "hash") + 2 * (kind == "clustered"))
    assert res["ok"], res
    assert res["world"] == 8 and res["expected"] > 500
    assert all(s["owned"] > 0 for s in res["stats"])


@pytest.mark.gpu
@pytest.mark.parametrize("world,partition,kind,n", [(2, "morton", "uniform", 60000), (2, "hash", "uniform", 20000),
                                                    (4, "morton", "clustered", 40000), (3, "morton", "uniform", 50000),
                                                    (4, "hash", "clustered", 30000)])
def test_gloo_gpu_rehearsal(tmp_path, world, partition, kind, n):
    """The real HIP engine on every rank, ranks sharing the one GPU of the test box (at most 6 processes
    may have it open; the test runner and the launcher count: 4 ranks at most).  Besides the global pair set, every rank's sorted codes / ids, node records and boxes
    are compared with the oracle on the spheres that rank owns (dist_worker.per_rank_parity)."""
    res = _run(world, "gpu", partition, n, tmp_path, kind, port=29631 + world + 10 * (partition == "hash"))
    assert res["ok"], res
    assert all(s["rank_parity"] == "ok" for s in res["stats"])
    if partition == "morton" and kind == "uniform" and world == 2:      # a spatial partition keeps the halo thin
        assert max(s["ghosts"] for s in res["stats"]) < 0.6 * n / world


@pytest.mark.gpu
@pytest.mark.parametrize("partition", ["morton", "hash"])
def test_gloo_gpu_rehearsal_float64(tmp_path, partition):
    """f64 coordinates through the real HIP engine (9-word transport records, f64 tree and ghost walk), with the
    per-rank parity check against the f64 oracle."""
    res = _run(3, "gpu", partition, 30000, tmp_path, "uniform", port=29649 + (partition == "hash"), dtype="float64")
    assert res["ok"], res
    assert all(s["rank_parity"] == "ok" for s in res["stats"])


@pytest.mark.gpu
def test_halo_slot_overflow_is_repaired_on_the_gpu(tmp_path):
    res = _run(3, "gpu", "morton", 30000, tmp_path, "clustered", port=29646, halo_slot=64)
    assert res["ok"], res
    assert all(s["repeats"] >= 1 for s in res["stats"])


@pytest.mark.gpu
@pytest.mark.parametrize("world,partition,kind,n", [(4, "morton", "uniform", 1200000), (3, "morton", "clustered", 150000),
                                                    (2, "hash", "uniform", 300000)])
def test_gloo_gpu_rehearsal_at_size(tmp_path, world, partition, kind, n):
    """The protocol at sizes near the bench's (hundreds of thousands of spheres per rank: big tiles, the MSD plan,
    adapted slots): the union of the ranks' pairs against the single-GPU path on the whole scene, plus the per-rank
    parity against the oracle."""
    res = _run(world, "gpu", partition, n, tmp_path, kind, port=29660 + world, check="single")
    assert res["ok"], {k: v for k, v in res.items() if k != "stats"}
    assert all(s["rank_parity"] == "ok" for s in res["stats"])
    if partition == "morton" and kind == "uniform":      # eight boxes per region keep the halo a thin shell
        assert max(s["ghosts"] for s in res["stats"]) < 0.25 * n / world


@pytest.mark.gpu
def test_adopting_the_owned_spheres_keeps_the_result_on_the_gpu(tmp_path):
    res = _run(4, "gpu", "morton", 400000, tmp_path, "uniform", port=29645, check="single", adopt="adopt")
    assert res["ok"], {k: v for k, v in res.items() if k != "stats"}
    assert all(s["rank_parity"] == "ok" and s["partition_slot"] <= 8192 for s in res["stats"])


@pytest.mark.gpu
def test_partition_slot_overflow_loses_nothing_on_the_gpu(tmp_path):
    res = _run(3, "gpu", "morton", 30000, tmp_path, "uniform", port=29647, part_slot=256)
    assert res["ok"], res
    assert all(s["partition_overflows"] >= 1 and s["rank_parity"] == "ok" for s in res["stats"])
    assert sum(s["owned"] for s in res["stats"]) == 30000


@pytest.mark.gpu
def test_moving_spheres_with_adopted_ownership_on_the_gpu(tmp_path):
    res = _run(4, "gpu", "morton", 12000, tmp_path, "clustered", port=29674, adopt="move")
    assert res["ok"], res["problems"]
    assert res["steps"] >= 10
    assert any(s["partition_overflows"] >= 1 for s in res["stats"])
    assert all(s["rank_parity"] == "ok" for s in res["stats"])


@pytest.mark.gpu
def test_unequal_rank_sizes_on_the_gpu(tmp_path):
    res = _run(3, "gpu", "morton", 12811, tmp_path, "uniform", port=29675)
    assert res["ok"], res


@pytest.mark.gpu
def test_rccl_code_paths_on_one_gpu(tmp_path):
    """world_size 1 over the nccl (= RCCL) backend with every exchange forced to run: the real
    all_gather_into_tensor / all_to_all_single / all_reduce calls on device tensors, talking to
    themselves.  (More ranks need one GPU each: bench.py --gpus N on the 8-GPU node.)"""
    script = tmp_path / "one_rank.py"
    script.write_text('''
import os, sys, json
import numpy as np
sys.path.insert(0, %r)
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
from collision_amd import hip
from collision_amd.multi import DistributedCollider
import oracle
n = 50000
rng = np.random.RandomState(4)
coords = np.zeros((n, 4), np.float32); coords[:, :3] = rng.random_sample((n, 3))
radii = np.full(n, 0.006, np.float32)
for partition in ("morton", "hash"):
    dc = DistributedCollider(hip.Context(0), dist, n, group_size=64, pair_capacity=1 << 20, partition=partition,
                             exercise_single_rank=True)
    dc.set_local_spheres(coords, radii, np.arange(n, dtype=np.uint32))
    for _ in range(2):
        dc.step()
    dc.synchronize()
    cnt, ref = oracle.brute_force(coords, radii)
    got = set(tuple(sorted(p)) for p in dc.local_pairs().tolist())
    assert dc.global_pair_count() == cnt == len(got), (partition, dc.global_pair_count(), cnt, len(got))
    assert got == set(map(tuple, ref.tolist()))
dist.destroy_process_group()
print("ok")
''' % str(ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29677", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    proc = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0 and "ok" in proc.stdout, proc.stdout[-2000:] + proc.stderr[-4000:]


@pytest.mark.gpu
def test_bench_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` typed as a plain command (no launcher, WORLD_SIZE unset): the parent starts the ranks
    as a child torch.distributed.run and rank 0's JSON line comes out with n_gpus 2 (gloo here: both ranks share
    the one GPU of the test box; the driver's 8-GPU node runs the same command over RCCL)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(COLLISION_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    proc = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2",
                           "--no-radix", "--no-cpu"], env=env, capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, proc.stdout[-2000:] + proc.stderr[-4000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["config"]["spheres_total"] == 4000000 and res["value"] > 0
    assert res["pairs_found"] > 100000
