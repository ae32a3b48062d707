"""Several ranks of the multi-GPU protocol in ONE process (collision_amd.multi.LoopbackWorld).

CPU: world 8 with the NumPy/oracle engine double -- the lockstep driver and its copy "collectives" must give the
brute-force pair set like the gloo runs of test_multi.py.
GPU: BASELINE config 4 as specified -- world 8 x 2 M = 16 M uniform spheres, hash arrival -- with eight real HIP
engines sharing the one GPU of the test box: union of the eight ranks' pairs == the single-GPU path on the 16 M
scene, each pair once; every rank's sorted codes / ids / node records / boxes == the oracle on what it owns; both
partitions; every rank has its 3-4 halo peers.
"""
import numpy as np
import pytest

from collision_amd.multi import LoopbackWorld, handles, hash_owner
from tests.util import packed_pairs


def _undirected(p):
    p = np.asarray(p, dtype=np.uint32).reshape(-1, 2)
    return packed_pairs(np.stack([p.min(axis=1), p.max(axis=1)], axis=1))


def _load(world_obj, coords, radii, world):
    gids = np.arange(len(coords), dtype=np.uint32)
    owner = hash_owner(gids, world)
    for r in range(world):
        mine = owner == r
        world_obj.set_local_spheres(r, coords[mine], radii[mine], gids[mine])


@pytest.mark.parametrize("partition,kind", [("morton", "clustered"), ("hash", "uniform")])
def test_loopback_world8_cpu(oracle, partition, kind):
    from tests.dist_worker import OracleEngine, scene
    n, world = 3000, 8
    coords, radii = scene(n, world, kind)
    owner = hash_owner(np.arange(n, dtype=np.uint32), world)
    lw = LoopbackWorld(None, world, [int((owner == r).sum()) for r in range(world)],
                       engine=lambda cap: OracleEngine(cap, 1 << 20), group_size=64, pair_capacity=1 << 20,
                       partition=partition, slack=3.0)
    _load(lw, coords, radii, world)
    for _ in range(2):
        lw.step()
    pairs = lw.pairs()
    cnt, ref = oracle.brute_force(coords, radii)
    got = _undirected(np.concatenate(pairs))
    assert len(got) == cnt == lw.global_pair_count() and (np.diff(got) != 0).all()
    np.testing.assert_array_equal(got, _undirected(ref))
    assert sum(dc.n_owned for dc in lw.ranks) == n
    assert all(3 <= len(dc.peers_in) <= 4 and 3 <= len(dc.peers_out) <= 4 for dc in lw.ranks)


def test_loopback_moving_scene_cpu(oracle):
    """Motion + adopt_owned in the in-process world (same protocol as test_multi.py's gloo run, world 8)."""
    from tests import motion
    from tests.dist_worker import OracleEngine, scene
    from collision_amd.multi import DistributedCollider
    n, world = 2000, 8
    coords, radii = scene(n, world, "clustered")
    gids = np.arange(n, dtype=np.uint32)
    owner = hash_owner(gids, world)
    floor = DistributedCollider.MIN_PARTITION_SLOT
    DistributedCollider.MIN_PARTITION_SLOT = 8
    try:
        lw = LoopbackWorld(None, world, [int((owner == r).sum()) for r in range(world)],
                           engine=lambda cap: OracleEngine(cap, 1 << 20), group_size=64, pair_capacity=1 << 20,
                           partition="morton", slack=3.0)
        _load(lw, coords, radii, world)
        lw.step(); lw.synchronize(); lw.adopt_owned(); lw.step(); lw.synchronize()
        unit = motion.unit_for(float(np.median(radii)))
        ref = coords.copy()
        for k in range(10):
            lw.adopt_owned()
            for dc in lw.ranks:
                motion.advance_torch(dc.local_rows(), dc.local_gids(), k, unit)
            motion.advance_numpy(ref, gids, k, unit)
            lw.step()
            got = _undirected(np.concatenate(lw.pairs()))
            cnt, want = oracle.brute_force(ref, radii)
            assert len(got) == cnt, (k, len(got), cnt)
            np.testing.assert_array_equal(got, _undirected(want))
            owned = np.concatenate([dc.own_gids[:dc.n_owned].numpy().view(np.uint32) for dc in lw.ranks])
            np.testing.assert_array_equal(np.sort(owned), gids)
        assert sum(dc.stats["partition_overflows"] for dc in lw.ranks) >= 1
    finally:
        DistributedCollider.MIN_PARTITION_SLOT = floor


# ------------------------------------------------------------------------------------------------ GPU: config 4
def _config4_scene(n_per_rank, world):
    n = n_per_rank * world
    rng = np.random.RandomState(4)
    coords = np.zeros((n, 4), np.float32)
    coords[:, :3] = rng.random_sample((n, 3))
    r = np.float32(0.001 * (1e6 / n) ** (1.0 / 3.0))          # contacts per sphere of config 2
    return coords, np.full(n, r, np.float32)


@pytest.mark.gpu
@pytest.mark.parametrize("partition", ["morton", "hash"])
def test_config4_world8_two_million_per_rank(hip_env, oracle, partition):
    import torch
    from collision_amd.collision import Collider
    from tests.dist_worker import per_rank_parity
    from tests.util import run_collider
    ctx, cq = hip_env
    world, per = 8, 2000000
    coords, radii = _config4_scene(per, world)
    n = len(coords)
    gids = np.arange(n, dtype=np.uint32)
    owner = hash_owner(gids, world)
    lw = LoopbackWorld(ctx, world, [int((owner == r).sum()) for r in range(world)], group_size=256,
                       pair_capacity=1 << 20, partition=partition)
    _load(lw, coords, radii, world)
    lw.step()
    lw.synchronize()                       # slots adapt to the scene
    lw.step()
    pairs = lw.pairs()
    assert all(dc.repeats <= 1 for dc in lw.ranks)
    assert sum(dc.n_owned for dc in lw.ranks) == n
    assert all(3 <= len(dc.peers_in) <= 4 and 3 <= len(dc.peers_out) <= 4 for dc in lw.ranks)
    ghosts = [dc.stats["ghosts"] for dc in lw.ranks]
    if partition == "hash":                # every rank answers for everything its 3-4 peers own
        assert min(ghosts) > 2.9 * per
    else:                                  # compact Morton ranges: the halo is a thin shell
        assert max(ghosts) < 0.25 * per
    for dc in lw.ranks:
        assert per_rank_parity(dc) == "ok"
    got = _undirected(np.concatenate(pairs))
    assert (np.diff(got) != 0).all()       # each pair once
    # reference: the single-GPU path on the whole 16 M scene (oracle-exact at 20 M: test_pipeline_parity)
    del lw
    torch.cuda.empty_cache()
    cnt, ref = run_collider(ctx, cq, Collider(ctx, n, 64, 256), coords, radii, 1 << 21)
    assert cnt <= 1 << 21 and cnt == len(got)
    np.testing.assert_array_equal(got, _undirected(ref))


@pytest.mark.gpu
def test_a_step_never_waits_for_the_owned_count(hip_env, oracle):
    """Round 4: after the repartition every launch reads the number of owned spheres from the device word the unpack leaves
    (col_collide_plan_dev and the *_dev halo entry points), so step() only enqueues: the host-visible word is not looked at
    before synchronize() / n_owned.  Four ranks with real HIP engines on the one GPU, Morton repartition; the owned counts
    differ from rank to rank and from the capacity; pairs == brute force, per-rank arrays == the oracle on what a rank owns."""
    from collision_amd.multi import HipEngine
    from tests.dist_worker import per_rank_parity, scene
    ctx, _ = hip_env
    world, n = 4, 30000
    coords, radii = scene(n, world, "clustered")
    gids = np.arange(n, dtype=np.uint32)
    owner = hash_owner(gids, world)
    lw = LoopbackWorld(ctx, world, [int((owner == r).sum()) for r in range(world)], group_size=64,
                       pair_capacity=1 << 20, partition="morton")
    assert all(dc.engine.device_count for dc in lw.ranks)
    _load(lw, coords, radii, world)
    polls = []
    real = HipEngine.owned_count
    HipEngine.owned_count = lambda self: polls.append(1) or real(self)
    try:
        lw.step()
        lw.step()
        assert polls == []                      # two steps enqueued, nobody asked
        pairs = lw.pairs()                      # synchronize(): now the counts are read
        assert len(polls) >= world
    finally:
        HipEngine.owned_count = real
    owned = [dc.n_owned for dc in lw.ranks]
    assert sum(owned) == n and len(set(owned)) > 1 and all(m < dc.capacity for m, dc in zip(owned, lw.ranks))
    for dc in lw.ranks:
        assert per_rank_parity(dc) == "ok"
    cnt, ref = oracle.brute_force(coords, radii)
    got = _undirected(np.concatenate(pairs))
    assert len(got) == cnt and (np.diff(got) != 0).all()
    np.testing.assert_array_equal(got, _undirected(ref))
    # the launches of a step are sized for a bound a few per cent above the counts seen; a rank that comes to own more works on
    # the first `bound` of its spheres only, which synchronize() sees (all ranks repeat the step with the bound raised)
    assert all(dc.engine.run_bound < dc.capacity and dc.engine.run_bound >= dc.n_owned for dc in lw.ranks)
    repeats = [dc.repeats for dc in lw.ranks]
    lw.ranks[1].engine.run_bound = 512                  # far too small
    lw.ranks[2].engine.run_bound = lw.ranks[2].n_owned - 1
    lw.step()
    got = _undirected(np.concatenate(lw.pairs()))
    np.testing.assert_array_equal(got, _undirected(ref))
    assert all(dc.repeats == r0 + 1 for dc, r0 in zip(lw.ranks, repeats))
    assert all(dc.engine.run_bound >= dc.n_owned for dc in lw.ranks)
    for dc in lw.ranks:
        assert per_rank_parity(dc) == "ok"
