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
