"""Kernel-level tests of the multi-GPU path's device steps (collision_amd/csrc/multi.hip) through the C ABI,
against NumPy and the oracle's Morton codes.  Nothing in the reference to mirror (it is single-device,
SURVEY.md 8e): what is pinned here is the protocol's own contract -- sample payload, global range, splitters
(exact order statistics of the gathered codes), owners, stable grouping, slot layout, unpack order, region boxes."""
import ctypes as C

import numpy as np
import pytest

from collision_amd import hip
from collision_amd._lib import call
from collision_amd.multi import REGION_BOXES, SAMPLES
from tests.util import download, upload

pytestmark = pytest.mark.gpu
DTYPES = ["float32", "float64"]


def _rows(rng, n, dt, clustered=False):
    pts = rng.normal(0.4, 0.05, size=(n, 3)) if clustered else rng.random_sample((n, 3))
    rows = np.zeros((n, 4), dt)
    rows[:, :3] = pts
    rows[:, 3] = rng.uniform(0.001, 0.004, size=n)
    return rows


def _payload(rows):
    n = len(rows)
    pos = (np.arange(SAMPLES, dtype=np.int64) * (n - 1)) // (SAMPLES - 1)
    return np.concatenate([rows[pos], rows.min(axis=0)[None], rows.max(axis=0)[None]])


def _sample(ctx, cq, rows, dt):
    scratch = hip.Buffer(ctx, call.col_partition_scratch_bytes())
    flags = upload(ctx, np.full(4, 7, np.uint32))
    out = hip.Buffer(ctx, (SAMPLES + 2) * 4 * np.dtype(dt).itemsize)
    rows_buf = upload(ctx, rows) if len(rows) else hip.Buffer(ctx, 64)
    call.col_partition_sample(cq.stream, rows_buf.ptr, len(rows), SAMPLES, out.ptr, scratch.ptr, flags.ptr, 4,
                              np.dtype(dt).itemsize)
    return download(cq, out, dt, (SAMPLES + 2, 4)), download(cq, flags, np.uint32, 4)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("n", [1, 5, 1022, 3000, 700001])
def test_partition_sample(hip_env, dt, n):
    ctx, cq = hip_env
    rows = _rows(np.random.RandomState(n), n, dt)
    got, flags = _sample(ctx, cq, rows, dt)
    np.testing.assert_array_equal(got, _payload(rows))
    assert not flags.any()


def test_partition_sample_of_an_empty_rank(hip_env):
    ctx, cq = hip_env
    got, _ = _sample(ctx, cq, np.zeros((0, 4), np.float32), "float32")
    assert np.isposinf(got[:SAMPLES, :3]).all() and (got[:SAMPLES, 3] == 0).all()
    assert np.isposinf(got[SAMPLES]).all() and np.isneginf(got[SAMPLES + 1]).all()


def _plan(ctx, cq, gathered, rows, world, dt):
    n, cb = len(rows), np.dtype(dt).itemsize
    tile = call.col_radix_tile(max(n, 1), 4, 4)
    nb = -(-max(n, 1) // tile)
    bufs = dict(range8=hip.Buffer(ctx, 8 * cb), split=upload(ctx, np.zeros(256, np.uint32)),
                dest=hip.Buffer(ctx, 4 * max(n, 1)), hist=upload(ctx, np.zeros(256 * nb, np.uint32)),
                counts=upload(ctx, np.full(256, 99, np.uint32)))
    g, r = upload(ctx, gathered), upload(ctx, rows) if n else hip.Buffer(ctx, 64)
    call.col_partition_plan(cq.stream, g.ptr, world, SAMPLES, r.ptr, n, bufs["range8"].ptr, bufs["split"].ptr,
                            bufs["dest"].ptr, bufs["hist"].ptr, bufs["counts"].ptr, cb)
    cq.finish()
    return bufs, nb


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("world,clustered", [(1, False), (2, False), (3, True), (8, False), (8, True), (16, True)])
def test_partition_plan(hip_env, oracle, dt, world, clustered):
    """Global range = fold of the gathered ranges; splitter q = the code of rank q * (count / world) among the codes
    of ALL gathered rows (the device finds it by radix select, here by sorting); owner = number of splitters <= code."""
    ctx, cq = hip_env
    rng = np.random.RandomState(world)
    per_rank = [_rows(rng, int(rng.randint(2000, 9000)), dt, clustered and q % 2 == 0) for q in range(world)]
    if world == 3:
        per_rank[1] = np.zeros((0, 4), dt)                       # an empty rank contributes rows at +inf
    gathered = np.stack([_payload(r) if len(r) else
                         np.concatenate([np.tile(np.array([np.inf, np.inf, np.inf, 0], dt), (SAMPLES, 1)),
                                         np.full((1, 4), np.inf, dt), np.full((1, 4), -np.inf, dt)]) for r in per_rank])
    rows = per_rank[0]
    bufs, nb = _plan(ctx, cq, gathered, rows, world, dt)
    grange = np.stack([gathered[:, SAMPLES].min(axis=0), gathered[:, SAMPLES + 1].max(axis=0)])
    np.testing.assert_array_equal(download(cq, bufs["range8"], dt, (2, 4)), grange)
    codes = np.sort(oracle.morton(gathered.reshape(-1, 4), grange))
    want_split = codes[np.arange(1, world) * (len(codes) // world)]
    np.testing.assert_array_equal(download(cq, bufs["split"], np.uint32, 256)[:world - 1], want_split)
    mine = oracle.morton(rows, grange)
    want_dest = np.searchsorted(want_split, mine, side="right").astype(np.uint32)
    np.testing.assert_array_equal(download(cq, bufs["dest"], np.uint32, len(rows)), want_dest)
    np.testing.assert_array_equal(download(cq, bufs["counts"], np.uint32, 256)[:world], np.bincount(want_dest, minlength=world))


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("world,rank,slot", [(1, 0, 64), (4, 2, 4096), (4, 0, 100), (8, 7, 2048)])
def test_partition_group_and_unpack(hip_env, oracle, dt, world, rank, slot):
    """Grouping is stable; what the rank keeps lands at the front of its owned arrays; every other rank's list goes
    into its slot (header = full length, then min(length, slot) records) and what does not fit stays here, behind
    the kept rows in owner order; the unpack appends received slots in rank order and publishes the owned count."""
    ctx, cq = hip_env
    cb = np.dtype(dt).itemsize
    rw = cb + 1                                                  # words per record: 4 scalars + gid
    rng = np.random.RandomState(17 * world + rank)
    n = 20000
    rows = _rows(rng, n, dt)
    gids = rng.permutation(1 << 20)[:n].astype(np.uint32)
    gathered = np.stack([_payload(_rows(rng, 3000, dt)) for _ in range(world)])
    bufs, nb = _plan(ctx, cq, gathered, rows, world, dt)
    dest = download(cq, bufs["dest"], np.uint32, n)
    counts = np.bincount(dest, minlength=world)
    cap = n + 5000
    others = max(1, world - 1)
    b = dict(rows=upload(ctx, rows), gids=upload(ctx, gids),
             send=upload(ctx, np.full(others * (slot + 1) * rw, 0xABABABAB, np.uint32)),
             own_rows=upload(ctx, np.zeros((cap, 4), dt)), own_gids=upload(ctx, np.zeros(cap, np.uint32)),
             own_radii=upload(ctx, np.zeros(cap, dt)), flags=upload(ctx, np.zeros(4, np.uint32)),
             owned=upload(ctx, np.zeros(2, np.uint32)))
    call.col_partition_group(cq.stream, b["rows"].ptr, b["gids"].ptr, n, bufs["dest"].ptr, bufs["hist"].ptr,
                             bufs["counts"].ptr, world, rank, slot, b["send"].ptr,
                             b["own_rows"].ptr, b["own_gids"].ptr, b["own_radii"].ptr, cap, b["flags"].ptr, cb)
    perm = np.argsort(dest, kind="stable")                     # (the grouping is stable: lists keep the input order)
    kept = perm[dest[perm] == rank]
    stay = np.concatenate([kept] + [perm[dest[perm] == q][slot:] for q in range(world) if q != rank])
    if slot == 100:
        assert len(stay) > len(kept)                              # (this case really overflows its slots)
    kept = stay
    np.testing.assert_array_equal(download(cq, b["own_rows"], dt, (cap, 4))[:len(kept)], rows[kept])
    np.testing.assert_array_equal(download(cq, b["own_gids"], np.uint32, cap)[:len(kept)], gids[kept])
    np.testing.assert_array_equal(download(cq, b["own_radii"], dt, cap)[:len(kept)], rows[kept, 3])
    send = download(cq, b["send"], np.uint32, others * (slot + 1) * rw).reshape(others, slot + 1, rw)
    longest = 0
    for q in range(world):
        if q == rank:
            continue
        k = q if q < rank else q - 1
        lst = perm[dest[perm] == q]
        longest = max(longest, len(lst))
        assert send[k, 0, 0] == len(lst) and not send[k, 0, 1:].any()
        cnt = min(len(lst), slot)
        np.testing.assert_array_equal(send[k, 1:1 + cnt, :rw - 1], rows[lst[:cnt]].view(np.uint32).reshape(cnt, rw - 1))
        np.testing.assert_array_equal(send[k, 1:1 + cnt, rw - 1], gids[lst[:cnt]])
    assert download(cq, b["flags"], np.uint32, 4)[2] == longest

    # unpack: pretend every other rank sent `send` (slot k of this rank's send buffer comes back as slot k)
    word = C.c_void_p()
    call.col_host_alloc(C.byref(word), 64)
    try:
        C.c_uint64.from_address(word.value).value = 0
        call.col_partition_unpack(cq.stream, b["send"].ptr, world, rank, slot, bufs["counts"].ptr, b["own_rows"].ptr,
                                  b["own_gids"].ptr, b["own_radii"].ptr, cap, b["owned"].ptr, word.value, 41, b["flags"].ptr, cb)
        cq.finish()
        expect_rows, expect_gids = [rows[kept]], [gids[kept]]
        for q in range(world):
            if q != rank:
                lst = perm[dest[perm] == q][:slot]
                expect_rows.append(rows[lst])
                expect_gids.append(gids[lst])
        expect_rows, expect_gids = np.concatenate(expect_rows), np.concatenate(expect_gids)
        m = len(expect_rows)
        assert C.c_uint64.from_address(word.value).value == (41 << 32) | m
        np.testing.assert_array_equal(download(cq, b["owned"], np.uint32, 2), [min(m, cap), m])
        np.testing.assert_array_equal(download(cq, b["own_rows"], dt, (cap, 4))[:m], expect_rows)
        np.testing.assert_array_equal(download(cq, b["own_gids"], np.uint32, cap)[:m], expect_gids)
        np.testing.assert_array_equal(download(cq, b["own_radii"], dt, cap)[:m], expect_rows[:, 3])
    finally:
        call.col_host_free(word.value)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("n,with_range", [(0, True), (1, True), (50000, True), (50000, False), (300001, True)])
def test_region_boxes(hip_env, oracle, dt, n, with_range):
    ctx, cq = hip_env
    cb = np.dtype(dt).itemsize
    rng = np.random.RandomState(n + 1)
    rows = _rows(rng, n, dt)
    if n > 1000:
        rows = rows[rows[:, 0] + rows[:, 1] < 1.2]               # leave some octants empty
        n = len(rows)
    grange = np.array([[0, 0, 0, 0], [1, 1, 1, 0]], dt)
    scratch = hip.Buffer(ctx, call.col_partition_scratch_bytes())
    counters = upload(ctx, np.full(8, 5, np.uint32))
    out = hip.Buffer(ctx, REGION_BOXES * 8 * cb)
    r, g = upload(ctx, rows) if n else hip.Buffer(ctx, 64), upload(ctx, grange)
    call.col_region_boxes(cq.stream, r.ptr, n, g.ptr if with_range else None, scratch.ptr, out.ptr, counters.ptr, 8, cb)
    got = download(cq, out, dt, (REGION_BOXES, 2, 4))
    assert not download(cq, counters, np.uint32, 8).any()
    octant = (oracle.morton(rows, grange) >> 27) if (with_range and n) else np.zeros(n, np.int64)
    for o in range(REGION_BOXES):
        sel = rows[octant == o]
        if len(sel) == 0:
            assert np.isposinf(got[o, 0, :3]).all() and np.isneginf(got[o, 1, :3]).all()
            continue
        rmax = sel[:, 3].max()
        np.testing.assert_array_equal(got[o, 0, :3], sel[:, :3].min(axis=0) - rmax)
        np.testing.assert_array_equal(got[o, 1, :3], sel[:, :3].max(axis=0) + rmax)
    assert (got[:, :, 3] == 0).all()


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("n,n_peers,slot", [(1, 1, 16), (5000, 3, 4096), (70001, 4, 512), (300000, 8, 300000)])
def test_halo_select_pack_and_ghost_queries(hip_env, oracle, dt, n, n_peers, slot):
    """The halo branch's device steps in isolation: col_select_overlap_multi (lists = the owned spheres overlapping
    any of a peer's eight region boxes; one reservation per block and peer), col_pack_slots (header = full length,
    then min(length, slot) records) and col_traverse_ghost_slots (the received records as queries against a tree:
    every (ghost, local) AABB overlap, no position pruning) against NumPy."""
    from collision_amd.collision import Collider
    from collision_amd.misc import roundUp
    ctx, cq = hip_env
    cb = np.dtype(dt).itemsize
    rw = cb + 1
    rng = np.random.RandomState(n + n_peers)
    rows = _rows(rng, n, dt)
    gids = rng.permutation(1 << 22)[:n].astype(np.uint32)
    world = n_peers + 2
    boxes = np.zeros((world, REGION_BOXES, 2, 4), dt)
    boxes[:, :, 0, :3], boxes[:, :, 1, :3] = np.inf, -np.inf              # empty octants
    for q in range(world):
        for o in rng.choice(REGION_BOXES, size=3, replace=False):
            lo = rng.uniform(0, 0.9, size=3)
            boxes[q, o, 0, :3], boxes[q, o, 1, :3] = lo, lo + rng.uniform(0.01, 0.25, size=3)
    peers = list(rng.choice(world, size=n_peers, replace=False))
    rows_buf, gids_buf, boxes_buf = upload(ctx, rows), upload(ctx, gids), upload(ctx, boxes)
    lists, counts = hip.Buffer(ctx, 8 * n * 4), upload(ctx, np.zeros(8, np.uint32))
    call.col_select_overlap_multi(cq.stream, rows_buf.ptr, n, boxes_buf.ptr, (C.c_int * n_peers)(*peers), n_peers, n,
                                  lists.ptr, counts.ptr, cb)
    got_counts = download(cq, counts, np.uint32, 8)
    got_lists = download(cq, lists, np.uint32, (8, n))
    lo3, hi3 = rows[:, :3] - rows[:, 3:4], rows[:, :3] + rows[:, 3:4]
    want = []
    for k, q in enumerate(peers):
        hit = np.zeros(n, bool)
        for b in boxes[q]:
            hit |= ((hi3 > b[0, :3]) & (lo3 < b[1, :3])).all(axis=1)
        want.append(np.nonzero(hit)[0])
        assert got_counts[k] == len(want[k])
        np.testing.assert_array_equal(np.sort(got_lists[k, :len(want[k])]), want[k])
    # slots
    send = upload(ctx, np.full(n_peers * (slot + 1) * rw, 0xCDCDCDCD, np.uint32))
    call.col_pack_slots(cq.stream, rows_buf.ptr, gids_buf.ptr, lists.ptr, n, counts.ptr, n_peers, min(max(n, 1), slot),
                        send.ptr, n_peers * (slot + 1), slot, cb)
    rec = download(cq, send, np.uint32, (n_peers, slot + 1, rw))
    for k in range(n_peers):
        cnt = min(len(want[k]), slot)
        assert rec[k, 0, 0] == len(want[k]) and not rec[k, 0, 1:].any()
        src = got_lists[k, :cnt]
        np.testing.assert_array_equal(rec[k, 1:1 + cnt, :rw - 1], rows[src].view(np.uint32).reshape(cnt, rw - 1))
        np.testing.assert_array_equal(rec[k, 1:1 + cnt, rw - 1], gids[src])
    # the packed slots as ghosts against a tree over OTHER spheres
    m = 20000
    local = _rows(np.random.RandomState(99), m, dt)
    local[:, 3] *= 4
    lg = np.arange(m, dtype=np.uint32) + 5000000
    col = Collider(ctx, m, 16, 64, dt)
    nb, pb = hip.Buffer(ctx, 4), hip.Buffer(ctx, 8)
    col.get_collisions(cq, upload(ctx, local), upload(ctx, np.ascontiguousarray(local[:, 3])), nb, None, 0)
    cap = 1 << 22
    results = []
    for packets in (True, False):          # the packet walk (Morton-ordered ghosts, 64 per wave) and the lane-per-ghost walk
        pairs, counter, flags = hip.Buffer(ctx, cap * 8), upload(ctx, np.zeros(1, np.uint32)), upload(ctx, np.zeros(4, np.uint32))
        scratch = hip.Buffer(ctx, call.col_ghost_scratch_bytes(n_peers, slot)) if packets else None
        call.col_traverse_ghost_slots(cq.stream, send.ptr, n_peers, slot, col._bounds_buf.ptr, m, upload(ctx, lg).ptr,
                                      pairs.ptr, counter.ptr, cap, flags.ptr, cb, scratch.ptr if packets else None)
        count = int(download(cq, counter, np.uint32, 1)[0])
        results.append((count, download(cq, pairs, np.uint32, (min(count, cap), 2)), download(cq, flags, np.uint32, 4)))
    (count, got, f), (count_lane, got_lane, f_lane) = results
    assert count == count_lane and sorted(map(tuple, got.tolist())) == sorted(map(tuple, got_lane.tolist()))
    np.testing.assert_array_equal(f, f_lane)
    llo, lhi = local[:, :3] - local[:, 3:4], local[:, :3] + local[:, 3:4]
    expect = []
    n_ghosts = 0
    for k in range(n_peers):
        cnt = min(len(want[k]), slot)
        n_ghosts += cnt
        for s in got_lists[k, :cnt]:
            hit = ((hi3[s] > llo) & (lo3[s] < lhi)).all(axis=1)
            expect += [(int(gids[s]), int(lg[j])) for j in np.nonzero(hit)[0]]
    assert count == len(expect) <= cap
    assert sorted(map(tuple, got.tolist())) == sorted(expect)
    assert f[0] == max([len(w) for w in want] + [0]) and f[1] == n_ghosts
