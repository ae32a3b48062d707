"""Worker for the multi-rank tests: run by torch.distributed.run, one process per rank.

    python -m torch.distributed.run --nproc-per-node R tests/dist_worker.py {cpu|gpu} {morton|hash} N OUT

cpu: gloo backend, the device engine is replaced by a NumPy/oracle test double (tests only).
gpu: gloo backend as well, but every rank drives the real HIP engine on cuda:0 (a rehearsal of the
     N-GPU path on the one-GPU box; RCCL itself needs one GPU per rank and is exercised by bench.py).
Rank 0 checks the union of all ranks' pairs against a brute force over the whole scene.
"""
import json
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def _protocol_ops():
    from collision_amd.multi import ProtocolOps
    return ProtocolOps


class OracleEngine(_protocol_ops()):
    """CPU stand-in for collision_amd.multi.HipEngine (same methods, torch CPU tensors); the small
    tensor steps between the collectives are ProtocolOps' tensor-library versions."""

    def __init__(self, capacity, pair_capacity, coord_dtype=np.dtype("float32")):
        import contextlib
        import torch
        import oracle
        self.torch, self.oracle, self._null = torch, oracle, contextlib.nullcontext
        self.device = torch.device("cpu")
        self.capacity, self.pair_capacity = capacity, pair_capacity
        self.coord_dtype = np.dtype(coord_dtype)
        self._f = torch.float32 if self.coord_dtype.itemsize == 4 else torch.float64
        self._wf = self.coord_dtype.itemsize // 4                      # int32 words per scalar
        z4 = lambda n: torch.zeros((n, 4), dtype=self._f)
        zi = lambda n: torch.zeros(n, dtype=torch.int32)
        self._z5 = self._recs = lambda n: torch.zeros((n, 4 * self._wf + 1), dtype=torch.int32)
        self.rows_in, self.gids_in = z4(capacity), zi(capacity)
        self.owned_rows, self.owned_gids = z4(capacity), zi(capacity)
        self.part_send = self.part_recv = None
        self.halo_send = self.halo_recv = None
        self.found = []
        self.n_owned = 0
        self._longest = self._ghosts = self._longest_part = self._kept = self._m = 0

    def load(self, coords4, radii, gids):
        n = len(coords4)
        host = np.array(coords4, dtype=self.coord_dtype, copy=True)
        host[:, 3] = radii
        self.rows_in[:n] = self.torch.from_numpy(host)
        self.gids_in[:n] = self.torch.from_numpy(np.asarray(gids).astype(np.uint32).view(np.int32))
        return n

    def swap_input_and_owned(self):
        self.rows_in, self.owned_rows = self.owned_rows, self.rows_in
        self.gids_in, self.owned_gids = self.owned_gids, self.gids_in

    def begin_step(self, sampled):
        self._longest = self._ghosts = self._longest_part = 0

    def mark_fork(self):
        pass

    def fork(self):
        pass

    def join(self):
        pass

    def halo_stream(self):
        return self._null()

    def codes_of_rows(self, rows, range8):
        codes = self.oracle.morton(rows.numpy().copy(), range8.numpy().reshape(2, 4))
        return self.torch.from_numpy(codes.view(np.int32).copy())

    def pack5(self, rows, gids, idx, idx_offset, n, out5, out_offset=0):
        t = self.torch
        sel = slice(0, n) if idx is None else idx[idx_offset:idx_offset + n].long()
        out5[out_offset:out_offset + n, :4 * self._wf] = rows[sel].contiguous().view(t.int32)
        out5[out_offset:out_offset + n, 4 * self._wf] = gids[sel]

    def unpack5(self, rec5, n, rows, gids, radii=None):
        t = self.torch
        rows[:n] = rec5[:n, :4 * self._wf].contiguous().view(self._f)
        gids[:n] = rec5[:n, 4 * self._wf]

    def collide(self, rows, gids, n):
        self.n_owned, self.found = n, []
        self._rows, self._gids = rows[:n].numpy().copy(), gids[:n].numpy().view(np.uint32).copy()
        if n == 0:
            return
        c4 = self._rows.copy()
        radii = c4[:, 3].copy()
        if n >= 2:
            count = self.oracle.collide(c4, radii, capacity=0, want=False)["count"]
            res = self.oracle.collide(c4, radii, capacity=count, want=False)
            self.found = [(int(self._gids[a]), int(self._gids[b])) for a, b in res["pairs"]]
        self._lo = (self._rows[:, :3] - radii[:, None]).astype(self.coord_dtype)
        self._hi = (self._rows[:, :3] + radii[:, None]).astype(self.coord_dtype)

    def ensure_slots(self, slot, n_out, n_in):
        want_s, want_r = max(1, n_out) * (slot + 1), max(1, n_in) * (slot + 1)
        if self.halo_send is None or self.halo_send.shape[0] != want_s:
            self.halo_send = self._z5(want_s)
        if self.halo_recv is None or self.halo_recv.shape[0] != want_r:
            self.halo_recv = self._z5(want_r)

    def select_and_pack(self, rows, gids, n, boxes_dev, peers, slot):
        t = self.torch
        boxes = boxes_dev.numpy()
        r = rows[:n].numpy()
        lo, hi = r[:, :3] - r[:, 3:4], r[:, :3] + r[:, 3:4]
        for k, q in enumerate(peers):
            hit = np.zeros(n, bool)
            for b in boxes[q]:                  # a region is several boxes
                hit |= ((hi > b[0:3]) & (lo < b[4:7])).all(axis=1)
            idx = t.from_numpy(np.nonzero(hit)[0].astype(np.int32))
            self._longest = max(self._longest, len(idx))
            base = k * (slot + 1)
            self.halo_send[base] = 0
            self.halo_send[base, 0] = len(idx)
            self.pack5(rows, gids, idx, 0, min(len(idx), slot), self.halo_send, base + 1)

    def ghost_queries(self, n_in, slot, owned_gids, expected=None):
        for k in range(n_in):
            base = k * (slot + 1)
            length = int(self.halo_recv[base, 0])
            self._longest = max(self._longest, length)
            cnt = min(length, slot)
            if self.n_owned == 0 or cnt == 0:
                continue
            rec = self.halo_recv[base + 1:base + 1 + cnt]
            g = rec[:, :4 * self._wf].contiguous().view(self._f).numpy()
            glo = (g[:, :3] - g[:, 3:4]).astype(self.coord_dtype)
            ghi = (g[:, :3] + g[:, 3:4]).astype(self.coord_dtype)
            gg = rec[:, 4 * self._wf].contiguous().numpy().view(np.uint32)
            self._ghosts += cnt
            for i in range(cnt):
                hit = ((ghi[i] > self._lo) & (glo[i] < self._hi)).all(axis=1)
                self.found += [(int(gg[i]), int(self._gids[j])) for j in np.nonzero(hit)[0]]

    def halo_stats(self):
        return self._longest, self._ghosts, self._longest_part

    def pair_count(self):
        return len(self.found)

    def read_pairs(self):
        return np.array(self.found, dtype=np.uint32).reshape(-1, 2)

    def synchronize(self):
        pass


def scene(n, world, kind, seed=4):
    rng = np.random.RandomState(seed)
    if kind == "clustered":
        centres = rng.uniform(0.2, 0.8, size=(4, 3))
        pts = np.concatenate([rng.normal(c, 0.06, size=(n // 4, 3)) for c in centres])
        pts = np.concatenate([pts, rng.uniform(0, 1, size=(n - len(pts), 3))])
    else:
        pts = rng.random_sample((n, 3))
    coords = np.zeros((n, 4), np.float32)
    coords[:, :3] = pts
    radii = rng.uniform(0.2, 1.0, size=n).astype(np.float32) * np.float32(0.6 * n ** (-1.0 / 3.0))
    return coords, radii


def per_rank_parity(dc):
    """this rank's sorted codes / ids, node records and node boxes must equal the oracle run
    on the spheres this rank owns (the local pipeline normalises with the bounds of that subset, exactly
    as the single-GPU path does).  Returns "ok" or a description of the first mismatch."""
    import oracle
    from collision_amd.collision import Node
    from collision_amd.misc import roundUp
    e, m = dc.engine, dc.n_owned
    if m < 2:
        return "ok"
    rows = dc.own_rows[:m].cpu().numpy().copy()
    radii = rows[:, 3].copy()
    c = e.collider
    padded = roundUp(m, 2 * e.group_size)
    ref = oracle.collide(rows, radii, padded=padded, capacity=0, want=True)
    cq = e.cq
    codes = hip_read(cq, c._codes_bufs[1], np.uint32, padded)
    ids = hip_read(cq, c._ids_bufs[1], np.uint32, padded)
    nodes = hip_read(cq, c._nodes_buf, Node, 2 * m - 1)
    bounds = hip_read(cq, c._bounds_buf, e.coord_dtype, (2 * m - 1, 2, 4))
    leaf = m - 1
    checks = (("codes", codes, ref["codes"]), ("ids", ids, ref["ids"]),
              ("right_edge", nodes["right_edge"], ref["nodes"]["right_edge"]),
              ("parent", nodes["parent"][1:], ref["nodes"]["parent"][1:]),
              ("children", nodes["data"][:leaf], ref["nodes"]["data"][:leaf]),
              ("leaf ids", nodes["data"][leaf:, 0], ref["nodes"]["data"][leaf:, 0]),
              ("boxes", bounds[:, :, :3], ref["bounds"][:, :, :3]))
    for name, got, want in checks:
        if not np.array_equal(got, want):
            return "rank %d: %s differ from the oracle on its %d owned spheres" % (dc.rank, name, m)
    return "ok"


def hip_read(cq, buf, dtype, shape):
    from collision_amd import hip
    return hip.read_buffer(cq, buf, dtype, shape)


def moving_scene(dc, dist, mode, coords, radii, gids, out, steps=12):
    """Every sphere moves a few radii per step (tests/motion.py) where it lives: step, adopt_owned, advance, step ...
    with the repartition slots adapted to what the first (motionless) adopted step needed, so the lists of the
    moving scene outgrow them.  Every step: the union of all ranks' pairs == brute force on the advanced scene, and
    the owned sets of all ranks are a partition of the scene (no sphere lost or duplicated)."""
    import oracle
    from tests import motion
    rank, world = dist.get_rank(), dist.get_world_size()
    type(dc).MIN_PARTITION_SLOT = 8                # (let the slots really shrink to the lists of a scene at rest)
    unit = motion.unit_for(float(np.median(radii)))
    ref = coords.copy()
    dc.step()
    dc.synchronize()
    dc.adopt_owned()
    dc.step()
    dc.synchronize()                               # nothing moved: every list is empty, the slots shrink
    problems, cnt = [], 0
    for k in range(steps):
        dc.adopt_owned()
        motion.advance_torch(dc.local_rows(), dc.local_gids(), k, unit)
        motion.advance_numpy(ref, gids, k, unit)
        dc.step()
        pairs = dc.local_pairs()
        total = dc.global_pair_count()
        owned = dc.own_gids[:dc.n_owned].cpu().numpy().view(np.uint32).copy()
        rows = dc.own_rows[:dc.n_owned].cpu().numpy().copy()
        gathered = [None] * world
        dist.gather_object((pairs, owned, rows), gathered if rank == 0 else None, dst=0)
        if rank == 0:
            cnt, want = oracle.brute_force(ref, radii)
            got = [tuple(sorted(t)) for p, _, _ in gathered for t in p.tolist()]
            all_owned = np.concatenate([o for _, o, _ in gathered])
            all_rows = np.concatenate([r for _, _, r in gathered])
            order = np.argsort(all_owned, kind="stable")
            if not (len(got) == len(set(got)) == cnt == total and set(got) == set(map(tuple, want.tolist()))):
                problems.append("step %d: %d pairs (%d unique, counted %d), brute force %d" % (k, len(got), len(set(got)), total, cnt))
            if not np.array_equal(all_owned[order], gids):
                problems.append("step %d: the owned sets are not a partition of the scene (%d of %d)" % (k, len(all_owned), len(gids)))
            elif not (np.array_equal(all_rows[order][:, :3], ref[:, :3]) and np.array_equal(all_rows[order][:, 3], radii)):
                problems.append("step %d: owned rows differ from the advanced scene" % k)
    stats = dict(dc.stats)
    stats["repeats"] = dc.repeats
    if mode == "gpu":
        stats["rank_parity"] = per_rank_parity(dc)
    gathered = [None] * world
    dist.gather_object(stats, gathered if rank == 0 else None, dst=0)
    if rank == 0:
        result = {"ok": not problems, "problems": problems, "stats": gathered, "world": world, "steps": steps,
                  "expected": cnt, "mode": mode}
        Path(out).write_text(json.dumps(result))
        print(json.dumps(result))
    dist.barrier()
    dist.destroy_process_group()


def main():
    mode, partition, n, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    kind = sys.argv[5] if len(sys.argv) > 5 else "uniform"
    halo_slot = int(sys.argv[6]) if len(sys.argv) > 6 and int(sys.argv[6]) > 0 else None
    part_slot = int(sys.argv[8]) if len(sys.argv) > 8 and int(sys.argv[8]) > 0 else None
    check = sys.argv[9] if len(sys.argv) > 9 else "brute"      # "single": against the single-GPU path (large scenes)
    adopt = len(sys.argv) > 10 and sys.argv[10] == "adopt"     # feed the owned spheres back as the next step's input
    move = len(sys.argv) > 10 and sys.argv[10] == "move"       # ... and advance every sphere between the steps
    coord_dtype = np.dtype(sys.argv[7]) if len(sys.argv) > 7 else np.dtype("float32")
    import torch  # noqa: F401
    import torch.distributed as dist
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    from collision_amd import hip
    from collision_amd.multi import DistributedCollider, hash_owner

    coords, radii = scene(n, world, kind)
    coords, radii = coords.astype(coord_dtype), radii.astype(coord_dtype)
    gids = np.arange(n, dtype=np.uint32)
    mine = hash_owner(gids, world) == rank
    ctx = hip.Context(0)
    engine = None
    if mode == "cpu":            # (a factory: the rank capacity is agreed over all ranks inside DistributedCollider)
        engine = lambda cap: OracleEngine(cap, 1 << 20, coord_dtype)
    dc = DistributedCollider(ctx, dist, int(mine.sum()), group_size=64, pair_capacity=1 << 22, partition=partition,
                             slack=3.0, engine=engine, halo_slot=halo_slot, coord_dtype=coord_dtype,
                             partition_slot=part_slot)
    dc.set_local_spheres(coords[mine], radii[mine], gids[mine])
    if move:
        return moving_scene(dc, dist, mode, coords, radii, gids, out)
    for _ in range(2):                       # twice: buffers are reused across steps
        dc.step()
    dc.synchronize()
    if adopt:
        # a simulation that advances positions in place: what a rank owns is its next input (nothing moves here, so
        # after the slots have adapted nothing travels: every list is empty and every sphere is 'kept')
        for _ in range(3):
            dc.adopt_owned()
            dc.step()
            dc.synchronize()
        dc.adopt_owned()
        dc.step()
    pairs = dc.local_pairs()
    total = dc.global_pair_count()
    stats = dict(dc.stats)
    stats["repeats"] = dc.repeats
    if mode == "gpu":
        stats["rank_parity"] = per_rank_parity(dc)
    gathered = [None] * world
    dist.gather_object((pairs, stats), gathered if rank == 0 else None, dst=0)
    if rank == 0:
        parity_ok = all(s.get("rank_parity", "ok") == "ok" for _, s in gathered)
        if check == "single":
            # large scenes: the reference pair set is the single-GPU path's (itself oracle-exact: test_pipeline_parity)
            from collision_amd.collision import Collider
            from tests.util import packed_pairs, run_collider
            cq = hip.CommandQueue(ctx)
            cnt, ref = run_collider(ctx, cq, Collider(ctx, n, 64, 256, coord_dtype), coords, radii, 1 << 24)
            assert cnt <= 1 << 24
            got = np.concatenate([p for p, _ in gathered]).reshape(-1, 2)
            g = packed_pairs(np.stack([got.min(axis=1), got.max(axis=1)], axis=1))
            w = packed_pairs(np.stack([ref.min(axis=1), ref.max(axis=1)], axis=1))
            same = len(g) == len(w) and bool((g == w).all())
            unique = int(len(g) - (np.diff(g) == 0).sum()) if len(g) else 0
            result = {"ok": same and unique == len(g) and total == cnt and parity_ok, "expected": cnt, "found": len(g),
                      "unique": unique, "global_count": total}
        else:
            import oracle
            cnt, ref = oracle.brute_force(coords, radii)
            expect = set(map(tuple, ref.tolist()))
            got = []
            for p, _ in gathered:
                got += [tuple(sorted(t)) for t in p.tolist()]
            result = {"ok": len(got) == len(set(got)) == cnt and set(got) == expect and total == cnt and parity_ok,
                      "expected": cnt, "found": len(got), "unique": len(set(got)), "global_count": total}
        result.update({"stats": [s for _, s in gathered], "world": world, "partition": partition, "mode": mode})
        Path(out).write_text(json.dumps(result))
        print(json.dumps(result))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
