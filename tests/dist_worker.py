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

    def __init__(self, capacity, pair_capacity, ghost_capacity):
        import torch
        import oracle
        self.torch, self.oracle = torch, oracle
        self.device = torch.device("cpu")
        self.capacity, self.pair_capacity, self.ghost_capacity = capacity, pair_capacity, ghost_capacity
        z4 = lambda n: torch.zeros((n, 4), dtype=torch.float32)
        zi = lambda n: torch.zeros(n, dtype=torch.int32)
        self.rows_in, self.gids_in = z4(capacity), zi(capacity)
        z5 = lambda n: torch.zeros((n, 5), dtype=torch.int32)
        self.send5, self.recv5 = z5(capacity), z5(capacity)
        self.owned_rows, self.owned_gids = z4(capacity), zi(capacity)
        self.radii = torch.zeros(capacity)
        self.halo5, self.ghost5 = z5(ghost_capacity), z5(ghost_capacity)
        self.ghost_rows, self.ghost_gids = z4(ghost_capacity), zi(ghost_capacity)
        self.codes_sorted, self.perm = zi(capacity), zi(capacity)
        self.max_peers = 8
        self.sel_lists = zi(self.max_peers * capacity)
        self.found = []
        self.n_owned = 0
        self._boxes = None

    def load(self, coords4, radii, gids):
        n = len(coords4)
        host = np.array(coords4, dtype=np.float32, copy=True)
        host[:, 3] = radii
        self.rows_in[:n] = self.torch.from_numpy(host)
        self.gids_in[:n] = self.torch.from_numpy(np.asarray(gids).astype(np.uint32).view(np.int32))
        return n

    def centre_range(self, rows, n):
        t = self.torch
        if n == 0:
            return t.tensor([np.inf] * 4 + [-np.inf] * 4, dtype=t.float32)
        return t.cat([rows[:n].min(dim=0).values, rows[:n].max(dim=0).values])

    def codes_of(self, rows, n, range8):
        codes = self.oracle.morton(rows[:n].numpy(), range8.numpy().reshape(2, 4))
        self.codes = self.torch.zeros(self.capacity, dtype=self.torch.int32)
        self.codes[:n] = self.torch.from_numpy(codes.view(np.int32))
        return self.codes

    def group_by_owner(self, codes, n, splitters):
        c = codes[:n].numpy().view(np.uint32)
        sp = splitters.numpy().view(np.uint32)
        dest = np.searchsorted(sp, c, side="right")
        perm = np.argsort(dest, kind="stable")
        self.perm[:n] = self.torch.from_numpy(perm.astype(np.int32))
        counts = np.bincount(dest, minlength=len(sp) + 1).astype(np.int32)
        return self.perm, self.torch.from_numpy(counts)

    def pack5(self, rows, gids, idx, idx_offset, n, out5, out_offset=0):
        t = self.torch
        sel = slice(0, n) if idx is None else idx[idx_offset:idx_offset + n].long()
        out5[out_offset:out_offset + n, :4] = rows[sel].view(t.int32)
        out5[out_offset:out_offset + n, 4] = gids[sel]

    def unpack5(self, rec5, n, rows, gids, radii=None):
        t = self.torch
        rows[:n] = rec5[:n, :4].contiguous().view(t.float32)
        gids[:n] = rec5[:n, 4]

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
        self._lo = (self._rows[:, :3] - radii[:, None]).astype(np.float32)
        self._hi = (self._rows[:, :3] + radii[:, None]).astype(np.float32)

    def region_box(self):
        t = self.torch
        if self.n_owned == 0:
            return t.tensor([np.inf] * 4 + [-np.inf] * 4, dtype=t.float32)
        lo, hi = self._lo.min(axis=0), self._hi.max(axis=0)
        return t.tensor([lo[0], lo[1], lo[2], 0, hi[0], hi[1], hi[2], 0], dtype=t.float32)

    def select_multi(self, rows, n, boxes_dev, peers):
        t = self.torch
        counts = t.zeros(self.max_peers, dtype=t.int32)
        boxes = boxes_dev.numpy()
        for k, q in enumerate(peers):
            b = boxes[q]
            hit = ((self._hi > b[0:3]) & (self._lo < b[4:7])).all(axis=1) if n else np.zeros(0, bool)
            idx = np.nonzero(hit)[0].astype(np.int32)
            self.sel_lists[k * self.capacity:k * self.capacity + len(idx)] = t.from_numpy(idx)
            counts[k] = len(idx)
        return self.sel_lists, self.capacity, counts

    def pack5_lists(self, rows, gids, lists, stride, counts_dev, n_lists, n, out5):
        off = 0
        for k in range(n_lists):
            c = int(counts_dev[k])
            self.pack5(rows, gids, lists, k * stride, c, out5, off)
            off += c

    def ghost_queries(self, rows, gids, n_ghost, owned_gids):
        if self.n_owned == 0 or n_ghost == 0:
            return
        g = rows[:n_ghost].numpy()
        glo = (g[:, :3] - g[:, 3:4]).astype(np.float32)
        ghi = (g[:, :3] + g[:, 3:4]).astype(np.float32)
        gg = gids[:n_ghost].numpy().view(np.uint32)
        for k in range(n_ghost):
            hit = ((ghi[k] > self._lo) & (glo[k] < self._hi)).all(axis=1)
            self.found += [(int(gg[k]), int(self._gids[j])) for j in np.nonzero(hit)[0]]

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
    bounds = hip_read(cq, c._bounds_buf, np.float32, (2 * m - 1, 2, 4))
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


def main():
    mode, partition, n, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    kind = sys.argv[5] if len(sys.argv) > 5 else "uniform"
    import torch  # noqa: F401
    import torch.distributed as dist
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    from collision_amd import hip
    from collision_amd.multi import DistributedCollider, hash_owner

    coords, radii = scene(n, world, kind)
    gids = np.arange(n, dtype=np.uint32)
    mine = hash_owner(gids, world) == rank
    ctx = hip.Context(0)
    engine = None
    if mode == "cpu":
        cap = 2 * n + 64
        engine = OracleEngine(cap, 1 << 20, cap * world)
    dc = DistributedCollider(ctx, dist, int(mine.sum()), group_size=64, pair_capacity=1 << 22, partition=partition,
                             slack=3.0, engine=engine)
    dc.set_local_spheres(coords[mine], radii[mine], gids[mine])
    for _ in range(2):                       # twice: buffers are reused across steps
        dc.step()
    dc.synchronize()
    pairs = dc.local_pairs()
    total = dc.global_pair_count()
    stats = dict(dc.stats)
    if mode == "gpu":
        stats["rank_parity"] = per_rank_parity(dc)
    gathered = [None] * world
    dist.gather_object((pairs, stats), gathered if rank == 0 else None, dst=0)
    if rank == 0:
        import oracle
        cnt, ref = oracle.brute_force(coords, radii)
        expect = set(map(tuple, ref.tolist()))
        got = []
        for p, _ in gathered:
            got += [tuple(sorted(t)) for t in p.tolist()]
        parity_ok = all(s.get("rank_parity", "ok") == "ok" for _, s in gathered)
        result = {"ok": len(got) == len(set(got)) == cnt and set(got) == expect and total == cnt and parity_ok,
                  "expected": cnt, "found": len(got), "unique": len(set(got)), "global_count": total,
                  "stats": [s for _, s in gathered], "world": world, "partition": partition, "mode": mode}
        Path(out).write_text(json.dumps(result))
        print(json.dumps(result))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
