"""Multi-GPU broad phase: one process per GPU, torch.distributed (RCCL over xGMI) for the exchanges.

New work -- the reference is single-device.  Spheres arrive partitioned by
``hash(id) mod R`` (BASELINE config 4).  One step on every rank:

1. **AABB all-gather #1** (32 B/rank): min/max of the local centres -> the global scene range, so
   Morton codes mean the same thing on every rank.
2. **Spatial repartition** (``partition="morton"``, default): Morton code per sphere, R-1
   splitters from gathered samples (balanced for clustered scenes too), one radix pass that groups
   the spheres by owner rank, one variable-size all-to-all of 5-word records ``(x, y, z, r, id)``.  Each rank now owns a
   contiguous Morton range, i.e. a compact region.  With ``partition="hash"`` this step is skipped
   and every rank keeps its hash subset (its region is then the whole scene).
3. **Local path**: exactly the single-GPU pipeline (``col_collide``) on the owned spheres; pair ids
   are translated from local indices to global ids.
4. **AABB all-gather #2**: each rank's region box (root of its LBVH, radii included).
5. **Halo exchange**: for every peer that is responsible for the (me -> peer) direction, the
   owned spheres whose box overlaps that peer's region box are packed and sent (all-to-all-v;
   direct peer-to-peer over xGMI, no ring).
6. **Ghost queries**: received spheres are QUERIES against the local tree (never inserted) and
   emit ``(ghost id, local id)`` pairs.

A cross-rank pair {a in r, b in q} is reported by exactly one side: rank r answers the ghosts of
rank q iff ``handles(r, q, R)``.  The union over ranks of the unordered id pairs equals the
single-GPU pair set.

The device work goes through an *engine* object (``HipEngine``: the C ABI on torch CUDA tensors);
the distributed protocol itself only needs ``torch.distributed`` and tensors on ``engine.device``,
so the world_size-2 ``gloo`` tests drive it on the CPU with a test double for the engine.
"""
import ctypes as C

import numpy as np

from . import hip
from ._lib import call
from .misc import roundUp

SAMPLES = 1024          # splitter samples per rank


def handles(r, q, world):
    """True iff rank r answers the ghost spheres of rank q (exactly one of (r,q), (q,r) holds)."""
    if r == q:
        return False
    d = (q - r) % world
    return 2 * d < world or (2 * d == world and r < q)


def hash_owner(gids, world):
    """Initial owner of a sphere: a multiplicative hash of its id (BASELINE config 4)."""
    return ((np.asarray(gids, dtype=np.uint64) * np.uint64(2654435761)) >> np.uint64(7)) % np.uint64(world)


def make_rank_scene(n_per_rank, rank, world, radius, seed=4):
    """Rank-local share of a (world * n_per_rank)-sphere uniform scene (RandomState(seed), as
    BASELINE.md section 5), hash-partitioned by id.  Returns (coords4, radii, gids)."""
    n = n_per_rank * world
    rng = np.random.RandomState(seed)
    coords = rng.random_sample((n, 3)).astype(np.float32)
    gids = np.arange(n, dtype=np.uint32)
    mine = hash_owner(gids, world) == rank
    rows = np.zeros((int(mine.sum()), 4), np.float32)
    rows[:, :3] = coords[mine]
    # same contacts-per-sphere as the 1-GPU scene: shrink r with the density
    r = np.float32(radius * (1.0 / world) ** (1.0 / 3.0))
    return rows, np.full(len(rows), r, np.float32), gids[mine]


# --------------------------------------------------------------------------- exchange layer
class Exchange:
    """The three collectives of the path on tensors of ``device``.  With the ``gloo`` backend
    (CPU tests, or several ranks sharing one GPU in rehearsals) device tensors are staged through
    the host; with ``nccl`` (= RCCL) they go device to device."""

    def __init__(self, dist, device):
        import torch
        self.torch, self.dist, self.device = torch, dist, device
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.stage = dist.get_backend() != "nccl" and device.type != "cpu"

    def all_gather(self, t):
        """[...] -> [world, ...]"""
        torch = self.torch
        if self.dist.get_backend() == "nccl":
            out = torch.empty((self.world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
            self.dist.all_gather_into_tensor(out, t.contiguous())
            return out
        src = t.cpu() if self.stage else t.contiguous()
        out = [torch.empty_like(src) for _ in range(self.world)]
        self.dist.all_gather(out, src)
        return torch.stack(out).to(self.device)

    def exchange_counts(self, send_counts_dev):
        """send_counts_dev: device tensor [world], entry q = rows I send to q.  One all-gather and one
        host sync give both directions: (send_counts, recv_counts) as Python lists."""
        mat = self.all_gather(send_counts_dev).cpu()
        return [int(v) for v in mat[self.rank].tolist()], [int(v) for v in mat[:, self.rank].tolist()]

    def all_to_all_v(self, send, send_counts, recv, recv_counts):
        """Rows grouped by destination in `send`, received grouped by source into `recv`."""
        torch, dist = self.torch, self.dist
        ns, nr = sum(send_counts), sum(recv_counts)
        if dist.get_backend() == "nccl":
            dist.all_to_all_single(recv[:nr], send[:ns], list(recv_counts), list(send_counts))
            return
        s = send[:ns].cpu() if self.stage else send[:ns]
        r = torch.empty((nr,) + tuple(send.shape[1:]), dtype=send.dtype)
        so = np.concatenate([[0], np.cumsum(send_counts)]).astype(int)
        ro = np.concatenate([[0], np.cumsum(recv_counts)]).astype(int)
        r[ro[self.rank]:ro[self.rank + 1]] = s[so[self.rank]:so[self.rank + 1]]
        ops = []
        for q in range(self.world):
            if q == self.rank:
                continue
            if send_counts[q]:
                ops.append(dist.P2POp(dist.isend, s[so[q]:so[q + 1]].contiguous(), q))
            if recv_counts[q]:
                ops.append(dist.P2POp(dist.irecv, r[ro[q]:ro[q + 1]], q))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        recv[:nr] = r.to(recv.device)

    def all_reduce_sum(self, value):
        torch = self.torch
        t = torch.tensor([value], dtype=torch.int64)
        if self.dist.get_backend() == "nccl":
            t = t.to(self.device)
        self.dist.all_reduce(t)
        return int(t.item())


# --------------------------------------------------------------------------- device engine
class ProtocolOps:
    """The small tensor steps between the collectives, written with tensor-library calls.  HipEngine
    replaces each of them by ONE launch through the C ABI; the CPU test double keeps these."""

    def fold_ranges(self, ranges):
        """[R, 8] gathered (min row, max row) boxes -> [8]."""
        torch = self.torch
        return torch.cat([ranges[:, :4].min(dim=0).values, ranges[:, 4:].max(dim=0).values]).contiguous()

    def sample_codes(self, codes, n):
        """SAMPLES evenly strided codes (int32 tensor); an empty rank contributes the code ceiling."""
        torch = self.torch
        if n == 0:
            return torch.full((SAMPLES,), 1 << 30, dtype=torch.int32, device=codes.device)
        pos = (torch.arange(SAMPLES, device=codes.device, dtype=torch.int64) * (n - 1)) // (SAMPLES - 1)
        return codes[pos].contiguous()

    def splitters(self, allsamples, world):
        """world - 1 quantiles of the gathered samples (uint32 order), as an int32 tensor."""
        torch = self.torch
        flat = (allsamples.reshape(-1).to(torch.int64) & 0xFFFFFFFF).sort().values
        cut = torch.arange(1, world, device=flat.device, dtype=torch.int64) * (flat.numel() // world)
        return flat[cut].to(torch.int32).contiguous()

    def expand_counts(self, counts, peers, world):
        """int32[world]: counts[k] at index peers[k], 0 elsewhere."""
        torch = self.torch
        out = torch.zeros(world, dtype=torch.int32, device=counts.device)
        if peers:
            out[torch.tensor(peers, device=counts.device)] = counts[:len(peers)].to(torch.int32)
        return out


class HipEngine(ProtocolOps):
    """Device work of one rank through the C ABI, on torch CUDA tensors (torch = memory + stream)."""

    def __init__(self, ctx, capacity, group_size, pair_capacity, ghost_capacity):
        import torch
        from .collision import Collider
        self.torch = torch
        self.ctx = ctx
        self.device = torch.device("cuda", ctx.device)
        torch.cuda.set_device(self.device)
        self.cq = hip.CommandQueue(ctx, stream=torch.cuda.current_stream().cuda_stream)
        self.capacity, self.pair_capacity, self.ghost_capacity = capacity, pair_capacity, ghost_capacity
        self.group_size = group_size
        f32, i32 = torch.float32, torch.int32
        dev = self.device

        def rows(n):
            return torch.zeros((n, 4), dtype=f32, device=dev)

        def ints(n):
            return torch.zeros(n, dtype=i32, device=dev)

        self.rows_in, self.gids_in = rows(capacity), ints(capacity)
        self.codes, self.codes_sorted, self.iota, self.perm = ints(capacity), ints(capacity), ints(capacity), ints(capacity)
        self.dest = ints(capacity)
        nb_max = -(-capacity // call.col_radix_tile(1, 4, 4))            # the small tile bounds the block count
        self.hist = ints(256 * nb_max)
        self._scan_scratch = hip.Buffer(ctx, call.col_scan_scratch_bytes(256 * nb_max))
        def recs(n):
            return torch.zeros((n, 5), dtype=i32, device=dev)       # transport records (x, y, z, r, gid)

        self.send5, self.recv5 = recs(capacity), recs(capacity)
        self.owned_rows, self.owned_gids = rows(capacity), ints(capacity)
        self.radii = torch.zeros(capacity, dtype=f32, device=dev)
        self.max_peers = 8
        self.sel_lists, self.sel_counts = ints(self.max_peers * capacity), ints(self.max_peers)
        self.halo5, self.ghost5 = recs(ghost_capacity), recs(ghost_capacity)
        self.ghost_rows, self.ghost_gids = rows(ghost_capacity), ints(ghost_capacity)
        self.pairs = torch.zeros((pair_capacity, 2), dtype=i32, device=dev)
        self.counter = ints(1)
        self.range8 = torch.zeros(8, dtype=f32, device=dev)
        self.grange8 = torch.zeros(8, dtype=f32, device=dev)
        self.box8 = torch.zeros(8, dtype=f32, device=dev)
        self.sample, self.split, self.owner_counts, self.rank_counts = ints(SAMPLES), ints(256), ints(256), ints(256)
        self.collider = Collider(ctx, capacity, 64, group_size)
        self.collider._allocate()
        self._reduce_scratch = hip.Buffer(ctx, call.col_reduce_scratch_bytes(0, 4))
        self._sort_scratch = hip.Buffer(ctx, call.col_radix_scratch_bytes(capacity, 4, 4))
        self.n_owned = 0

    # -- inputs
    def load(self, coords4, radii, gids):
        torch = self.torch
        n = len(coords4)
        if n > self.capacity:
            raise ValueError("rank capacity %d < %d local spheres" % (self.capacity, n))
        host = np.array(coords4, dtype=np.float32, copy=True)
        host[:, 3] = radii
        self.rows_in[:n] = torch.from_numpy(host).to(self.device)
        self.gids_in[:n] = torch.from_numpy(np.asarray(gids).astype(np.uint32).view(np.int32)).to(self.device)
        return n

    # -- steps (all asynchronous on the current torch stream)
    def centre_range(self, rows, n):
        """min / max rows (lane w is the radius range, ignored) -> tensor[8]."""
        if n == 0:
            self.range8[:4] = float("inf")
            self.range8[4:] = float("-inf")
        else:
            call.col_reduce(self.cq.stream, rows.data_ptr(), n, 0, 4, 0, self._reduce_scratch.ptr, self.range8.data_ptr())
        return self.range8

    def fold_ranges(self, ranges):
        call.col_fold_boxes(self.cq.stream, ranges.data_ptr(), int(ranges.shape[0]), self.grange8.data_ptr())
        return self.grange8

    def sample_codes(self, codes, n):
        call.col_sample_u32(self.cq.stream, codes.data_ptr(), n, SAMPLES, self.sample.data_ptr())
        return self.sample

    def splitters(self, allsamples, world):
        flat = allsamples.reshape(-1)
        call.col_splitters_u32(self.cq.stream, flat.data_ptr(), int(flat.numel()), world, self.split.data_ptr())
        return self.split[:world - 1]

    def expand_counts(self, counts, peers, world):
        arr = (C.c_int * max(len(peers), 1))(*peers)
        call.col_expand_counts(self.cq.stream, counts.data_ptr(), arr, len(peers), world, self.rank_counts.data_ptr())
        return self.rank_counts[:world]

    def codes_of(self, rows, n, range8):
        """Morton codes of the rows under the global scene range (unsorted) + the index ramp."""
        call.col_morton(self.cq.stream, rows.data_ptr(), range8.data_ptr(), n, n, 4, self.codes.data_ptr(),
                        self.iota.data_ptr())
        return self.codes

    def group_by_owner(self, codes, n, splitters):
        """Stable grouping of the local spheres by destination rank: ONE 8-bit radix pass over the
        owner index (histogram -> scan -> scatter of the production sort) instead of a full sort by
        code.  Returns (perm, counts) on the device: perm lists the spheres owner by owner."""
        torch, s = self.torch, self.cq.stream
        world = int(splitters.numel()) + 1
        call.col_bucketize_u32(s, codes.data_ptr(), n, splitters.data_ptr(), world - 1, self.dest.data_ptr())
        nb = -(-max(n, 1) // call.col_radix_tile(max(n, 1), 4, 4))
        hist = self.hist[:256 * nb]
        if n == 0:
            return self.perm, torch.zeros(world, dtype=torch.int32, device=self.device)
        call.col_radix_histogram(s, self.dest.data_ptr(), n, 4, 4, 0, hist.data_ptr())
        call.col_scan_u32(s, hist.data_ptr(), 256 * nb, self._scan_scratch.ptr)
        call.col_digit_counts(s, hist.data_ptr(), nb, world, n, self.owner_counts.data_ptr())
        call.col_radix_scatter(s, self.dest.data_ptr(), self.codes_sorted.data_ptr(), self.iota.data_ptr(),
                               self.perm.data_ptr(), n, 4, 4, 0, hist.data_ptr())
        return self.perm, self.owner_counts[:world]

    def pack5(self, rows, gids, idx, idx_offset, n, out5, out_offset=0):
        """out5[out_offset + i] = (rows[idx[idx_offset + i]], gids[...]) for i < n (idx None = identity)."""
        call.col_pack5(self.cq.stream, rows.data_ptr(), gids.data_ptr(),
                       None if idx is None else idx.data_ptr() + 4 * idx_offset, n,
                       out5.data_ptr() + 20 * out_offset)

    def unpack5(self, rec5, n, rows, gids, radii=None):
        call.col_unpack5(self.cq.stream, rec5.data_ptr(), n, rows.data_ptr(), gids.data_ptr(),
                         None if radii is None else radii.data_ptr())

    def collide(self, rows, gids, n):
        """Single-GPU path on the owned spheres; pairs come out as global ids."""
        s = self.cq.stream
        self.n_owned = n
        if n == 0:
            self.counter.zero_()
            return                                # (col_collide zeroes the counter itself)
        c = self.collider
        if rows is not self.owned_rows:           # owned rows come out of unpack5 with radii already split off
            call.col_unpack_radii(s, rows.data_ptr(), n, self.radii.data_ptr())
        call.col_collide_plan(s, rows.data_ptr(), self.radii.data_ptr(), n, roundUp(n, 2 * self.group_size), 4,
                              c._codes_bufs[0].ptr, c._codes_bufs[1].ptr, c._ids_bufs[0].ptr, c._ids_bufs[1].ptr,
                              c._nodes_buf.ptr, c._bounds_buf.ptr, None, c._alloc["scratch"].ptr,
                              self.counter.data_ptr(), self.pairs.data_ptr(), self.pair_capacity,
                              c._choose_sort_plan(), c._plan_word)
        call.col_translate_pairs(s, self.pairs.data_ptr(), self.counter.data_ptr(), 0, self.pair_capacity,
                                 gids.data_ptr())

    def region_box(self):
        """Box of everything this rank owns = root of its tree (lo.xyz, -, hi.xyz, -)."""
        if self.n_owned == 0:
            self.box8[:4] = float("inf")
            self.box8[4:] = float("-inf")
        else:
            call.col_memcpy_d2d(self.cq.stream, self.box8.data_ptr(), self.collider._bounds_buf.ptr, 32)
        return self.box8

    def select_multi(self, rows, n, boxes_dev, peers):
        """Halo lists for up to 8 peers in one launch; boxes_dev = the gathered [world, 8] region boxes
        (device).  Returns (lists, stride, counts) -- all on the device, no sync."""
        self.sel_counts.zero_()
        arr = (C.c_int * len(peers))(*peers)
        call.col_select_overlap_multi(self.cq.stream, rows.data_ptr(), n, boxes_dev.data_ptr(), arr, len(peers),
                                      self.capacity, self.sel_lists.data_ptr(), self.sel_counts.data_ptr())
        return self.sel_lists, self.capacity, self.sel_counts

    def pack5_lists(self, rows, gids, lists, stride, counts_dev, n_lists, n, out5):
        call.col_pack5_lists(self.cq.stream, rows.data_ptr(), gids.data_ptr(), lists.data_ptr(), stride,
                             counts_dev.data_ptr(), n_lists, n, out5.data_ptr(), self.ghost_capacity)

    def ghost_queries(self, rows, gids, n_ghost, owned_gids):
        if self.n_owned == 0 or n_ghost == 0:
            return
        call.col_traverse_ghost(self.cq.stream, rows.data_ptr(), gids.data_ptr(), n_ghost,
                                self.collider._bounds_buf.ptr, self.n_owned, owned_gids.data_ptr(),
                                self.pairs.data_ptr(), self.counter.data_ptr(), self.pair_capacity)

    def pair_count(self):
        return int(self.counter.item()) & 0xFFFFFFFF

    def read_pairs(self):
        n = min(self.pair_count(), self.pair_capacity)
        return self.pairs[:n].cpu().numpy().view(np.uint32)

    def synchronize(self):
        self.torch.cuda.current_stream().synchronize()


# --------------------------------------------------------------------------- the protocol
class DistributedCollider:
    def __init__(self, ctx, dist, n_local, group_size=256, pair_capacity=1 << 19, partition="morton",
                 slack=1.6, engine=None, exercise_single_rank=False):
        if partition not in ("morton", "hash"):
            raise ValueError("partition must be 'morton' or 'hash'")
        self.dist, self.partition = dist, partition
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        capacity = roundUp(int(n_local * slack) + 4096, 2 * group_size)
        # hash partition: a rank answers about half of everybody else's spheres
        ghost_capacity = capacity * (max(1, self.world // 2) if partition == "hash" else 1)
        self.engine = engine or HipEngine(ctx, capacity, group_size, pair_capacity, ghost_capacity)
        self.capacity, self.ghost_capacity = capacity, ghost_capacity
        self.x = Exchange(dist, self.engine.device)
        self.cq = getattr(self.engine, "cq", None)
        self.n_in = 0
        self.stats = {}
        # run every exchange even when world_size == 1 (each collective then talks to itself): lets a
        # one-GPU box drive the real RCCL code paths
        self.exercise = exercise_single_rank

    def set_local_spheres(self, coords4, radii, gids):
        self.n_in = self.engine.load(coords4, radii, gids)

    # -- one step ------------------------------------------------------------------------------
    def step(self):
        e, x, R, r = self.engine, self.x, self.world, self.rank
        torch = x.torch
        rows, gids, n = e.rows_in, e.gids_in, self.n_in

        # 1. global scene range of the centres (AABB all-gather #1)
        ranges = x.all_gather(e.centre_range(rows, n))                 # [R, 8]
        grange = e.fold_ranges(ranges)

        # 2. spatial repartition
        if self.partition == "morton" and (R > 1 or self.exercise):
            codes = e.codes_of(rows, n, grange)
            splitters = self._splitters(codes, n)
            perm, counts = e.group_by_owner(codes, n, splitters)
            e.pack5(rows, gids, perm, 0, n, e.send5)
            send_counts, recv_counts = x.exchange_counts(counts)                   # the step's 1st host sync
            m = sum(recv_counts)
            if m > self.capacity:
                raise RuntimeError("rank %d would own %d spheres > capacity %d" % (r, m, self.capacity))
            x.all_to_all_v(e.send5, send_counts, e.recv5, recv_counts)
            e.unpack5(e.recv5, m, e.owned_rows, e.owned_gids, e.radii)
            own_rows, own_gids = e.owned_rows, e.owned_gids
        else:
            own_rows, own_gids, m = rows, gids, n
        self.stats["owned"] = m
        self.own_rows, self.own_gids, self.n_owned = own_rows, own_gids, m      # (tests read these back)

        # 3. the single-GPU path on the owned spheres
        e.collide(own_rows, own_gids, m)
        if R == 1 and not self.exercise:
            return

        # 4. region boxes (AABB all-gather #2); they stay on the device
        boxes = x.all_gather(e.region_box())                           # [R, 8]

        # 5. halo exchange: my boundary spheres go to the peers that answer for me
        peers = [q for q in range(R) if handles(q, r, R)]
        if len(peers) > e.max_peers:
            raise NotImplementedError("more than %d halo peers per rank" % e.max_peers)
        if peers:
            lists, stride, counts = e.select_multi(own_rows, m, boxes, peers)
            e.pack5_lists(own_rows, own_gids, lists, stride, counts, len(peers), m, e.halo5)
            per_rank = e.expand_counts(counts, peers, R)
        else:
            per_rank = torch.zeros(R, dtype=torch.int32, device=boxes.device)
        send_counts, recv_counts = x.exchange_counts(per_rank)                     # the step's 2nd host sync
        g = sum(recv_counts)
        if sum(send_counts) > self.ghost_capacity or g > self.ghost_capacity:
            raise RuntimeError("halo of rank %d (%d out, %d in) exceeds its capacity %d"
                               % (r, sum(send_counts), g, self.ghost_capacity))
        x.all_to_all_v(e.halo5, send_counts, e.ghost5, recv_counts)
        e.unpack5(e.ghost5, g, e.ghost_rows, e.ghost_gids)
        self.stats["ghosts"] = g

        # 6. ghosts as queries against my tree
        e.ghost_queries(e.ghost_rows, e.ghost_gids, g, own_gids)

    def _splitters(self, codes, n):
        """R-1 global quantiles of the Morton codes from SAMPLES strided local samples (the codes are
        in input order, i.e. id-hash order: a strided sample is a random sample)."""
        e = self.engine
        return e.splitters(self.x.all_gather(e.sample_codes(codes, n)), self.world)

    # -- results -------------------------------------------------------------------------------
    def synchronize(self):
        self.engine.synchronize()

    def local_pair_count(self):
        return self.engine.pair_count()

    def global_pair_count(self):
        return self.x.all_reduce_sum(self.engine.pair_count())

    def local_pairs(self):
        """(count, 2) uint32 global ids found by this rank (local x local, then ghost x local)."""
        return self.engine.read_pairs()
