"""Multi-GPU broad phase: one process per GPU, torch.distributed (RCCL over xGMI) for the exchanges.

New work -- the reference is single-device.  Spheres arrive partitioned by
``hash(id) mod R`` (BASELINE config 4).  One step on every rank:

1. **AABB all-gather** (one collective): every rank contributes the min/max of its centres and
   SAMPLES of its spheres.  Folding the boxes gives the global scene range, so Morton codes mean the
   same thing on every rank; the codes of the gathered samples give the splitters.
2. **Spatial repartition** (``partition="morton"``, default): Morton code per sphere, R-1 splitters =
   quantiles of the sample codes (balanced for clustered scenes too), one radix pass that groups the
   spheres by owner rank, a count all-gather (the step's ONE host sync -- the local pipeline needs the
   number of owned spheres on the host; the grouping scatter and the packing run meanwhile) and one
   variable-size all-to-all of transport records ``(x, y, z, r, id)`` (5 words for f32 coordinates, 9 for f64).  Each rank now owns a contiguous
   Morton range, i.e. a compact region.  With ``partition="hash"`` this step is skipped and every rank
   keeps its hash subset (its region is then the whole scene).
3. Two branches run concurrently from here:
   a. **Local path** (main stream): exactly the single-GPU pipeline (``col_collide``) on the owned
      spheres; pair ids are translated from local indices to global ids.
   b. **Halo exchange** (side stream): the rank's region box -- from a bounds reduction over the owned
      spheres, NOT from the tree, so it does not wait for 3a -- goes through the second AABB
      all-gather; one launch selects, for every peer that answers for me, the owned spheres whose box
      overlaps that peer's region, one launch packs them into fixed-size SLOTS (a header record with
      the count, then the records), and one fixed-size all-to-all moves the slots (direct peer-to-peer
      over xGMI, no ring).  No count exchange, no host sync: the counts travel in the headers.
4. **Ghost queries**: received spheres are QUERIES against the local tree (never inserted) and emit
   ``(ghost id, local id)`` pairs; the kernel reads the slot lengths from the headers.

A slot that is too small for its list is detected from its header (``synchronize`` reads the flag): the
slot size is then raised on every rank and the step repeated, so results read after ``synchronize`` are
always exact.  ``synchronize`` also adapts the slot size to 1.5 x the longest list seen.

A cross-rank pair {a in r, b in q} is reported by exactly one side: rank r answers the ghosts of
rank q iff ``handles(r, q, R)``.  The union over ranks of the unordered id pairs equals the
single-GPU pair set.

The device work goes through an *engine* object (``HipEngine``: the C ABI on torch CUDA tensors);
the distributed protocol itself only needs ``torch.distributed`` and tensors on ``engine.device``,
so the world_size-2..8 ``gloo`` tests drive it on the CPU with a test double for the engine.
"""
import ctypes as C

import numpy as np

from . import hip
from ._lib import call
from .misc import roundUp

SAMPLES = 1022          # splitter samples per rank (+ 2 range rows = 1024 rows per rank; 16 ranks fill k_splitters' LDS)
MAX_PEERS = 8           # halo peers per rank (col_select_overlap_multi): world sizes up to 16


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
    """The collectives of the path on tensors of ``device``.  With the ``gloo`` backend (CPU tests, or
    several ranks sharing one GPU in rehearsals) device tensors are staged through the host; with
    ``nccl`` (= RCCL) they go device to device on the stream that is current when they are called."""

    def __init__(self, dist, device):
        import torch
        self.torch, self.dist, self.device = torch, dist, device
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.nccl = dist.get_backend() == "nccl"
        self.stage = not self.nccl and device.type != "cpu"

    def all_gather(self, t):
        """[...] -> [world, ...]"""
        return self.all_gather_finish(self.all_gather_start(t))

    def all_gather_start(self, t):
        """Enqueue the all-gather; with RCCL the current stream does NOT wait for it until
        all_gather_finish, so kernels launched in between overlap it."""
        torch = self.torch
        if self.nccl:
            out = torch.empty((self.world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
            return out, self.dist.all_gather_into_tensor(out, t.contiguous(), async_op=True)
        src = t.cpu() if self.stage else t.contiguous()
        out = [torch.empty_like(src) for _ in range(self.world)]
        self.dist.all_gather(out, src)
        return torch.stack(out).to(self.device), None

    def all_gather_finish(self, started):
        out, work = started
        if work is not None:
            work.wait()                       # the current stream waits for the collective (no host sync)
        return out

    def counts_matrix(self, started):
        """Finish an all-gather of per-destination counts and bring it to the host (the host sync):
        (send_counts, recv_counts) as Python lists."""
        mat = self.all_gather_finish(started).cpu()
        return [int(v) for v in mat[self.rank].tolist()], [int(v) for v in mat[:, self.rank].tolist()]

    def all_to_all_v(self, send, send_counts, recv, recv_counts):
        """Rows grouped by destination in `send`, received grouped by source into `recv`; the counts
        (rows per rank) are host lists."""
        torch, dist = self.torch, self.dist
        ns, nr = sum(send_counts), sum(recv_counts)
        if self.nccl:
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

    def all_reduce(self, value, op="sum"):
        torch = self.torch
        t = torch.tensor([value], dtype=torch.int64)
        if self.nccl:
            t = t.to(self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX if op == "max" else self.dist.ReduceOp.SUM)
        return int(t.item())


# --------------------------------------------------------------------------- device engine
class ProtocolOps:
    """The small tensor steps between the collectives, written with tensor-library calls.  HipEngine
    replaces each of them by ONE launch through the C ABI; the CPU test double keeps these."""

    def sample_and_range(self, rows, n):
        """[SAMPLES + 2, 4]: SAMPLES evenly strided rows, then the min row and the max row of all n."""
        torch = self.torch
        if n == 0:
            inf = float("inf")
            body = torch.tensor([inf, inf, inf, 0.0], dtype=rows.dtype, device=rows.device).repeat(SAMPLES, 1)
            tail = torch.tensor([[inf] * 4, [-inf] * 4], dtype=rows.dtype, device=rows.device)
            return torch.cat([body, tail])
        pos = (torch.arange(SAMPLES, device=rows.device, dtype=torch.int64) * (n - 1)) // (SAMPLES - 1)
        return torch.cat([rows[pos], rows[:n].min(dim=0).values[None], rows[:n].max(dim=0).values[None]]).contiguous()

    def fold_ranges(self, gathered):
        """[R, SAMPLES + 2, 4] -> [8]: min of the min rows, max of the max rows."""
        torch = self.torch
        return torch.cat([gathered[:, SAMPLES].min(dim=0).values, gathered[:, SAMPLES + 1].max(dim=0).values]).contiguous()

    def splitters_from(self, gathered, grange, world):
        """world - 1 quantiles of the Morton codes (uint32 order) of all gathered rows, int32 tensor."""
        torch = self.torch
        codes = self.codes_of_rows(gathered.reshape(-1, 4), grange)
        flat = (codes.to(torch.int64) & 0xFFFFFFFF).sort().values
        cut = torch.arange(1, world, device=flat.device, dtype=torch.int64) * (flat.numel() // world)
        return flat[cut].to(torch.int32).contiguous()

    def region_box(self, rows, n):
        """Box of everything this rank owns, conservative: (min centre - max r, max centre + max r)."""
        torch = self.torch
        if n == 0:
            inf = float("inf")
            return torch.tensor([inf] * 4 + [-inf] * 4, dtype=rows.dtype, device=rows.device)
        mn, mx = rows[:n].min(dim=0).values, rows[:n].max(dim=0).values
        out = torch.cat([mn - mx[3], mx + mx[3]])
        out[3] = 0
        out[7] = 0
        return out.contiguous()


class HipEngine(ProtocolOps):
    """Device work of one rank through the C ABI, on torch CUDA tensors (torch = memory + streams)."""

    def __init__(self, ctx, capacity, group_size, pair_capacity, coord_dtype=np.dtype("float32")):
        import torch
        from .collision import Collider
        self.torch = torch
        self.ctx = ctx
        self.coord_dtype = np.dtype(coord_dtype)
        if self.coord_dtype not in (np.dtype("float32"), np.dtype("float64")):
            raise ValueError("Unsupported coordinate dtype on the multi-GPU path: {}".format(coord_dtype))
        self.cb = self.coord_dtype.itemsize                   # coord_bytes of every C call below
        self.rw = 4 * self.cb // 4 + 1                        # words per transport record: (x, y, z, r) + gid
        self.device = torch.device("cuda", ctx.device)
        torch.cuda.set_device(self.device)
        self.main = torch.cuda.current_stream()
        self.side = torch.cuda.Stream()                       # the halo branch
        self.cq = hip.CommandQueue(ctx, stream=self.main.cuda_stream)
        self.cq_side = hip.CommandQueue(ctx, stream=self.side.cuda_stream)
        self.capacity, self.pair_capacity = capacity, pair_capacity
        self.group_size = group_size
        f32, i32 = (torch.float32 if self.cb == 4 else torch.float64), torch.int32
        dev = self.device
        self._f = f32

        def rows(n):
            return torch.zeros((n, 4), dtype=f32, device=dev)

        def ints(n):
            return torch.zeros(n, dtype=i32, device=dev)

        def recs(n):
            return torch.zeros((n, self.rw), dtype=i32, device=dev)       # transport records (x, y, z, r, gid)

        self._recs = recs
        self.rows_in, self.gids_in = rows(capacity), ints(capacity)
        self.codes, self.codes_sorted, self.iota, self.perm = ints(capacity), ints(capacity), ints(capacity), ints(capacity)
        self.dest = ints(capacity)
        nb_max = -(-capacity // call.col_radix_tile(1, 4, 4))            # the small tile bounds the block count
        self.hist = ints(256 * nb_max)
        self._scan_scratch = hip.Buffer(ctx, call.col_scan_scratch_bytes(256 * nb_max))
        self.send5, self.recv5 = recs(capacity), recs(capacity)
        self.owned_rows, self.owned_gids = rows(capacity), ints(capacity)
        self.radii = torch.zeros(capacity, dtype=f32, device=dev)
        self.sel_lists, self.sel_counts = ints(MAX_PEERS * capacity), ints(MAX_PEERS)
        self.halo_send = self.halo_recv = None
        self.pairs = torch.zeros((pair_capacity, 2), dtype=i32, device=dev)
        self.counter = ints(1)
        self.flags = ints(2)                                   # [longest slot header seen, ghosts queried]
        self.payload = rows(SAMPLES + 2)
        self.minmax8 = torch.zeros(8, dtype=f32, device=dev)
        self.grange8 = torch.zeros(8, dtype=f32, device=dev)
        self.box8 = torch.zeros(8, dtype=f32, device=dev)
        self.sample_codes = ints(16 * (SAMPLES + 2))
        self.split, self.owner_counts = ints(256), ints(256)
        self.collider = Collider(ctx, capacity, 64, group_size, self.coord_dtype)
        self.collider._allocate()
        self._tc = 0 if self.cb == 4 else 1                   # COL_F32 / COL_F64
        self._reduce_scratch = hip.Buffer(ctx, call.col_reduce_scratch_bytes(self._tc, 4))
        self._reduce_scratch_side = hip.Buffer(ctx, call.col_reduce_scratch_bytes(self._tc, 4))
        self.n_owned = 0
        self._nb = 0

    # -- inputs
    def load(self, coords4, radii, gids):
        torch = self.torch
        n = len(coords4)
        if n > self.capacity:
            raise ValueError("rank capacity %d < %d local spheres" % (self.capacity, n))
        host = np.array(coords4, dtype=self.coord_dtype, copy=True)
        host[:, 3] = radii
        self.rows_in[:n] = torch.from_numpy(host).to(self.device)
        self.gids_in[:n] = torch.from_numpy(np.asarray(gids).astype(np.uint32).view(np.int32)).to(self.device)
        return n

    # -- streams
    def begin_step(self):
        self.flags.zero_()

    def fork(self):
        """The halo branch (side stream) starts after everything enqueued on the main stream so far."""
        self.side.wait_stream(self.main)

    def halo_stream(self):
        return self.torch.cuda.stream(self.side)

    def join(self):
        self.main.wait_stream(self.side)

    # -- steps (all asynchronous, on the main stream unless they belong to the halo branch)
    def sample_and_range(self, rows, n):
        s = self.cq.stream
        call.col_sample_rows(s, rows.data_ptr(), n, SAMPLES, self.payload.data_ptr(), self.cb)
        if n == 0:
            self.payload[SAMPLES] = float("inf")
            self.payload[SAMPLES + 1] = float("-inf")
        else:
            call.col_reduce(s, rows.data_ptr(), n, self._tc, 4, 0, self._reduce_scratch.ptr,
                            self.payload.data_ptr() + 4 * self.cb * SAMPLES)
        return self.payload

    def fold_ranges(self, gathered):
        world = int(gathered.shape[0])
        call.col_fold_boxes_strided(self.cq.stream, gathered.data_ptr() + 4 * self.cb * SAMPLES, world, 4 * (SAMPLES + 2),
                                    self.grange8.data_ptr(), self.cb)
        return self.grange8

    def codes_of_rows(self, rows, range8):
        """Morton codes of (m, 4) gathered sample rows under the range."""
        m = int(rows.shape[0])
        if m > self.sample_codes.numel():
            raise ValueError("more than %d sample rows" % self.sample_codes.numel())
        call.col_morton(self.cq.stream, rows.data_ptr(), range8.data_ptr(), m, m, self.cb, self.sample_codes.data_ptr(), None)
        return self.sample_codes[:m]

    def splitters_from(self, gathered, grange, world):
        codes = self.codes_of_rows(gathered.reshape(-1, 4), grange)
        call.col_splitters_u32(self.cq.stream, codes.data_ptr(), int(codes.numel()), world, self.split.data_ptr())
        return self.split[:world - 1]

    def codes_of(self, rows, n, range8):
        """Morton codes of the local rows under the global scene range (unsorted) + the index ramp."""
        call.col_morton(self.cq.stream, rows.data_ptr(), range8.data_ptr(), n, n, self.cb, self.codes.data_ptr(),
                        self.iota.data_ptr())
        return self.codes

    def owner_counts_of(self, codes, n, splitters):
        """First half of the stable grouping by destination rank -- ONE 8-bit radix pass over the owner
        index (histogram -> scan of the production sort) instead of a full sort by code: the number of
        spheres per owner, on the device."""
        torch, s = self.torch, self.cq.stream
        world = int(splitters.numel()) + 1
        if n == 0:
            return torch.zeros(world, dtype=torch.int32, device=self.device)
        call.col_bucketize_u32(s, codes.data_ptr(), n, splitters.data_ptr(), world - 1, self.dest.data_ptr())
        self._nb = nb = -(-n // call.col_radix_tile(n, 4, 4))
        hist = self.hist[:256 * nb]
        call.col_radix_histogram(s, self.dest.data_ptr(), n, 4, 4, 0, hist.data_ptr())
        call.col_scan_u32(s, hist.data_ptr(), 256 * nb, self._scan_scratch.ptr)
        call.col_digit_counts(s, hist.data_ptr(), nb, world, n, self.owner_counts.data_ptr())
        return self.owner_counts[:world]

    def finish_grouping(self, n):
        """Second half: the scatter; perm lists the spheres owner by owner."""
        if n:
            call.col_radix_scatter(self.cq.stream, self.dest.data_ptr(), self.codes_sorted.data_ptr(), self.iota.data_ptr(),
                                   self.perm.data_ptr(), n, 4, 4, 0, self.hist.data_ptr())
        return self.perm

    def pack5(self, rows, gids, idx, idx_offset, n, out5, out_offset=0):
        """out5[out_offset + i] = (rows[idx[idx_offset + i]], gids[...]) for i < n (idx None = identity)."""
        call.col_pack_records(self.cq.stream, rows.data_ptr(), gids.data_ptr(),
                              None if idx is None else idx.data_ptr() + 4 * idx_offset, n,
                              out5.data_ptr() + 4 * self.rw * out_offset, self.cb)

    def unpack5(self, rec5, n, rows, gids, radii=None):
        call.col_unpack_records(self.cq.stream, rec5.data_ptr(), n, rows.data_ptr(), gids.data_ptr(),
                                None if radii is None else radii.data_ptr(), self.cb)

    def collide(self, rows, gids, n):
        """Single-GPU path on the owned spheres (main stream); pairs come out as global ids."""
        s = self.cq.stream
        self.n_owned = n
        if n == 0:
            self.counter.zero_()
            return                                # (col_collide zeroes the counter itself)
        c = self.collider
        if rows is not self.owned_rows:           # owned rows come out of unpack5 with radii already split off
            call.col_unpack_radii(s, rows.data_ptr(), n, self.radii.data_ptr(), self.cb)
        call.col_collide_plan(s, rows.data_ptr(), self.radii.data_ptr(), n, roundUp(n, 2 * self.group_size), self.cb,
                              c._codes_bufs[0].ptr, c._codes_bufs[1].ptr, c._ids_bufs[0].ptr, c._ids_bufs[1].ptr,
                              c._nodes_buf.ptr, c._bounds_buf.ptr, None, c._alloc["scratch"].ptr,
                              self.counter.data_ptr(), self.pairs.data_ptr(), self.pair_capacity,
                              c._choose_sort_plan(), c._plan_word)
        call.col_translate_pairs(s, self.pairs.data_ptr(), self.counter.data_ptr(), 0, self.pair_capacity,
                                 gids.data_ptr())

    # -- halo branch (side stream)
    def region_box(self, rows, n):
        s = self.cq_side.stream
        if n == 0:
            self.box8[:4] = float("inf")
            self.box8[4:] = float("-inf")
        else:
            call.col_reduce(s, rows.data_ptr(), n, self._tc, 4, 0, self._reduce_scratch_side.ptr, self.minmax8.data_ptr())
            call.col_region_box(s, self.minmax8.data_ptr(), self.box8.data_ptr(), self.cb)
        return self.box8

    def ensure_slots(self, slot, n_out, n_in):
        """Send / receive buffers of the slotted halo exchange: (slot + 1) records per peer."""
        want_s, want_r = max(1, n_out) * (slot + 1), max(1, n_in) * (slot + 1)
        if self.halo_send is None or self.halo_send.shape[0] != want_s:
            self.halo_send = self._recs(want_s)
        if self.halo_recv is None or self.halo_recv.shape[0] != want_r:
            self.halo_recv = self._recs(want_r)

    def select_and_pack(self, rows, gids, n, boxes_dev, peers, slot):
        """Halo lists for up to 8 peers in one launch (boxes_dev = the gathered [world, 8] region boxes, on
        the device), packed into one slot per peer by a second launch.  No sync: counts stay on the device."""
        s = self.cq_side.stream
        self.sel_counts.zero_()
        if not peers:
            return
        arr = (C.c_int * len(peers))(*peers)
        call.col_select_overlap_multi(s, rows.data_ptr(), n, boxes_dev.data_ptr(), arr, len(peers),
                                      self.capacity, self.sel_lists.data_ptr(), self.sel_counts.data_ptr(), self.cb)
        call.col_pack_slots(s, rows.data_ptr(), gids.data_ptr(), self.sel_lists.data_ptr(), self.capacity,
                            self.sel_counts.data_ptr(), len(peers), min(max(n, 1), slot), self.halo_send.data_ptr(),
                            int(self.halo_send.shape[0]), slot, self.cb)

    def ghost_queries(self, n_in, slot, owned_gids):
        if self.n_owned == 0 or n_in == 0:
            return
        call.col_traverse_ghost_slots(self.cq.stream, self.halo_recv.data_ptr(), n_in, slot,
                                      self.collider._bounds_buf.ptr, self.n_owned, owned_gids.data_ptr(),
                                      self.pairs.data_ptr(), self.counter.data_ptr(), self.pair_capacity,
                                      self.flags.data_ptr(), self.cb)

    # -- results (these synchronise)
    def halo_stats(self):
        """(longest halo list of this rank, sent or received; ghosts queried).  Host sync."""
        f = self.flags.cpu().numpy().view(np.uint32)
        sent = int(self.sel_counts.cpu().numpy().view(np.uint32).max())
        return max(int(f[0]), sent), int(f[1])

    def pair_count(self):
        return int(self.counter.item()) & 0xFFFFFFFF

    def read_pairs(self):
        n = min(self.pair_count(), self.pair_capacity)
        return self.pairs[:n].cpu().numpy().view(np.uint32)

    def synchronize(self):
        self.side.synchronize()
        self.main.synchronize()


# --------------------------------------------------------------------------- the protocol
class DistributedCollider:
    def __init__(self, ctx, dist, n_local, group_size=256, pair_capacity=1 << 19, partition="morton",
                 slack=1.6, engine=None, exercise_single_rank=False, halo_slot=None, coord_dtype=np.dtype("float32")):
        if partition not in ("morton", "hash"):
            raise ValueError("partition must be 'morton' or 'hash'")
        self.dist, self.partition = dist, partition
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        capacity = roundUp(int(n_local * slack) + 4096, 2 * group_size)
        self.engine = engine or HipEngine(ctx, capacity, group_size, pair_capacity, coord_dtype)
        self.capacity = capacity
        self.x = Exchange(dist, self.engine.device)
        self.cq = getattr(self.engine, "cq", None)
        self.n_in = 0
        self.stats = {}
        # run every exchange even when world_size == 1 (each collective then talks to itself): lets a
        # one-GPU box drive the real RCCL code paths
        self.exercise = exercise_single_rank
        r, R = self.rank, self.world
        self.peers_out = [q for q in range(R) if handles(q, r, R)]      # they answer for me: my halo goes there
        self.peers_in = [q for q in range(R) if handles(r, q, R)]       # I answer for them: their halo comes here
        if max(len(self.peers_out), len(self.peers_in)) > MAX_PEERS:
            raise NotImplementedError("more than %d halo peers per rank" % MAX_PEERS)
        # records per halo slot: a hash partition sends a peer everything (its region is the whole scene),
        # a spatial one a thin shell -- synchronize() adapts it to what the scene needs
        self.slot = capacity if partition == "hash" else max(4096, roundUp(capacity // 6, 1024))
        if halo_slot is not None:
            self.slot = int(halo_slot)           # (tests: start too small and let synchronize() repair it)
        self.repeats = 0                         # steps repeated because a slot overflowed
        self._dirty = False

    def set_local_spheres(self, coords4, radii, gids):
        self.n_in = self.engine.load(coords4, radii, gids)

    # -- one step ------------------------------------------------------------------------------
    def step(self):
        e, x, R, r = self.engine, self.x, self.world, self.rank
        rows, gids, n = e.rows_in, e.gids_in, self.n_in
        self._dirty = True
        e.begin_step()

        # 1. ONE all-gather: every rank's centre range and sample rows -> global scene range, splitters
        gathered = x.all_gather(e.sample_and_range(rows, n))           # [R, SAMPLES + 2, 4]
        grange = e.fold_ranges(gathered)

        # 2. spatial repartition
        if self.partition == "morton" and (R > 1 or self.exercise):
            splitters = e.splitters_from(gathered, grange, R)
            codes = e.codes_of(rows, n, grange)
            counts = e.owner_counts_of(codes, n, splitters)
            started = x.all_gather_start(counts)
            perm = e.finish_grouping(n)                                # these two overlap the count exchange ...
            e.pack5(rows, gids, perm, 0, n, e.send5)
            send_counts, recv_counts = x.counts_matrix(started)        # ... and the step's one host sync
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

        halo = R > 1 or self.exercise
        if halo:
            # 3b. halo branch on the side stream, concurrent with the local pipeline below
            slot = self.slot
            e.ensure_slots(slot, len(self.peers_out), len(self.peers_in))
            e.fork()
            with e.halo_stream():
                boxes = x.all_gather(e.region_box(own_rows, m))        # [R, 8], stays on the device
                e.select_and_pack(own_rows, own_gids, m, boxes, self.peers_out, slot)
                out_rows = [slot + 1 if q in self.peers_out else 0 for q in range(R)]
                in_rows = [slot + 1 if q in self.peers_in else 0 for q in range(R)]
                x.all_to_all_v(e.halo_send, out_rows, e.halo_recv, in_rows)

        # 3a. the single-GPU path on the owned spheres
        e.collide(own_rows, own_gids, m)

        if halo:
            # 4. ghosts as queries against my tree (slot lengths are read from the headers on the device)
            e.join()
            e.ghost_queries(len(self.peers_in), slot, own_gids)

    # -- results -------------------------------------------------------------------------------
    def synchronize(self):
        """Wait for the enqueued steps.  If a halo slot overflowed in the last step (seen in its header),
        every rank raises the slot size and the step is repeated, so what is read afterwards is exact;
        otherwise the slot size follows 1.5 x the longest list any rank has seen."""
        e = self.engine
        e.synchronize()
        if not self._dirty or not (self.world > 1 or self.exercise):
            self._dirty = False
            return
        while True:
            longest, ghosts = e.halo_stats()
            longest = self.x.all_reduce(longest, "max")
            self.stats["ghosts"] = ghosts
            self.stats["halo_slot"] = self.slot
            want = max(4096, roundUp(longest + longest // 2 + 1024, 1024))
            if longest <= self.slot:
                if self.partition != "hash" and (want < self.slot // 2 or want > self.slot):
                    self.slot = min(want, roundUp(self.capacity, 1024))
                break
            self.slot = min(want, roundUp(self.capacity, 1024))
            self.repeats += 1
            self.step()
            e.synchronize()
        self._dirty = False

    def local_pair_count(self):
        self.synchronize()
        return self.engine.pair_count()

    def global_pair_count(self):
        self.synchronize()
        return self.x.all_reduce(self.engine.pair_count())

    def local_pairs(self):
        """(count, 2) uint32 global ids found by this rank (local x local, then ghost x local)."""
        self.synchronize()
        return self.engine.read_pairs()
