"""Multi-GPU broad phase: one process per GPU, torch.distributed (RCCL over xGMI) for the exchanges.

New work -- the reference is single-device.  Spheres arrive partitioned by
``hash(id) mod R`` (BASELINE config 4).  One step on every rank:

1. **AABB all-gather** (one collective, ``partition="morton"``): every rank contributes the min/max of
   its centres and SAMPLES of its spheres (one launch).  Folding the boxes gives the global scene range, so
   Morton codes mean the same thing on every rank; the codes of the gathered samples give the splitters.
2. **Spatial repartition** (``partition="morton"``, default): R-1 splitters = quantiles of the sample codes
   (balanced for clustered scenes too); owner of every sphere and the per-tile owner histogram (one launch,
   the codes are never stored), scanned; ONE launch groups and packs the spheres in a single pass over the rows:
   what the rank keeps goes STRAIGHT into its owned arrays (it never travels), the rest into one fixed-size
   SLOT per other rank -- a header record with the list's length, then transport records ``(x, y, z, r, id)``
   (5 words for f32 coordinates, 9 for f64).  ONE fixed-size all-to-all moves the slots: no count exchange;
   the host enqueues all of this without waiting for anything.  The unpack launch appends the received
   spheres to the owned arrays and publishes the owned count in a host-visible word: the step's ONE host
   wait (the local pipeline's launch sizes depend on it) is a poll of that word.  Each rank now owns a
   contiguous Morton range, i.e. a compact region.  With ``partition="hash"`` steps 1-2 are skipped and every
   rank keeps its hash subset (its region is then the whole scene).
3. Two branches run concurrently from here:
   a. **Local path** (main stream, enqueued first): exactly the single-GPU pipeline (``col_collide``) on the
      owned spheres; pair ids are translated from local indices to global ids.
   b. **Halo exchange** (side stream): the rank's region box -- one bounds-reduction launch over the owned
      spheres, NOT from the tree, so it does not wait for 3a -- goes through the second AABB
      all-gather; one launch selects, for every peer that answers for me, the owned spheres whose box
      overlaps that peer's region, one launch packs them into fixed-size slots as above, and one fixed-size
      all-to-all moves the slots (direct peer-to-peer over xGMI, no ring).
4. **Ghost queries**: received spheres are QUERIES against the local tree (never inserted) and emit
   ``(ghost id, local id)`` pairs; the kernel reads the slot lengths from the headers.

A HALO slot that is too small for its list is detected from its header (``synchronize`` reads the flags): the
slot size is then raised on every rank and the step repeated, so results read after ``synchronize`` are always
exact.  A REPARTITION slot that is too small loses nothing: what does not fit stays with the sending rank (which
rank owns a sphere only decides load balance and halo size, never the pair set), so ``adopt_owned`` is safe
without a host check; ``synchronize`` lets both slot sizes follow the longest lists seen.

The protocol is written once, as a generator that yields at each collective (``_step_gen``): ``step()`` drives
it with the rank's ``Exchange`` (torch.distributed); ``LoopbackWorld`` drives the generators of R ranks that live
in ONE process in lockstep and copies between their buffers -- config 4's eight ranks on a one-GPU box.

A cross-rank pair {a in r, b in q} is reported by exactly one side: rank r answers the ghosts of
rank q iff ``handles(r, q, R)``.  The union over ranks of the unordered id pairs equals the
single-GPU pair set.

The device work goes through an *engine* object (``HipEngine``: the C ABI on torch CUDA tensors);
the distributed protocol itself only needs ``torch.distributed`` and tensors on ``engine.device``,
so the world_size-2..8 ``gloo`` tests drive it on the CPU with a test double for the engine.
"""
import ctypes as C
import os

import numpy as np

from . import hip
from ._lib import call
from .misc import roundUp

REGION_BOXES = 8        # boxes per rank region (one per octant of the scene: COL_REGION_BOXES)
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
        self._gathered = {}              # all-gather outputs by input buffer (nccl)
        self._a2a_views = {}             # (send view, recv view, split lists) by buffers and counts (nccl)

    def all_gather(self, t):
        """[...] -> [world, ...].  With RCCL a synchronous op: it is enqueued on the CURRENT stream (no hand-off to the
        process group's own stream and back: two cross-stream event waits of 10-15 us each per collective on this
        platform).  The output tensor of a given input buffer is kept (the host side of a step is nearly as long as
        its device side at 1 M spheres: every allocation counts)."""
        torch = self.torch
        if self.nccl:
            key = (t.data_ptr(), tuple(t.shape), t.dtype)
            out = self._gathered.get(key)
            if out is None:
                out = self._gathered[key] = torch.empty((self.world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
            self.dist.all_gather_into_tensor(out, t)
            return out
        src = t.cpu() if self.stage else t.contiguous()
        out = [torch.empty_like(src) for _ in range(self.world)]
        self.dist.all_gather(out, src)
        return torch.stack(out).to(self.device)

    def all_to_all_v(self, send, send_counts, recv, recv_counts):
        """Rows grouped by destination in `send`, received grouped by source into `recv`; the counts
        (rows per rank) are host lists."""
        torch, dist = self.torch, self.dist
        if self.nccl:
            key = (send.data_ptr(), recv.data_ptr(), tuple(send_counts), tuple(recv_counts))
            v = self._a2a_views.get(key)
            if v is None:
                if len(self._a2a_views) > 64:
                    self._a2a_views.clear()
                v = self._a2a_views[key] = (recv[:sum(recv_counts)], send[:sum(send_counts)], list(recv_counts), list(send_counts))
            dist.all_to_all_single(*v)
            return
        ns, nr = sum(send_counts), sum(recv_counts)
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
        """An int, or a list of ints reduced element-wise in one collective."""
        torch = self.torch
        many = isinstance(value, (list, tuple))
        t = torch.tensor(list(value) if many else [value], dtype=torch.int64)
        if self.nccl:
            t = t.to(self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX if op == "max" else self.dist.ReduceOp.SUM)
        out = [int(v) for v in t.cpu().tolist()]
        return out if many else out[0]


# --------------------------------------------------------------------------- device engine
class ProtocolOps:
    """The device steps of the protocol written with tensor-library calls (what the CPU test double runs).
    HipEngine replaces each of them by one C-ABI call of one to three launches.  An engine provides the
    buffers (owned_rows, owned_gids, part_send, part_recv, ...), codes_of_rows, pack5 and unpack5."""

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
        """world - 1 quantiles of the Morton codes (uint32 order) of all gathered rows, int64 tensor."""
        torch = self.torch
        codes = self.codes_of_rows(gathered.reshape(-1, 4), grange)
        flat = (codes.to(torch.int64) & 0xFFFFFFFF).sort().values
        cut = torch.arange(1, world, device=flat.device, dtype=torch.int64) * (flat.numel() // world)
        return flat[cut].contiguous()

    def partition_plan(self, gathered, rows, n, world):
        """Global range, splitters, owner of every row (number of splitters <= its code), rows per owner."""
        torch = self.torch
        grange = self._grange = self.fold_ranges(gathered)
        splitters = self.splitters_from(gathered, grange, world)
        codes = self.codes_of_rows(rows[:n], grange).to(torch.int64) & 0xFFFFFFFF
        self._dest = torch.searchsorted(splitters, codes, right=True)
        self._owner_counts = torch.bincount(self._dest, minlength=world)

    def ensure_partition_slots(self, slot, world):
        """Send / receive buffers of the slotted repartition: (slot + 1) records per other rank."""
        want = max(1, world - 1) * (slot + 1)
        if self.part_send is None or self.part_send.shape[0] != want:
            self.part_send, self.part_recv = self._recs(want), self._recs(want)

    def partition_group(self, rows, gids, n, world, rank, slot):
        """Stable grouping by owner; kept rows to the front of the owned arrays, the others into their slots; what
        does not fit into a slot stays here, behind the kept rows (owner order)."""
        torch = self.torch
        perm = torch.sort(self._dest, stable=True).indices
        starts = [0] + [int(v) for v in torch.cumsum(self._owner_counts, 0).tolist()]
        off = starts[rank + 1] - starts[rank]
        idx = perm[starts[rank]:starts[rank + 1]]
        self.owned_rows[:off] = rows[idx]
        self.owned_gids[:off] = gids[idx]
        for q in range(world):
            if q == rank:
                continue
            idx = perm[starts[q]:starts[q + 1]]
            cnt = int(idx.numel())
            base = (q if q < rank else q - 1) * (slot + 1)
            self.part_send[base] = 0
            self.part_send[base, 0] = cnt
            self._longest_part = max(self._longest_part, cnt)
            self.pack5(rows, gids, idx.to(torch.int32), 0, min(cnt, slot), self.part_send, base + 1)
            if cnt > slot:
                extra = idx[slot:][:max(0, self.capacity - off)]
                k = int(extra.numel())
                self.owned_rows[off:off + k] = rows[extra]
                self.owned_gids[off:off + k] = gids[extra]
                off += cnt - slot
        self._kept = off

    def partition_unpack(self, world, rank, slot):
        """Received slots, in rank order, behind the rows that stayed; the owned count m."""
        off = self._kept
        for k in range(world - 1):
            base = k * (slot + 1)
            length = int(self.part_recv[base, 0])
            self._longest_part = max(self._longest_part, length)
            cnt = min(length, slot)
            take = max(0, min(cnt, self.capacity - off))
            self.unpack5(self.part_recv[base + 1:base + 1 + take], take, self.owned_rows[off:], self.owned_gids[off:])
            off += cnt
        self._m = off

    def owned_count(self):
        return self._m

    def region_boxes(self, rows, n, repartitioned):
        """[REGION_BOXES, 8]: this rank's region = one conservative box (min centre - max r, 0, max centre + max r,
        0) per octant of the global scene range (the top three bits of the Morton code) over the owned spheres of that octant (an empty octant: an inverted
        box).  A Morton range is a compact piece inside an octant; one box around a range that spills over an
        octant boundary by a few spheres would cover a quarter of the scene.  Without a repartition: one box."""
        torch = self.torch
        inf = float("inf")
        out = torch.tensor([inf] * 3 + [0.0] + [-inf] * 3 + [0.0], dtype=rows.dtype, device=rows.device).repeat(REGION_BOXES, 1)
        if n == 0:
            return out
        r = rows[:n]
        octant = torch.zeros(n, dtype=torch.int64, device=rows.device)
        if repartitioned:
            octant = (self.codes_of_rows(r, self._grange).to(torch.int64) & 0xFFFFFFFF) >> 27
        for o in range(REGION_BOXES):
            sel = r[octant == o]
            if sel.shape[0]:
                mn, mx = sel.min(dim=0).values, sel.max(dim=0).values
                out[o, :3], out[o, 4:7] = mn[:3] - mx[3], mx[:3] + mx[3]
        return out.contiguous()


class HipEngine(ProtocolOps):
    """Device work of one rank through the C ABI, on torch CUDA tensors (torch = memory + streams)."""

    OWNED_COUNT_TIMEOUT_S = 600

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
        # the halo branch.  A stream of the other priority class: torch hands out its pooled streams round-robin and
        # HIP maps them onto a few hardware queues -- the first pooled stream of a process shared the queue of the
        # current stream, and the branch then ran AFTER the local pipeline instead of beside it (+50 us per step)
        self.side = torch.cuda.Stream(priority=-1)
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
            # transport records (x, y, z, r, gid).  NOT filled: a fill launch on the allocating stream could land
            # after the other stream's pack launch; every slot header is written by the pack launches and nothing
            # behind a list's length is ever read
            return torch.empty((n, self.rw), dtype=i32, device=dev)

        self._recs = recs
        self.rows_in, self.gids_in = rows(capacity), ints(capacity)
        self.dest = ints(capacity)
        nb_max = -(-capacity // call.col_radix_tile(1, 4, 4))            # the small tile bounds the block count
        self.hist = ints(256 * nb_max)
        self.owned_rows, self.owned_gids = rows(capacity), ints(capacity)
        self.radii = torch.zeros(capacity, dtype=f32, device=dev)
        self.sel_lists, self.sel_counts = ints(MAX_PEERS * capacity), ints(MAX_PEERS)
        self.part_send = self.part_recv = None
        self.halo_send = self.halo_recv = None
        self._ghost_scratch = None
        self.pairs = torch.zeros((pair_capacity, 2), dtype=i32, device=dev)
        self.counter = ints(1)
        self.flags = ints(4)               # [longest halo header seen, ghosts queried, longest repartition list, -]
        self.owned2 = ints(2)              # [min(m, capacity), m] of the last repartition
        self.payload = rows(SAMPLES + 2)
        self.grange8 = torch.zeros(8, dtype=f32, device=dev)
        self.boxes = torch.zeros((REGION_BOXES, 8), dtype=f32, device=dev)
        self.split, self.owner_counts = ints(256), ints(256)
        self.collider = Collider(ctx, capacity, 64, group_size, self.coord_dtype)
        self.collider._allocate()
        nscratch = call.col_partition_scratch_bytes()
        self._range_scratch = torch.zeros(nscratch, dtype=torch.uint8, device=dev)        # main stream
        self._range_scratch_side = torch.zeros(nscratch, dtype=torch.uint8, device=dev)   # halo branch
        self._bounds_partials = torch.zeros(256 * 8 * self.cb, dtype=torch.uint8, device=dev)     # col_minmax4_stage1_dev
        self._bounds_parts = C.c_uint32(0)
        self._have_partials = False
        self._early_partials = not os.environ.get("COLLISION_NO_EARLY_PARTIALS")
        # DEVICE-SIDE OWNED COUNT (round 4): after a repartition every launch that needs the number of owned spheres reads it
        # from the device word the unpack leaves (owned2[0] = min(m, capacity)) with the rank capacity as its host-known bound
        # (col_collide_plan_dev and the *_dev entry points of include/collision_hip.h), so a step is enqueue-only: the host
        # never waits for m.  owned_count() reads the same number from the host-visible word when somebody asks for it
        # (synchronize(), adopt_owned(), tests).  COLLISION_HOST_OWNED_COUNT=1: the round-3 behaviour, a poll inside the step.
        self.device_count = self._early_partials and not os.environ.get("COLLISION_HOST_OWNED_COUNT")
        # the bound the device-count launches are sized for: the rank capacity at first, then a few per cent above the counts
        # seen (DistributedCollider.synchronize(): a bound of 1.3 n sorts 30 % pads and takes the next tile class)
        self.run_bound = capacity
        word = C.c_void_p()
        call.col_host_alloc(C.byref(word), 64)
        self._host_word = word.value                           # host-visible: (step number << 32 | owned count)
        C.c_uint64.from_address(self._host_word).value = 0
        self._seq = 0
        self._fork_event = torch.cuda.Event()
        self._owned_event = torch.cuda.Event()
        self.n_owned = 0
        # pointers of the persistent buffers (the per-step C calls then convert nothing but a few integers)
        self._p = {k: getattr(self, k).data_ptr() for k in
                   ("rows_in", "gids_in", "dest", "hist", "owned_rows", "owned_gids",
                    "radii", "sel_lists", "sel_counts", "pairs", "counter", "flags", "owned2", "payload", "grange8",
                    "boxes", "split", "owner_counts", "_range_scratch", "_range_scratch_side", "_bounds_partials")}

    def __del__(self):
        word, self._host_word = getattr(self, "_host_word", None), None
        if word:
            try:
                call.col_host_free(word)
            except Exception:
                pass

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

    def swap_input_and_owned(self):
        """The owned arrays become the input arrays and vice versa (stream-ordered: nothing is copied)."""
        self.rows_in, self.owned_rows = self.owned_rows, self.rows_in
        self.gids_in, self.owned_gids = self.owned_gids, self.gids_in
        for k in ("rows_in", "gids_in", "owned_rows", "owned_gids"):
            self._p[k] = getattr(self, k).data_ptr()

    # -- streams
    def begin_step(self, sampled):
        """Per-step flags are cleared by the sample launch when there is one."""
        if not sampled:
            self.flags.zero_()

    def mark_fork(self):
        """The halo branch may start after everything enqueued on the main stream SO FAR."""
        self._fork_event.record(self.main)

    def fork(self):
        self.side.wait_event(self._fork_event)

    def halo_stream(self):
        return self.torch.cuda.stream(self.side)

    def join(self):
        self.main.wait_stream(self.side)

    # -- repartition (main stream, nothing here waits for the device)
    def sample_and_range(self, rows, n):
        p = self._p
        call.col_partition_sample(self.cq.stream, rows.data_ptr(), n, SAMPLES, p["payload"], p["_range_scratch"],
                                  p["flags"], 4, self.cb)
        return self.payload

    def partition_plan(self, gathered, rows, n, world):
        p = self._p
        call.col_partition_plan(self.cq.stream, gathered.data_ptr(), world, SAMPLES, rows.data_ptr(), n, p["grange8"],
                                p["split"], p["dest"], p["hist"], p["owner_counts"], self.cb)

    def partition_group(self, rows, gids, n, world, rank, slot):
        p = self._p
        call.col_partition_group(self.cq.stream, rows.data_ptr(), gids.data_ptr(), n, p["dest"], p["hist"],
                                 p["owner_counts"], world, rank, slot, self.part_send.data_ptr(), p["owned_rows"],
                                 p["owned_gids"], p["radii"], self.capacity, p["flags"], self.cb)

    def partition_unpack(self, world, rank, slot):
        p = self._p
        self._seq = (self._seq % 0xFFFFFFF0) + 1
        call.col_partition_unpack(self.cq.stream, self.part_recv.data_ptr(), world, rank, slot, p["owner_counts"],
                                  p["owned_rows"], p["owned_gids"], p["radii"], self.capacity, p["owned2"],
                                  self._host_word, self._seq, p["flags"], self.cb)
        # the first launch of the local pipeline (block partials of the scene bounds) does not need m on the host: it
        # reads the owned count from the device, so the device is busy while the host polls and enqueues the rest
        self._owned_event.record(self.main)
        if self._early_partials:
            call.col_minmax4_stage1_dev(self.cq.stream, p["owned_rows"], p["owned2"], self.capacity, self.cb,
                                        p["_bounds_partials"], C.byref(self._bounds_parts))
            self._have_partials = True

    def owned_count(self):
        """The step's one host wait: the unpack launch writes (step number << 32 | m) into a host-visible word."""
        word, seq, ev = C.c_uint64.from_address(self._host_word), self._seq, self._owned_event
        spins, t0 = 0, None
        while (word.value >> 32) != seq:
            spins += 1
            if (spins & 255) == 0:
                if ev.query():                                 # the launch has completed: its store is visible
                    if (word.value >> 32) != seq:
                        raise RuntimeError("the repartition did not publish its owned count")
                    break
                if (spins & 0xFFFFF) == 0:                     # (a peer that never joins the exchange must not hang this rank for ever)
                    import time
                    t0 = t0 or time.monotonic()
                    if time.monotonic() - t0 > self.OWNED_COUNT_TIMEOUT_S:
                        raise RuntimeError("no owned count after %d s: the repartition exchange did not complete"
                                           % self.OWNED_COUNT_TIMEOUT_S)
        return int(word.value & 0xFFFFFFFF)

    # -- local path (main stream)
    def collide(self, rows, gids, n):
        """Single-GPU path on the owned spheres; pairs come out as global ids."""
        s = self.cq.stream
        self.n_owned = n
        c, p = self.collider, self._p
        if n is None:                             # the count is on the device (owned2[0]): the rank capacity is the bound
            if not (rows is self.owned_rows and self._have_partials):
                raise RuntimeError("a device-side owned count needs the repartition's early bounds partials")
            self._have_partials = False
            call.col_collide_plan_dev(s, rows.data_ptr(), p["radii"], self.run_bound, roundUp(self.run_bound, 2 * self.group_size), self.cb,
                                      c._codes_bufs[0].ptr, c._codes_bufs[1].ptr, c._ids_bufs[0].ptr, c._ids_bufs[1].ptr,
                                      c._nodes_buf.ptr, c._bounds_buf.ptr, None, c._alloc["scratch"].ptr,
                                      p["counter"], p["pairs"], self.pair_capacity, c._choose_plan(self.pair_capacity), c._plan_word,
                                      p["_bounds_partials"], self._bounds_parts.value, p["owned2"])
            call.col_translate_pairs(s, p["pairs"], p["counter"], 0, self.pair_capacity, gids.data_ptr())
            return
        if n == 0:
            self.counter.zero_()
            return                                # (col_collide zeroes the counter itself)
        if rows is not self.owned_rows:           # owned rows come out of the repartition with radii already split off
            call.col_unpack_radii(s, rows.data_ptr(), n, p["radii"], self.cb)
        partials = p["_bounds_partials"] if (rows is self.owned_rows and self._have_partials) else None
        self._have_partials = False
        call.col_collide_plan_partials(s, rows.data_ptr(), p["radii"], n, roundUp(n, 2 * self.group_size), self.cb,
                                       c._codes_bufs[0].ptr, c._codes_bufs[1].ptr, c._ids_bufs[0].ptr, c._ids_bufs[1].ptr,
                                       c._nodes_buf.ptr, c._bounds_buf.ptr, None, c._alloc["scratch"].ptr,
                                       p["counter"], p["pairs"], self.pair_capacity, c._choose_plan(self.pair_capacity), c._plan_word,
                                       partials, self._bounds_parts.value if partials else 0)
        call.col_translate_pairs(s, p["pairs"], p["counter"], 0, self.pair_capacity, gids.data_ptr())

    # -- halo branch (side stream)
    def region_boxes(self, rows, n, repartitioned):
        """One launch; also clears the halo list counters."""
        p = self._p
        if n is None:                             # (device-side owned count: see __init__)
            call.col_region_boxes_dev(self.cq_side.stream, rows.data_ptr(), self.run_bound, p["grange8"] if repartitioned else None,
                                      p["_range_scratch_side"], p["boxes"], p["sel_counts"], MAX_PEERS, self.cb, p["owned2"])
            return self.boxes
        call.col_region_boxes(self.cq_side.stream, rows.data_ptr(), n, p["grange8"] if repartitioned else None,
                              p["_range_scratch_side"], p["boxes"], p["sel_counts"], MAX_PEERS, self.cb)
        return self.boxes

    def ensure_slots(self, slot, n_out, n_in):
        """Send / receive buffers of the slotted halo exchange: (slot + 1) records per peer."""
        want_s, want_r = max(1, n_out) * (slot + 1), max(1, n_in) * (slot + 1)
        if self.halo_send is None or self.halo_send.shape[0] != want_s:
            self.halo_send = self._recs(want_s)
        if self.halo_recv is None or self.halo_recv.shape[0] != want_r:
            self.halo_recv = self._recs(want_r)

    def select_and_pack(self, rows, gids, n, boxes_dev, peers, slot):
        """Halo lists for up to 8 peers in one launch (boxes_dev = the gathered [world, 8, 8] region boxes, on
        the device), packed into one slot per peer by a second launch.  No sync: counts stay on the device."""
        if not peers:
            return
        s, p = self.cq_side.stream, self._p
        arr = (C.c_int * len(peers))(*peers)
        if n is None:                             # (device-side owned count: see __init__)
            call.col_select_overlap_multi_dev(s, rows.data_ptr(), self.run_bound, boxes_dev.data_ptr(), arr, len(peers),
                                              self.capacity, p["sel_lists"], p["sel_counts"], self.cb, p["owned2"])
            n = self.run_bound
        else:
            call.col_select_overlap_multi(s, rows.data_ptr(), n, boxes_dev.data_ptr(), arr, len(peers),
                                          self.capacity, p["sel_lists"], p["sel_counts"], self.cb)
        call.col_pack_slots(s, rows.data_ptr(), gids.data_ptr(), p["sel_lists"], self.capacity,
                            p["sel_counts"], len(peers), min(max(n, 1), slot), self.halo_send.data_ptr(),
                            int(self.halo_send.shape[0]), slot, self.cb)

    GHOST_PACKETS_FROM = 400000     # expected ghosts from which the packet walk (ordering them first) beats the lane walk

    def ghost_queries(self, n_in, slot, owned_gids, expected=None):
        """`expected`: how many ghosts the caller expects (the last step's count; None = unknown).  Many ghosts (a hash
        partition: millions) are ordered by a coarse Morton key and walk the tree as packets of 64 neighbours; a thin
        halo (a Morton repartition: 10^4..10^5) walks lane by lane -- ordering every slot entry would cost more than it
        saves (measured at 8 x 2 M spheres: 0.75-0.96 instead of 2.1-2.8 ms for 6-8 M ghosts, but 0.19 instead of
        0.06 ms for 25-128 k)."""
        if self.n_owned == 0 or n_in == 0:
            return
        p = self._p
        dev = self.n_owned is None                # (device-side owned count: see __init__)
        scratch = None
        want_packets = os.environ.get("COLLISION_GHOST_WALK", "auto")
        if want_packets == "packets" or (want_packets == "auto" and (expected is None or expected >= self.GHOST_PACKETS_FROM)
                                         and n_in * slot >= self.GHOST_PACKETS_FROM):
            # scratch for the packet walk: coarse Morton keys + record numbers of every slot entry, sorted (csrc/multi.hip)
            need = call.col_ghost_scratch_bytes(n_in, slot)
            if self._ghost_scratch is None or self._ghost_scratch.numel() < need:
                # written and read on the main stream only; this may run while the halo branch's side stream is torch's
                # current one (_drive), and the caching allocator ties a block to the stream that is current when it is made
                with self.torch.cuda.stream(self.main):
                    self._ghost_scratch = self.torch.empty(need, dtype=self.torch.uint8, device=self.device)
            scratch = self._ghost_scratch.data_ptr()
        if dev:
            call.col_traverse_ghost_slots_dev(self.cq.stream, self.halo_recv.data_ptr(), n_in, slot,
                                              self.collider._bounds_buf.ptr, self.run_bound, owned_gids.data_ptr(),
                                              p["pairs"], p["counter"], self.pair_capacity, p["flags"], self.cb, scratch, p["owned2"])
            return
        call.col_traverse_ghost_slots(self.cq.stream, self.halo_recv.data_ptr(), n_in, slot,
                                      self.collider._bounds_buf.ptr, self.n_owned, owned_gids.data_ptr(),
                                      p["pairs"], p["counter"], self.pair_capacity, p["flags"], self.cb, scratch)

    # -- results (these synchronise)
    def halo_stats(self):
        """(longest halo list of this rank, sent or received; ghosts queried; longest repartition list).  Host sync."""
        f = self.flags.cpu().numpy().view(np.uint32)
        sent = int(self.sel_counts.cpu().numpy().view(np.uint32).max())
        return max(int(f[0]), sent), int(f[1]), int(f[2])

    def pair_count(self):
        return int(self.counter.item()) & 0xFFFFFFFF

    def read_pairs(self):
        n = min(self.pair_count(), self.pair_capacity)
        return self.pairs[:n].cpu().numpy().view(np.uint32)

    def synchronize(self):
        self.side.synchronize()
        self.main.synchronize()


# --------------------------------------------------------------------------- the protocol
def _agree_max(dist, values, ctx):
    """Element-wise maximum of a few host integers over all ranks (sizes every rank must agree on)."""
    import torch
    t = torch.tensor([int(v) for v in values], dtype=torch.int64)
    if dist.get_backend() == "nccl":
        t = t.to(torch.device("cuda", ctx.device))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return [int(v) for v in t.cpu().tolist()]


class DistributedCollider:
    """One rank of the multi-GPU broad phase.

    ``n_local`` may differ from rank to rank (a hash partition is balanced to +-0.1 % only); every buffer size of the
    fixed-size exchanges -- rank capacity, halo slot, repartition slot -- is derived from the MAXIMUM over the ranks,
    agreed by one all-reduce in the constructor (``n_agreed`` skips it: the caller already knows the maximum).
    ``engine`` is a device engine, or a callable ``capacity -> engine`` (the CPU test double), or None (HipEngine).
    ``exchange`` replaces the torch.distributed ``Exchange`` (LoopbackWorld: several ranks in one process)."""

    MIN_PARTITION_SLOT = 1024      # records; synchronize() never shrinks a repartition slot below this

    def __init__(self, ctx, dist, n_local, group_size=256, pair_capacity=1 << 19, partition="morton",
                 slack=1.6, engine=None, exercise_single_rank=False, halo_slot=None, coord_dtype=np.dtype("float32"),
                 partition_slot=None, exchange=None, n_agreed=None):
        if partition not in ("morton", "hash"):
            raise ValueError("partition must be 'morton' or 'hash'")
        self.dist, self.partition = dist, partition
        if exchange is not None:
            self.rank, self.world = exchange.rank, exchange.world
            if n_agreed is None:
                raise ValueError("an external exchange needs n_agreed (the largest n_local of all ranks)")
        else:
            self.rank, self.world = dist.get_rank(), dist.get_world_size()
        if n_agreed is None:
            n_agreed = _agree_max(dist, [n_local], ctx)[0] if self.world > 1 else int(n_local)
        if n_local > n_agreed:
            raise ValueError("n_local %d > the agreed maximum %d" % (n_local, n_agreed))
        self.n_agreed = n_agreed = int(n_agreed)
        capacity = roundUp(int(n_agreed * slack) + 4096, 2 * group_size)
        if callable(engine):
            engine = engine(capacity)
        self.engine = engine or HipEngine(ctx, capacity, group_size, pair_capacity, coord_dtype)
        if self.engine.capacity != capacity:
            raise ValueError("engine capacity %d != the agreed rank capacity %d" % (self.engine.capacity, capacity))
        self.capacity = capacity
        self.x = exchange if exchange is not None else Exchange(dist, self.engine.device)
        self.cq = getattr(self.engine, "cq", None)
        self.n_in = 0
        self.stats = {"partition_overflows": 0}
        # run every exchange even when world_size == 1 (each collective then talks to itself): lets a
        # one-GPU box drive the real RCCL code paths
        self.exercise = exercise_single_rank
        r, R = self.rank, self.world
        if partition == "morton" and R * (SAMPLES + 2) > 16384:
            raise NotImplementedError("the Morton repartition gathers %d sample rows per rank: at most %d ranks"
                                      % (SAMPLES + 2, 16384 // (SAMPLES + 2)))
        self.peers_out = [q for q in range(R) if handles(q, r, R)]      # they answer for me: my halo goes there
        self.peers_in = [q for q in range(R) if handles(r, q, R)]       # I answer for them: their halo comes here
        if max(len(self.peers_out), len(self.peers_in)) > MAX_PEERS:
            raise NotImplementedError("more than %d halo peers per rank" % MAX_PEERS)
        # records per halo slot: a hash partition sends a peer everything (its region is the whole scene),
        # a spatial one a thin shell -- synchronize() adapts it to what the scene needs
        self.slot = capacity if partition == "hash" else max(4096, roundUp(capacity // 6, 1024))
        if halo_slot is not None:
            self.slot = int(halo_slot)           # (tests: start too small and let synchronize() repair it)
        # records per repartition slot (one per other rank): spheres that arrive hash-partitioned leave for every
        # rank in equal shares; synchronize() follows the longest list seen (a spatially coherent arrival
        # sends little, and what a rank keeps does not travel at all)
        self.part_slot = max(1024, roundUp(int(1.3 * n_agreed / R) + 1024, 1024)) if R > 1 else 1024
        if partition_slot is not None:
            self.part_slot = int(partition_slot)
        self.repeats = 0                         # steps repeated because a halo slot overflowed
        self._splits, self._halo_splits = {}, {}
        self._dirty = False
        self.own_rows = self.own_gids = None
        self.n_owned = 0

    @property
    def n_owned(self):
        """Spheres this rank owns after the last step.  With a device-side count (HipEngine.device_count) step() does not know
        it: the first reader takes it from the host-visible word the repartition's unpack wrote (a poll that returns at once
        when the step has run -- synchronize() asks -- and otherwise waits for the unpack, nothing more)."""
        if self._n_owned is None:
            m = self.engine.owned_count()
            if m > self.capacity:
                raise RuntimeError("rank %d would own %d spheres > capacity %d" % (self.rank, m, self.capacity))
            self._n_owned = m
            self.engine.n_owned = m
        self.stats["owned"] = self._n_owned
        return self._n_owned

    @n_owned.setter
    def n_owned(self, m):
        self._n_owned = m
        if m is not None:
            self.stats["owned"] = m

    def set_local_spheres(self, coords4, radii, gids):
        self.n_in = self.engine.load(coords4, radii, gids)

    def adopt_owned(self):
        """What this rank owns after the last step() becomes its input for the next one: a simulation that advances
        positions where the spheres live (``local_rows()`` / ``local_gids()`` are the arrays to advance).  Nothing is
        copied (the engine swaps its input and owned arrays, in stream order) and nothing waits.  Safe without looking
        at the last step's flags: a repartition slot that was too small left its surplus with the sender, so the
        owned sets of all ranks are always a partition of the scene; a halo slot that was too small only affects the
        pairs of that step, which synchronize() repairs by repeating it on the (identical) adopted input.
        Spheres that have not moved far stay with their rank: they are 'kept' by the next repartition and never
        travel, and synchronize() lets the repartition slots shrink to what still moves."""
        if self.own_rows is None or self.own_rows is self.engine.rows_in:
            return                            # (no repartition ran: the input already is what the rank owns)
        self.engine.swap_input_and_owned()
        self.n_in = self.n_owned
        self.own_rows, self.own_gids = self.engine.rows_in, self.engine.gids_in

    def local_rows(self):
        """[n_in, 4] (x, y, z, r) rows of the next step's input, on the engine's device (a view: advance in place)."""
        return self.engine.rows_in[:self.n_in]

    def local_gids(self):
        """[n_in] global ids (int32 bit patterns of uint32) of the next step's input."""
        return self.engine.gids_in[:self.n_in]

    # -- one step ------------------------------------------------------------------------------
    def _drive(self, gen):
        """Run a protocol generator against this rank's Exchange: every yield is (collective, stream, args...)."""
        x, e = self.x, self.engine
        side, reply = None, None
        try:
            while True:
                req = gen.send(reply)
                if req[1] == "side":
                    if side is None:             # RCCL enqueues on the CURRENT stream: the halo branch's collectives
                        side = e.halo_stream()   # run on the side stream (entered once, left when the step ends)
                        side.__enter__()
                elif side is not None:
                    side.__exit__(None, None, None)
                    side = None
                reply = getattr(x, req[0])(*req[2:])
        except StopIteration as stop:
            return stop.value
        finally:
            if side is not None:
                side.__exit__(None, None, None)

    def step(self):
        self._drive(self._step_gen())

    def _step_gen(self):
        e, R, r = self.engine, self.world, self.rank
        rows, gids, n = e.rows_in, e.gids_in, self.n_in
        self._dirty = True
        several = R > 1 or self.exercise
        repartition = self.partition == "morton" and several
        e.begin_step(repartition)

        if repartition:
            # 1. ONE all-gather: every rank's centre range and sample rows -> global scene range, splitters
            gathered = yield ("all_gather", "main", e.sample_and_range(rows, n))      # [R, SAMPLES + 2, 4]
            # 2. spatial repartition: fixed-size slots, nothing here waits for the device
            pslot = self.part_slot
            e.ensure_partition_slots(pslot, R)
            e.partition_plan(gathered, rows, n, R)
            e.partition_group(rows, gids, n, R, r, pslot)
            split = self._splits.get(pslot)
            if split is None:
                split = self._splits[pslot] = tuple(pslot + 1 if q != r else 0 for q in range(R))
            yield ("all_to_all_v", "main", e.part_send, split, e.part_recv, split)
            e.partition_unpack(R, r, pslot)
            e.mark_fork()
            own_rows, own_gids = e.owned_rows, e.owned_gids
            if getattr(e, "device_count", False):
                m = None                                               # on the device; the host reads it when asked (n_owned)
            else:
                m = e.owned_count()                                    # (engines without it: the step's one host wait)
                if m > self.capacity:
                    raise RuntimeError("rank %d would own %d spheres > capacity %d" % (r, m, self.capacity))
        else:
            e.mark_fork()
            own_rows, own_gids, m = rows, gids, n
        self.own_rows, self.own_gids, self.n_owned = own_rows, own_gids, m      # (tests read these back)

        # 3a. the single-GPU path on the owned spheres, enqueued first: the device works on it while the host
        #     enqueues the halo branch
        e.collide(own_rows, own_gids, m)

        if several:
            # 3b. halo branch on the side stream, concurrent with the local pipeline
            slot = self.slot
            e.ensure_slots(slot, len(self.peers_out), len(self.peers_in))
            e.fork()
            boxes = yield ("all_gather", "side", e.region_boxes(own_rows, m, repartition))   # [R, 8 boxes, 8], on the device
            e.select_and_pack(own_rows, own_gids, m, boxes, self.peers_out, slot)
            rows_io = self._halo_splits.get(slot)
            if rows_io is None:
                rows_io = self._halo_splits[slot] = (tuple(slot + 1 if q in self.peers_out else 0 for q in range(R)),
                                                     tuple(slot + 1 if q in self.peers_in else 0 for q in range(R)))
            yield ("all_to_all_v", "side", e.halo_send, rows_io[0], e.halo_recv, rows_io[1])
            # 4. ghosts as queries against my tree (slot lengths are read from the headers on the device)
            e.join()
            e.ghost_queries(len(self.peers_in), slot, own_gids, self.stats.get("ghosts"))

    # -- results -------------------------------------------------------------------------------
    def synchronize(self):
        """Wait for the enqueued steps.  If a halo slot overflowed in the last step (seen in its header), every
        rank raises the slot size and the step is repeated, so what is read afterwards is exact; otherwise the
        slot sizes follow the longest lists any rank has seen (a repartition slot that overflowed lost nothing --
        the surplus stayed with its sender -- and simply grows for the steps that follow)."""
        self._drive(self._sync_gen())

    def _sync_gen(self):
        e = self.engine
        e.synchronize()
        _ = self.n_owned                     # (a device-side owned count becomes known to the host here)
        if not self._dirty or not (self.world > 1 or self.exercise):
            self._dirty = False
            return
        cap = roundUp(self.capacity, 1024)
        while True:
            longest, ghosts, longest_part = e.halo_stats()
            # (device-side owned count: a rank that owns more than the bound its launches were sized for worked on the first
            # `bound` of its spheres only -- every rank repeats the step with the bound raised, like a halo slot that overflowed)
            bound_over = 0
            if getattr(e, "device_count", False) and self.partition == "morton":
                m, bound = self.n_owned, e.run_bound
                bound_over = 1 if m > bound else 0
                want_b = min(self.capacity, roundUp(m + m // 32 + 2048, 2 * e.group_size))
                if m > bound or want_b > bound or 10 * want_b < 9 * bound:
                    e.run_bound = want_b
            longest, longest_part, bound_over = yield ("all_reduce", "main", [longest, longest_part, bound_over], "max")
            # (both are all-reduced: every rank takes the same decisions below and leaves together)
            if max(longest, longest_part) > cap:     # (cannot happen: a list is a subset of what one rank holds)
                raise RuntimeError("a list of %d records cannot fit any slot (rank capacity %d)"
                                   % (max(longest, longest_part), self.capacity))
            self.stats["ghosts"] = ghosts
            self.stats["halo_slot"], self.stats["partition_slot"] = self.slot, self.part_slot
            want = max(4096, roundUp(longest + longest // 2 + 1024, 1024))
            floor_p = self.MIN_PARTITION_SLOT
            want_p = max(floor_p, roundUp(longest_part + longest_part // 16 + 2 * floor_p, floor_p))   # padding travels: keep it small
            if longest_part > self.part_slot:
                self.part_slot = min(want_p, cap)
                self.stats["partition_overflows"] += 1
            elif 20 * want_p <= 17 * self.part_slot:        # (shrink with hysteresis: lists that breathe by a few % keep their slot)
                self.part_slot = want_p
            again = bool(bound_over)
            if longest > self.slot:
                self.slot, again = min(want, cap), True
            elif self.partition != "hash" and (want < self.slot // 2 or want > self.slot):
                self.slot = min(want, cap)
            self.stats["halo_slot_next"], self.stats["partition_slot_next"] = self.slot, self.part_slot
            if not again:
                break
            self.repeats += 1
            yield from self._step_gen()
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


# --------------------------------------------------------------------------- several ranks in one process
class LoopbackPort:
    """The 'exchange' of one rank of a LoopbackWorld: it only names the rank; the world moves the data."""
    nccl = False

    def __init__(self, rank, world):
        self.rank, self.world = rank, world

    def _refuse(self, *a, **k):
        raise RuntimeError("a LoopbackWorld rank is driven by the world (world.step() / world.synchronize())")

    all_gather = all_to_all_v = all_reduce = _refuse


class LoopbackWorld:
    """R ranks of the protocol in ONE process on ONE device, for rehearsing a multi-GPU configuration on a one-GPU
    box (BASELINE config 4: world 8 x 2 M spheres).  Every rank is a full DistributedCollider with its own engine
    (buffers, tree, streams); the world runs the ranks' protocol generators in lockstep -- all ranks up to their
    next collective, then the collective as plain copies between the ranks' buffers -- so each rank executes exactly
    the launches, slot layouts and host decisions it would execute under torch.distributed.  What it cannot show is
    the wire: collective latency and xGMI bandwidth.  With ``timed=True`` every segment between two collectives is
    run to completion rank by rank and its device time recorded (``segments[rank] = [(ends_with, ms), ...]``): the
    per-rank critical path of a step minus the collectives."""

    def __init__(self, ctx, world, n_locals, engine=None, **kw):
        import torch
        self.torch = torch
        n_agreed = max(int(v) for v in n_locals)
        self.ranks = [DistributedCollider(ctx, None, int(n_locals[r]), engine=engine, exchange=LoopbackPort(r, world),
                                          n_agreed=n_agreed, **kw) for r in range(world)]
        self.world = world
        self.segments = [[] for _ in range(world)]
        self._gpu = self.ranks[0].engine.device.type != "cpu"

    def _device_sync(self):
        if self._gpu:
            self.torch.cuda.synchronize()

    def _run(self, gens, timed=False):
        R, torch = self.world, self.torch
        replies = [None] * R
        if timed:
            self.segments = [[] for _ in range(R)]
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        while True:
            reqs = []
            for r, g in enumerate(gens):
                e = self.ranks[r].engine
                if timed:
                    self._device_sync()
                    ev[0].record(e.main)
                try:
                    reqs.append(g.send(replies[r]))
                except StopIteration:
                    reqs.append(None)
                if timed:
                    ev[1].record(e.main)
                    ev[2].record(e.side)
                    self._device_sync()
                    self.segments[r].append((reqs[-1][0] if reqs[-1] else "end",
                                             max(ev[0].elapsed_time(ev[1]), ev[0].elapsed_time(ev[2]))))
            done = [q is None for q in reqs]
            if all(done):
                return
            if any(done) or len({q[0] for q in reqs}) != 1:
                raise RuntimeError("the ranks left the protocol at different points: %r" % [q and q[0] for q in reqs])
            self._device_sync()
            replies = getattr(self, "_" + reqs[0][0])([q[2:] for q in reqs])
            self._device_sync()

    # the collectives, as copies between the ranks' buffers (same layouts as Exchange)
    def _all_gather(self, args):
        out = self.torch.stack([a[0] for a in args])
        return [out] * self.world

    def _all_to_all_v(self, args):
        R = self.world
        offs = [(np.concatenate([[0], np.cumsum(a[1])]).astype(int), np.concatenate([[0], np.cumsum(a[3])]).astype(int))
                for a in args]
        for r in range(R):
            recv, rc, ro = args[r][2], args[r][3], offs[r][1]
            for q in range(R):
                send, sc, so = args[q][0], args[q][1], offs[q][0]
                if sc[r] != rc[q]:
                    raise RuntimeError("rank %d sends %d records to rank %d, which expects %d" % (q, sc[r], r, rc[q]))
                if rc[q]:
                    recv[ro[q]:ro[q + 1]] = send[so[r]:so[r + 1]]
        return [None] * R

    def _all_reduce(self, args):
        vals = [a[0] if isinstance(a[0], (list, tuple)) else [a[0]] for a in args]
        op = max if (len(args[0]) > 1 and args[0][1] == "max") else sum
        out = [int(op(v[k] for v in vals)) for k in range(len(vals[0]))]
        return [out if isinstance(args[r][0], (list, tuple)) else out[0] for r in range(self.world)]

    # the world's view of the collider API
    def set_local_spheres(self, rank, coords4, radii, gids):
        self.ranks[rank].set_local_spheres(coords4, radii, gids)

    def step(self, timed=False):
        self._run([dc._step_gen() for dc in self.ranks], timed)

    def synchronize(self):
        self._run([dc._sync_gen() for dc in self.ranks])

    def adopt_owned(self):
        for dc in self.ranks:
            dc.adopt_owned()

    def pairs(self):
        """Every rank's pairs (global ids), after a synchronize()."""
        self.synchronize()
        return [dc.engine.read_pairs() for dc in self.ranks]

    def global_pair_count(self):
        self.synchronize()
        return sum(dc.engine.pair_count() for dc in self.ranks)
