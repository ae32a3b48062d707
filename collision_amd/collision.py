"""Broad-phase sphere collision: scene bounds -> Morton codes -> radix sort -> Karras LBVH ->
AABB refit -> pair-overlap traversal.

Mirrors ``collision/collision.py`` (Node :9, NO_NODE :11, CollisionProgram :13-29,
Collider :32-198): same constructor, ``resize``, ``n_nodes``, ``padded_size`` and
``get_collisions(cq, coords_buf, radii_buf, n_collisions_buf, collisions_buf, n_collisions,
wait_for)``, same ``ValueError`` contract.  The ~76 PyOpenCL enqueues of the reference become one
C-ABI call, ``col_collide``, which enqueues the whole chain on the caller's HIP stream.
"""
import ctypes as C

import numpy as np

from . import hip
from ._lib import call
from .bounds import Bounds
from .misc import ProgramHandle, np_float_dtypes, roundUp
from .radix import RadixSorter

# 16-byte node record (collision.py:9, collision.cl:42-53)
Node = np.dtype([("parent", "uint32"), ("right_edge", "uint32"), ("data", "uint32", 2)])
NO_NODE = np.iinfo(np.uint32).max


class CollisionProgram(ProgramHandle):
    """Typed handle (collision.py:13-29): the coordinate dtype the kernels are specialised for."""

    def __init__(self, ctx, coord_dtype=np.dtype("float32")):
        coord_dtype = np.dtype(coord_dtype)
        if coord_dtype.name not in np_float_dtypes:
            raise ValueError("Invalid dtype: {}".format(coord_dtype))
        if coord_dtype.itemsize not in (4, 8):
            raise ValueError("Unsupported coordinate dtype on this device path: {}".format(coord_dtype))
        self.coord_dtype = coord_dtype
        super().__init__(ctx)


class Collider:
    code_dtype = np.dtype("uint32")
    flag_dtype = np.dtype("uint32")
    counter_dtype = np.dtype("uint32")
    id_dtype = np.dtype("uint32")

    def __init__(self, ctx, size, ngroups, group_size, coord_dtype=np.dtype("float32"),
                 program=None, sorter_programs=(None, None), reducer_program=None):
        coord_dtype = np.dtype(coord_dtype)
        self.size = size
        self.group_size = group_size
        self.sorter = RadixSorter(ctx, self.padded_size, group_size, key_dtype=self.code_dtype,
                                  value_dtype=self.id_dtype, program=sorter_programs[0],
                                  scan_program=sorter_programs[1])
        self.reducer = Bounds(ctx, ngroups, group_size, coord_dtype=np.dtype((coord_dtype, 3)),
                              program=reducer_program)
        if program is None:
            program = CollisionProgram(ctx, coord_dtype)
        else:
            if program.context != ctx:
                raise ValueError("Collider and program context must match")
            if program.coord_dtype != coord_dtype:
                raise ValueError("Collider and program coord_dtype must match")
        self.program = program
        self._alloc = {}              # device scratch, (re)allocated at first use after a size change
        # Sort plan for inputs of up to 4 M spheres (include/collision_hip.h, col_collide_plan): the MSD sort
        # is 6 launches shorter but wants every top-digit bucket to fit one workgroup's LDS.  A kernel that
        # meets a larger bucket sorts it anyway (slowly) and says so in a pinned host word; the next calls
        # then take the LSD sort, and every LSD call reports the largest MSD bucket of its codes (read off the
        # sorted codes by the last workgroup of the tree build's first launch): while that says a bucket could not fit, the MSD
        # plan is not tried at all -- a clustered scene pays for ONE slow probe, not one every PLAN_RETRY
        # calls; when it no longer says so, the MSD plan is tried again after PLAN_RETRY calls (doubling up to
        # PLAN_RETRY_MAX while it keeps failing).  No host sync: the words are read when the next call is made
        # (a kernel still in flight may write word 0 after the host cleared it: the worst case is one more LSD
        # period).  sort_plan = "lsd" / "msd" pins the choice (a captured
        # hipGraph replays whichever plan the captured call chose: pin "lsd" for clustered scenes).
        self.sort_plan = "auto"
        # Pair-list allocation of the traversal: "exact" = every wave reserves its 512 staged pairs with one atomic on
        # the pair counter; "chunked" = workgroups reserve 8192 at a time and a small kernel closes the holes (scenes
        # with millions of pairs: one address retires ~88 atomics/us, BASELINE config 3 needs 50 000).  "auto": chunked
        # when the PREVIOUS call on the same counter buffer found more than DENSE_PAIRS pairs (each call publishes
        # the count it is about to zero in a host-visible word: no launch, no sync).
        self.traverse_plan = "auto"
        self._plan_word = None
        self._last_counter, self._same_counter_calls = None, 0
        self._lsd_calls_left = 0
        self._retry_after = self.PLAN_RETRY
        self._tried_msd = False

    # -- sizes -----------------------------------------------------------------
    @property
    def n_nodes(self):
        return 2 * self.size - 1

    @property
    def padded_size(self):
        # the sorter wants a multiple of 2 * group_size (collision.py:125-128)
        return roundUp(self.size, 2 * self.group_size)

    def _allocate(self):
        """(Re)allocate only the scratch whose size changed (collision.py:61-82,104-119)."""
        ctx = self.program.context
        coord_bytes = self.program.coord_dtype.itemsize
        want = {
            "ids0": self.padded_size * 4, "ids1": self.padded_size * 4,
            "codes0": self.padded_size * 4, "codes1": self.padded_size * 4,
            "nodes": self.n_nodes * Node.itemsize,
            "bounds": self.n_nodes * 2 * 4 * coord_bytes,     # also carries the traversal links
            "scratch": call.col_collide_scratch_bytes(self.size, self.padded_size, coord_bytes),
        }
        for name, nbytes in want.items():
            if name not in self._alloc or self._alloc[name].size != nbytes:
                self._alloc[name] = hip.Buffer(ctx, nbytes)
        a = self._alloc
        self._ids_bufs = [a["ids0"], a["ids1"]]
        self._codes_bufs = [a["codes0"], a["codes1"]]
        self._nodes_buf, self._bounds_buf = a["nodes"], a["bounds"]

    @property
    def _flags_buf(self):
        """Arrival counters of the reference's internalBounds (collision.py:147-150): the fused refit
        (csrc/lbvh.hip) does not use them, so they are only allocated if somebody asks (col_bvh_refit)."""
        nbytes = self.n_nodes * 4
        if "flags" not in self._alloc or self._alloc["flags"].size != nbytes:
            self._alloc["flags"] = hip.Buffer(self.program.context, nbytes)
        return self._alloc["flags"]

    def resize(self, size=None, ngroups=None, group_size=None, radix_bits=None):
        if size is not None:
            self.size = size
        if group_size is not None:
            self.group_size = group_size
        # The reference passes roundUp(size, group_size) here (collision.py:95), which fails its
        # own sorter's size rule for odd multiples of group_size; padded_size is what it means.
        self.sorter.resize(self.padded_size, group_size, radix_bits)
        self.reducer.resize(ngroups, group_size)

    # -- the path --------------------------------------------------------------
    def get_collisions(self, cq, coords_buf, radii_buf, n_collisions_buf, collisions_buf, n_collisions,
                       wait_for=None):
        """Enqueue the whole path.  n_collisions_buf (1 uint32) receives the TOTAL number of
        overlapping pairs; the first ``n_collisions`` of them are written to collisions_buf as
        (id, id) rows, in no particular order.  collisions_buf may be None when n_collisions == 0
        (count-only mode, collision.py:134-135)."""
        if collisions_buf is None and n_collisions > 0:
            raise ValueError("Invalid collisions_buf for n_collisions > 0")
        self._allocate()
        cq.wait_for(wait_for)
        # how many calls in a row have used this counter buffer (see _choose_plan: the published pair count is the
        # counter's content BEFORE a call zeroes it, which is a previous result only on a buffer that had one)
        self._same_counter_calls = self._same_counter_calls + 1 if n_collisions_buf.ptr == self._last_counter else 0
        self._last_counter = n_collisions_buf.ptr
        call.col_collide_plan(
            cq.stream, coords_buf.ptr, radii_buf.ptr, self.size, self.padded_size,
            self.program.coord_dtype.itemsize,
            self._codes_bufs[0].ptr, self._codes_bufs[1].ptr, self._ids_bufs[0].ptr, self._ids_bufs[1].ptr,
            self._nodes_buf.ptr, self._bounds_buf.ptr, None, self._alloc["scratch"].ptr,
            n_collisions_buf.ptr, None if collisions_buf is None else collisions_buf.ptr, n_collisions,
            self._choose_plan(n_collisions), self._plan_word)
        return hip.Event(cq)

    PLAN_RETRY, PLAN_RETRY_MAX = 64, 4096
    DENSE_PAIRS = 6000000           # previous pair count from which the traversal allocates in chunks (break-even ~5 M:
                                    # 0.374 / 0.400 ms exact / chunked at 3.3 M pairs, 0.85 / 0.60 ms at 25.4 M)

    def _choose_plan(self, capacity=0):
        """sort_plan argument of col_collide_plan: bit 0 = the sort (0 LSD, 1 MSD), bit 1 = chunked pair allocation.
        Chunked allocation leaves up to 512 x 8192 list slots unused until its last kernel closes them; a list that does
        not have that much room beyond the expected count would have to be rebuilt the exact way, so "auto" only
        chooses it with the room (or in count-only mode, capacity 0)."""
        if self._plan_word is None:      # three host-visible words the device reports into (include/collision_hip.h)
            word = C.c_void_p()
            call.col_host_alloc(C.byref(word), 64)
            self._plan_word = word.value
            for k in range(3):
                C.c_uint32.from_address(self._plan_word + 4 * k).value = 0
        plan = self._choose_sort_plan()
        if self.traverse_plan == "chunked":
            return plan | 2
        if self.traverse_plan == "auto" and self._plan_word:
            last = C.c_uint32.from_address(self._plan_word + 8).value
            # the word is what the counter held when a call zeroed it: on a fresh or rotating counter buffer that is
            # whatever was in memory, so it only counts once the two previous calls used this same buffer (a host
            # running further ahead of the device than that can still see the first call's word: the plan is a
            # choice of speed, never of results)
            if self._same_counter_calls >= 2 and last & 0x80000000 and (last & 0x7FFFFFFF) >= self.DENSE_PAIRS and \
                    (capacity == 0 or capacity >= (last & 0x7FFFFFFF) + 512 * 8192):
                return plan | 2
        return plan

    def _choose_sort_plan(self):
        """0 = LSD, 1 = MSD (see __init__)."""
        if self.sort_plan == "lsd":
            return 0
        if self.sort_plan == "msd":          # pinned: the kernel still reports an oversize bucket (self.oversize_bucket)
            return 1
        flag = C.c_uint32.from_address(self._plan_word)
        if flag.value:                               # an earlier call met a bucket that did not fit
            flag.value = 0
            self._lsd_calls_left = self._retry_after
            self._retry_after = min(2 * self._retry_after, self.PLAN_RETRY_MAX)      # back off while it keeps failing
            self._tried_msd = False
        elif self._tried_msd and not self._lsd_calls_left:
            self._retry_after = self.PLAN_RETRY      # the MSD plan went through: forget the back-off
        if self._lsd_calls_left:
            self._lsd_calls_left -= 1
            return 0
        # the last LSD call's report (word 1): its largest MSD bucket; above the capacity the plan would fail again
        report = C.c_uint32.from_address(self._plan_word + 4).value
        if report & 0x80000000 and (report & 0x7FFFFFFF) > self._msd_bucket_capacity():
            return 0
        self._tried_msd = True
        return 1

    def _msd_bucket_capacity(self):
        """Pairs one workgroup of the MSD plan's LDS finish holds (csrc/radix.hip: k_bucket_sort<8> / <16>)."""
        return 8192 if self.padded_size <= 1900000 else 16384

    @property
    def oversize_bucket(self):
        """Size of a top-digit bucket that did not fit the MSD plan's LDS finish in a call that has completed
        (0 if none was reported since the word was last cleared).  Read it after the queue has been waited on."""
        return C.c_uint32.from_address(self._plan_word).value if self._plan_word else 0

    def __del__(self):
        word, self._plan_word = getattr(self, "_plan_word", None), None
        if word:
            try:
                call.col_host_free(word)
            except Exception:
                pass
