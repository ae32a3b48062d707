"""Per-stage device timing of the path (HIP events on the launch stream).  Measurement aid for
bench.py and profiling; runs the same C-ABI entry points col_collide chains together."""
import numpy as np

from . import hip
from ._lib import call


def _timed(cq, fn, reps):
    start, stop = hip.Event(), hip.Event()
    fn()
    cq.finish()
    call.col_event_record(start.handle, cq.stream)
    for _ in range(reps):
        fn()
    call.col_event_record(stop.handle, cq.stream)
    stop.wait()
    return start.elapsed_ms(stop) / reps


def stage_times(hip_mod, ctx, cq, collider, coords_buf, radii_buf, n_buf, pairs_buf, capacity, reps=20):
    """Returns {stage: ms}.  Stages run in pipeline order once first, so each timed stage sees the
    real outputs of its predecessors."""
    c = collider
    c._allocate()
    n, p, cb = c.size, c.padded_size, c.program.coord_dtype.itemsize
    s = cq.stream
    codes0, codes1 = c._codes_bufs
    ids0, ids1 = c._ids_bufs
    nodes, bounds, flags = c._nodes_buf, c._bounds_buf, c._flags_buf
    red_scratch = hip.Buffer(ctx, call.col_reduce_scratch_bytes(0 if cb == 4 else 1, 4))
    rng = hip.Buffer(ctx, 256)
    sort_scratch = hip.Buffer(ctx, call.col_radix_scratch_bytes(p, 4, 4))
    tile = call.col_radix_tile(p, 4, 4)
    nb = -(-p // tile)
    hist = hip.Buffer(ctx, 256 * nb * 4)
    scan_scratch = hip.Buffer(ctx, call.col_scan_scratch_bytes(256 * nb))
    zero = np.zeros(1, np.uint32)

    def f_bounds():
        call.col_reduce(s, coords_buf.ptr, n, 0 if cb == 4 else 1, 4, 0, red_scratch.ptr, rng.ptr)

    def f_morton():
        call.col_morton(s, coords_buf.ptr, rng.ptr, n, p, cb, codes0.ptr, ids0.ptr)

    def f_sort():
        call.col_radix_sort(s, codes0.ptr, codes1.ptr, ids0.ptr, ids1.ptr, p, 4, 4, sort_scratch.ptr, 0)

    def f_hist():
        call.col_radix_histogram(s, codes0.ptr, p, 4, 4, 0, hist.ptr)

    def f_scan():
        call.col_scan_u32(s, hist.ptr, 256 * nb, scan_scratch.ptr)

    def f_scatter():
        call.col_radix_scatter(s, codes0.ptr, codes1.ptr, ids0.ptr, ids1.ptr, p, 4, 4, 0, hist.ptr)

    def f_build():
        call.col_bvh_build(s, codes1.ptr, ids1.ptr, nodes.ptr, bounds.ptr, n, cb)

    def f_refit():
        call.col_fill(s, flags.ptr, zero.ctypes.data, 4, 2 * n - 1)
        call.col_bvh_refit(s, bounds.ptr, flags.ptr, coords_buf.ptr, radii_buf.ptr, nodes.ptr, n, cb)

    lbvh_scratch = hip.Buffer(ctx, call.col_lbvh_scratch_bytes(n, cb))

    def f_lbvh():
        call.col_lbvh(s, codes1.ptr, ids1.ptr, coords_buf.ptr, radii_buf.ptr, nodes.ptr, bounds.ptr,
                      lbvh_scratch.ptr, n, cb)

    def f_traverse():
        call.col_fill(s, n_buf.ptr, zero.ctypes.data, 4, 1)
        call.col_traverse(s, pairs_buf.ptr, n_buf.ptr, capacity, nodes.ptr, bounds.ptr, n, cb)

    def f_all():
        c.get_collisions(cq, coords_buf, radii_buf, n_buf, pairs_buf, capacity)

    out = {}
    for name, fn in (("bounds", f_bounds), ("morton", f_morton), ("sort", f_sort),
                     ("generic_build", f_build), ("generic_refit", f_refit), ("lbvh_build_refit", f_lbvh),
                     ("traverse", f_traverse)):
        out[name] = round(_timed(cq, fn, reps), 4)
    # one radix pass, split
    f_hist(); f_scan()
    out["sort_pass_hist"] = round(_timed(cq, f_hist, reps), 4)
    out["sort_pass_scatter"] = round(_timed(cq, f_scatter, reps), 4)   # offsets from the scan above
    out["sort_pass_scan"] = round(_timed(cq, f_scan, reps), 4)
    out["whole_path"] = round(_timed(cq, f_all, reps), 4)
    # the traversal as visited nodes/s -- steps of the wave-uniform walk (each tests one
    # node record against a packet of 64 queries), from the counting instance of the kernel
    f_lbvh()
    stats = hip.Buffer(ctx, hostbuf=np.zeros(8, np.uint64))
    call.col_fill(s, n_buf.ptr, zero.ctypes.data, 4, 1)
    call.col_traverse_stats(s, pairs_buf.ptr, n_buf.ptr, capacity, bounds.ptr, n, cb, stats.ptr, 0)
    st = hip.read_buffer(cq, stats, np.uint64, 8)
    out["traverse_walk_steps"] = int(st[0])
    out["traverse_node_visits_per_s"] = float(st[0]) / (out["traverse"] * 1e-3)
    out["traverse_box_tests_per_s"] = 64.0 * float(st[0]) / (out["traverse"] * 1e-3)
    return out
