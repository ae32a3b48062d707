#!/usr/bin/env python3
"""Headline benchmark: million spheres/s end to end (bounds -> Morton -> radix sort -> LBVH ->
refit -> traversal) on BASELINE config 2 (1 M uniform-random spheres, r = 0.001, f32), plus the
radix-sort microbench of config 5 (64 Mi uint32 keys + uint32 ids) whose scatter pass carries
the HBM roofline claim.

    python bench.py [--gpus N] [--steps K] [--warmup W]

N = 1: one process, BASELINE config 2 (1 M spheres).  N > 1: launched by torch.distributed.run, one rank
per GPU (RCCL); BASELINE config 4's workload -- every rank ARRIVES with the hash(id) mod N share of an
(N x 2 M)-sphere uniform scene (16 M spheres at N = 8), see collision_amd/multi.py; the same run at
1 M spheres per rank is reported beside it (`weak_1M_per_rank`).  Rank 0 prints ONE JSON line.  A "step"
is one get_collisions (N = 1) / one DistributedCollider.step (N > 1) over device-resident inputs.
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0       # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
N_SPHERES = 1000000         # BASELINE config 2
RADIUS = 0.001
GROUP_SIZE = 256
NGROUPS = 64
PAIR_CAPACITY = 1 << 17
SORT_KEYS = 1 << 26         # BASELINE config 5
SORT_WARMUP = 20            # whole sorts before the timed ones
SCATTER_WARMUP, SCATTER_TIMED = 100, 200    # k_scatter launches: untimed, then the timed region (one event per launch)
N_PER_RANK_MULTI = 2000000  # BASELINE config 4: 16 M spheres over 8 GPUs
COMPULSORY_BYTES_PER_SPHERE = 340   # bounds + Morton + 4 sort passes + leaves + Karras + refit, f32 / u32


def uniform_scene(n, seed=4):
    rng = np.random.RandomState(seed)
    coords = np.zeros((n, 4), np.float32)
    coords[:, :3] = rng.random_sample((n, 3)).astype(np.float32)
    return coords, np.full(n, RADIUS, np.float32)


def time_events(hip, cq, fn, reps):
    """Average device time of fn() in ms, HIP events on the stream fn launches on."""
    start, stop = hip.Event(), hip.Event()
    from collision_amd._lib import call
    call.col_event_record(start.handle, cq.stream)
    for _ in range(reps):
        fn()
    call.col_event_record(stop.handle, cq.stream)
    stop.wait()
    return start.elapsed_ms(stop) / reps


def time_events_each(hip, cq, fn, reps):
    """Per-launch device times (ms) of `reps` back-to-back fn() calls: one HIP event between consecutive
    launches on the launch stream.  Returns the sorted list."""
    from collision_amd._lib import call
    events = [hip.Event() for _ in range(reps + 1)]
    call.col_event_record(events[0].handle, cq.stream)
    for i in range(reps):
        fn()
        call.col_event_record(events[i + 1].handle, cq.stream)
    events[-1].wait()
    return sorted(events[i].elapsed_ms(events[i + 1]) for i in range(reps))


def radix_microbench(hip, ctx, cq, n=SORT_KEYS, reps=5):
    """Config 5: sort n (uint32 key, uint32 id) pairs; time the whole sort and each scatter pass."""
    from collision_amd._lib import call
    rng = np.random.RandomState(4)
    keys = rng.randint(0, 2 ** 30, size=n).astype(np.uint32)             # Morton-like 30-bit keys
    kin, vin = hip.Buffer(ctx, hostbuf=keys), hip.Buffer(ctx, hostbuf=np.arange(n, dtype=np.uint32))
    kout, vout = hip.Buffer(ctx, n * 4), hip.Buffer(ctx, n * 4)
    scratch = hip.Buffer(ctx, call.col_radix_scratch_bytes(n, 4, 4))

    def whole():
        call.col_radix_sort(cq.stream, kin.ptr, kout.ptr, vin.ptr, vout.ptr, n, 4, 4, scratch.ptr, 0)
    for _ in range(SORT_WARMUP):                               # the uploads idled the GPU: let clocks settle
        whole()
    cq.finish()
    sort_ms = time_events(hip, cq, whole, 2 * reps)
    # sanity: sorted + permutation on a sample
    out = hip.read_buffer(cq, kout, np.uint32, 1 << 20)
    assert (np.diff(out.astype(np.int64)) >= 0).all()

    # the scatter pass alone (pass 0 geometry; offsets prepared once, untimed)
    tile = call.col_radix_tile(n, 4, 4)
    nblocks = -(-n // tile)
    hist = hip.Buffer(ctx, 256 * nblocks * 4)
    scan_scratch = hip.Buffer(ctx, call.col_scan_scratch_bytes(256 * nblocks))
    res = {}
    for name, rpass in (("pass3", 3), ("pass0", 0)):
        call.col_radix_histogram(cq.stream, kin.ptr, n, 4, 4, rpass, hist.ptr)
        call.col_scan_u32(cq.stream, hist.ptr, 256 * nblocks, scan_scratch.ptr)

        def scatter():
            call.col_radix_scatter(cq.stream, kin.ptr, kout.ptr, vin.ptr, vout.ptr, n, 4, 4, rpass, hist.ptr)
        cq.finish()
        time.sleep(0.2)                                        # an idle gap, as between two host-driven phases
        res[name + "_cold"] = time_events(hip, cq, scatter, 20)        # launches 1-20 after the gap
        for _ in range(SCATTER_WARMUP):
            scatter()
        cq.finish()
        each = time_events_each(hip, cq, scatter, SCATTER_TIMED)       # steady state: the timed region
        res[name] = each[len(each) // 2]                               # median launch
        res[name + "_mean"] = sum(each) / len(each)
        res[name + "_p10_p90"] = (each[len(each) // 10], each[(9 * len(each)) // 10])

    # What a read-n-write-n pass can reach on THIS box, in the same run (col_debug_copy, csrc/radix.hip): the plain float4
    # copy (one vector per thread: the fastest form found, EXPERIMENTS.md R4.1) and the scatter pass's own tile shape --
    # 64 KB per 512-thread workgroup through LDS, coalesced stores -- with no ranking.  Same bytes as the pass (1 GiB moved),
    # interleaved with the production kernel.
    ceiling = None
    try:
        src, dst = hip.Buffer(ctx, n * 8), hip.Buffer(ctx, n * 8)
        call.col_memcpy_d2d(cq.stream, src.ptr, kin.ptr, n * 4)
        call.col_memcpy_d2d(cq.stream, src.ptr + n * 4, vin.ptr, n * 4)

        def series(fn):
            for _ in range(20):
                fn()
            cq.finish()
            each = time_events_each(hip, cq, fn, SCATTER_TIMED // 2)
            return each[len(each) // 2], (each[len(each) // 10], each[(9 * len(each)) // 10])
        ceiling = {}
        for name, shape in (("float4_copy", 0), ("tile_shape_copy", 1)):
            ms, p1090 = series(lambda: call.col_debug_copy(cq.stream, src.ptr, dst.ptr, n * 8, shape))
            ceiling[name] = {"launch_ms": ms, "p10_p90": p1090}
        ms, p1090 = series(scatter)
        ceiling["k_scatter_right_after"] = {"launch_ms": ms, "p10_p90": p1090}
        got = hip.read_buffer(cq, dst, np.uint32, 1 << 20)
        ceiling["copied_correctly"] = bool((got == keys[:1 << 20]).all())
        del src, dst
    except Exception as exc:                                    # (diagnostics leg: never lose the bench line over it)
        ceiling = {"error": repr(exc)}

    def histo():
        call.col_radix_histogram(cq.stream, kin.ptr, n, 4, 4, 0, hist.ptr)
    for _ in range(10):
        histo()
    hist_ms = time_events(hip, cq, histo, 4 * reps)
    scatter_ms = res["pass0"]
    algo_bytes = n * 16.0                                  # 2*key + 2*value bytes per pair
    return {
        "n_keys": n, "sort_ms": sort_ms, "gkeys_per_s": n / sort_ms / 1e6, "tile": tile,
        "scatter_ms": scatter_ms, "scatter_ms_top_digit": res["pass3"], "hist_ms": hist_ms,
        "scatter_gbs": algo_bytes / scatter_ms / 1e6,
        "scatter_ms_cold": res["pass0_cold"], "scatter_ms_mean": res["pass0_mean"], "scatter_ms_p10_p90": res["pass0_p10_p90"],
        "algo_bytes_per_launch": algo_bytes, "copy_ceiling": ceiling,
    }


def clustered_scene(n, sigma, seed=4):
    """BASELINE config 3: 8 Gaussian clusters, centres U(0.2,0.8)^3, n/8 points each."""
    rng = np.random.RandomState(seed)
    centres = rng.uniform(0.2, 0.8, size=(8, 3))
    pts = np.concatenate([rng.normal(c, sigma, size=(n // 8, 3)) for c in centres])
    coords = np.zeros((n, 4), np.float32)
    coords[:, :3] = pts
    return coords, np.full(n, RADIUS, np.float32)


def config3_leg(hip, ctx, cq, reps=5):
    """Traversal-divergence stress: 1 M clustered spheres, sigma calibrated for ~50 AABB contacts
    per sphere (~25 M pairs, 200 MB of output)."""
    from collision_amd.collision import Collider
    sigma, cap = 0.0152, 1 << 25
    coords, radii = clustered_scene(N_SPHERES, sigma)
    cb, rb = hip.Buffer(ctx, hostbuf=coords), hip.Buffer(ctx, hostbuf=radii)
    nb, pb = hip.Buffer(ctx, 4), hip.Buffer(ctx, cap * 8)
    col = Collider(ctx, N_SPHERES, NGROUPS, GROUP_SIZE)

    def run():
        col.get_collisions(cq, cb, rb, nb, pb, cap)
    for _ in range(3):        # (the Collider settles on its plans -- LSD sort, chunked pair allocation -- from the words the
        run()                 # first calls publish: steady state from the third call on)
        cq.finish()
    ms = time_events(hip, cq, run, reps)
    pairs = int(hip.read_buffer(cq, nb, np.uint32, 1)[0])
    return {"workload": "BASELINE config 3: 1M spheres in 8 Gaussian clusters (sigma=%g), r=%g" % (sigma, RADIUS),
            "ms_per_step": round(ms, 4), "m_spheres_per_s": round(N_SPHERES / ms / 1e3, 1), "pairs": pairs,
            "contacts_per_sphere": round(2.0 * pairs / N_SPHERES, 1), "m_pairs_per_s": round(pairs / ms / 1e3, 1)}


def single_gpu_leg(hip, ctx, cq, n, reps=20):
    """The single-GPU path at another size (config 4's 2 M spheres per rank), same generator and contact density."""
    from collision_amd.collision import Collider
    coords, radii = uniform_scene(n)
    radii[:] = RADIUS * (1e6 / n) ** (1.0 / 3.0)
    cb, rb = hip.Buffer(ctx, hostbuf=coords), hip.Buffer(ctx, hostbuf=radii)
    nb, pb = hip.Buffer(ctx, 4), hip.Buffer(ctx, PAIR_CAPACITY * 8)
    col = Collider(ctx, n, NGROUPS, GROUP_SIZE)

    def run():
        col.get_collisions(cq, cb, rb, nb, pb, PAIR_CAPACITY)
    for _ in range(5):
        run()
    cq.finish()
    ms = time_events(hip, cq, run, reps)
    # ~340 compulsory bytes per sphere for everything before the traversal (the traversal's own bytes are
    # data-dependent and NOT counted: this understates the path's traffic and so its fraction of the HBM peak)
    gbs = COMPULSORY_BYTES_PER_SPHERE * n / ms / 1e6
    return {"spheres": n, "ms_per_step": round(ms, 4), "m_spheres_per_s": round(n / ms / 1e3, 1),
            "pairs": int(hip.read_buffer(cq, nb, np.uint32, 1)[0]),
            "compulsory_gb_per_s": round(gbs, 1), "compulsory_frac_of_hbm_peak": round(gbs / HBM_PEAK_GBS, 4)}


def config2_f64_leg(hip, ctx, cq, reps=20):
    """BASELINE config 2's scene with float64 coordinates (the reference's tests run both dtypes, tests/test_collision.py:20-29)."""
    from collision_amd.collision import Collider
    coords, radii = uniform_scene(N_SPHERES)
    cb, rb = hip.Buffer(ctx, hostbuf=coords.astype(np.float64)), hip.Buffer(ctx, hostbuf=radii.astype(np.float64))
    nb, pb = hip.Buffer(ctx, 4), hip.Buffer(ctx, PAIR_CAPACITY * 8)
    col = Collider(ctx, N_SPHERES, NGROUPS, GROUP_SIZE, coord_dtype="float64")

    def run():
        col.get_collisions(cq, cb, rb, nb, pb, PAIR_CAPACITY)
    for _ in range(5):
        run()
    cq.finish()
    ms = time_events(hip, cq, run, reps)
    return {"workload": "BASELINE config 2 with float64 coordinates and radii", "ms_per_step": round(ms, 4),
            "m_spheres_per_s": round(N_SPHERES / ms / 1e3, 1), "pairs": int(hip.read_buffer(cq, nb, np.uint32, 1)[0])}


def config5_variants(hip, ctx, cq, n=SORT_KEYS, reps=3):
    """Config 5 key distributions (tests/benchmarks/test_radix.py:51-55 shapes): Gkeys/s each."""
    from collision_amd._lib import call
    rng = np.random.RandomState(4)
    out = {}
    kout, vout = hip.Buffer(ctx, n * 4), hip.Buffer(ctx, n * 4)
    vin = hip.Buffer(ctx, hostbuf=np.arange(n, dtype=np.uint32))
    scratch = hip.Buffer(ctx, call.col_radix_scratch_bytes(n, 4, 4))
    for name, keys in (("uniform32", rng.randint(0, 2 ** 32, size=n, dtype=np.uint64).astype(np.uint32)),
                       ("arange", np.arange(n, dtype=np.uint32))):
        kin = hip.Buffer(ctx, hostbuf=keys)
        for with_values in (True, False):
            def run():
                call.col_radix_sort(cq.stream, kin.ptr, kout.ptr, vin.ptr if with_values else None,
                                    vout.ptr if with_values else None, n, 4, 4 if with_values else 0, scratch.ptr, 0)
            for _ in range(SORT_WARMUP):
                run()
            cq.finish()
            ms = time_events(hip, cq, run, 2 * reps)
            out["%s_%s" % (name, "pairs" if with_values else "keys")] = round(n / ms / 1e6, 2)
        del kin
    return out


def pmc_traffic(timeout_s=150):
    """HBM bytes per k_scatter launch, measured in THIS run: two child processes run the same radix
    microbench (tools/radix_only.py) under `rocprofv3 --pmc`, FETCH_SIZE and WRITE_SIZE in separate passes
    (MI355X_MICROARCH.md, rocprofv3 PMC slots: they do not fit one pass), after the timed region.
    gfx950 correction (same guide, HBM): FETCH_SIZE reports half the bytes of a coalesced 16-byte-per-lane
    streaming read -- calibrated here on k_hist of the same pass, which reads exactly n * 4 bytes -- so
    bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.  Returns (bytes or None, detail dict)."""
    import csv
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None, {"error": "rocprofv3 not on PATH"}
    detail = {}
    med = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        out = tempfile.mkdtemp(prefix="col_pmc_")
        try:
            env = dict(os.environ, TMPDIR="/tmp")
            proc = subprocess.run(["rocprofv3", "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", out,
                                   "-o", "pmc", "--", sys.executable, str(ROOT / "tools" / "radix_only.py"), "1"],
                                  cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout_s)
            if proc.returncode != 0:
                detail[counter] = "rocprofv3 rc %d: %s" % (proc.returncode, proc.stderr[-300:])
                return None, detail
            per = {"k_scatter": [], "k_hist": []}
            for f in Path(out).rglob("*counter_collection.csv"):
                for row in csv.DictReader(open(f)):
                    if row.get("Counter_Name") != counter:
                        continue
                    for kern in per:
                        if kern + "<unsigned int" in row["Kernel_Name"]:
                            per[kern].append(float(row["Counter_Value"]))
            for kern, v in per.items():
                if v:
                    v.sort()
                    med[(kern, counter)] = v[len(v) // 2]
                    detail["%s_%s_KB_median" % (kern, counter)] = v[len(v) // 2]
                    detail["%s_launches" % kern] = len(v)
        except Exception as exc:                       # a profiler problem must not lose the bench line
            detail[counter] = repr(exc)
            return None, detail
        finally:
            shutil.rmtree(out, ignore_errors=True)
    if ("k_scatter", "FETCH_SIZE") not in med or ("k_scatter", "WRITE_SIZE") not in med:
        return None, detail
    if ("k_hist", "FETCH_SIZE") in med:               # calibration: k_hist reads n * 4 bytes
        detail["fetch_correction_measured_on_k_hist"] = round(SORT_KEYS * 4 / 1024.0 / med[("k_hist", "FETCH_SIZE")], 3)
    return (2.0 * med[("k_scatter", "FETCH_SIZE")] + med[("k_scatter", "WRITE_SIZE")]) * 1024.0, detail


def reference_benchmark_shapes(hip, ctx, cq):
    """The reference's remaining pytest-benchmark workloads (tests/benchmarks/test_scan.py:29-53,
    test_bounds.py:18-41, test_offset.py:24-39, test_radix.py:51-139, test_collide.py:24-54), timed the
    way it does (device-resident inputs, enqueue -> done), ms per call."""
    from collision_amd._lib import call
    from collision_amd.bounds import Bounds
    from collision_amd.collision import Collider
    from collision_amd.offset import OffsetFinder
    from collision_amd.radix import RadixSorter
    from collision_amd.scan import PrefixScanner
    rng = np.random.RandomState(4)
    out = {}

    def timed(fn, reps=50):
        for _ in range(10):                                    # warmup_rounds=10 as the reference
            fn()
        cq.finish()
        return round(time_events(hip, cq, fn, reps), 5)

    for size in (307200, 1536000, 3072000):                    # test_scan.py
        vals = rng.randint(0, 128, size=size).astype(np.uint32)
        buf = hip.Buffer(ctx, hostbuf=vals)
        scanner = PrefixScanner(ctx, size, 128)
        out["scan_u32_%d" % size] = timed(lambda: scanner.prefix_sum(cq, buf))
    for size in (1536000, 3072000):                            # test_bounds.py
        rows = rng.uniform(0, 1, size=(size, 4)).astype(np.float32)
        buf, outb = hip.Buffer(ctx, hostbuf=rows), hip.Buffer(ctx, 64)
        bounds = Bounds(ctx, 64, 128, coord_dtype=np.dtype((np.float32, 4)))
        out["bounds_f32x4_%d" % size] = timed(lambda: bounds.reduce(cq, size, buf, outb))
    for maxval in (2000, 2000000):                             # test_offset.py
        size = 1 << 21
        vals = np.sort(rng.randint(0, maxval, size=size).astype(np.uint32))
        vb, ob = hip.Buffer(ctx, hostbuf=vals), hip.Buffer(ctx, (maxval + 1) * 4)
        finder = OffsetFinder(ctx)
        out["offsets_2e21_max%d" % maxval] = timed(lambda: finder.find_offsets(cq, vb, size, ob, maxval + 1))
    size = 307200                                              # test_radix.py
    for name, keys in (("randint1000", rng.randint(0, 1000, size=size)), ("randint307200", rng.randint(0, size, size=size)),
                       ("arange", np.arange(size))):
        for kd in ("uint32", "uint64"):
            kb = hip.Buffer(ctx, hostbuf=keys.astype(kd))
            ko = hip.Buffer(ctx, size * np.dtype(kd).itemsize)
            sorter = RadixSorter(ctx, size, 128, key_dtype=np.dtype(kd))
            out["radix_keys_%s_%s" % (kd, name)] = timed(lambda: sorter.sort(cq, kb, ko))
    keys = rng.randint(0, size, size=size).astype(np.uint32)
    for vname, vdt in (("u32", np.dtype("uint32")), ("f64", np.dtype("float64")), ("f32x3", np.dtype(("float32", 3))),
                       ("f32x4", np.dtype(("float32", 4)))):
        sorter = RadixSorter(ctx, size, 128, key_dtype=np.dtype("uint32"), value_dtype=vdt)
        kb, ko = hip.Buffer(ctx, hostbuf=keys), hip.Buffer(ctx, size * 4)
        vb, vo = hip.Buffer(ctx, size * sorter.value_bytes), hip.Buffer(ctx, size * sorter.value_bytes)
        out["radix_pairs_u32_%s" % vname] = timed(lambda: sorter.sort(cq, kb, ko, vb, vo))
    for size in (307200, 307201):                              # test_collide.py: count-only mode
        coords = np.zeros((size, 4), np.float32)
        coords[:, :3] = rng.uniform(-1, 1, size=(size, 3))
        radii = rng.uniform(0.006, 0.06, size=size).astype(np.float32)
        cb, rb, nb = hip.Buffer(ctx, hostbuf=coords), hip.Buffer(ctx, hostbuf=radii), hip.Buffer(ctx, 4)
        col = Collider(ctx, size, 8, 128)
        out["collide_count_only_%d" % size] = timed(lambda: col.get_collisions(cq, cb, rb, nb, None, 0), reps=10)
    return out


def cpu_baseline(coords, radii, budget_s=12.0, max_runs=20):
    """The CPU oracle (C port of the path, 1 thread) on the same 1 M scene, repeated for ~budget_s."""
    import oracle
    oracle.build()
    oracle.collide(coords[:1000], radii[:1000], capacity=0, want=False)      # load/warm the library
    runs, t0 = 0, time.perf_counter()
    count = 0
    while runs < max_runs and (time.perf_counter() - t0 < budget_s or runs == 0):
        count = oracle.collide(coords, radii, padded=len(coords), capacity=0, want=False)["count"]
        runs += 1
    dt = (time.perf_counter() - t0) / runs
    return {"value": len(coords) / dt / 1e6, "unit": "M spheres/s", "cores": 1, "kind": "port",
            "sample": "%d runs of the full config-2 scene (1M spheres) through oracle/collision_oracle.c "
                      "(single-thread C restatement; the reference has no CPU pipeline, only an O(n^2) NumPy "
                      "test oracle)" % runs,
            "host_cores": os.cpu_count(), "pairs": count}


def cpu_bruteforce_10k():
    """BASELINE config 1: the reference's own CPU-runnable case (NumPy brute force, restated)."""
    import oracle
    c, r = uniform_scene(10000)
    t0 = time.perf_counter()
    pairs = oracle.find_collisions(c[:, :3], r)
    dt = time.perf_counter() - t0
    return {"n": 10000, "seconds": dt, "m_spheres_per_s": 10000 / dt / 1e6, "pairs": len(pairs), "cores": 1}


def cpu_bruteforce_all_cores(n=100000):
    """BASELINE.md section 4's "fair many-core figure": the same O(n^2) test tiled over every host core
    (OpenMP rows of the upper triangle, oracle/collision_oracle.c), with the n^2 extrapolation to 1M."""
    import oracle
    c, r = uniform_scene(n)
    t0 = time.perf_counter()
    count, threads = oracle.brute_force_count_all_cores(c, r)
    dt = time.perf_counter() - t0
    return {"n": n, "seconds": dt, "cores": threads, "pairs": count,
            "pair_tests_per_s": n * (n - 1) / 2 / dt,
            "extrapolated_seconds_at_1M": dt * (N_SPHERES / n) ** 2}


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: run N ranks of this script under torch.distributed.run as a
    child process (never an exec: nothing here may replace a process that could hold the GPU) and return its
    exit code.  The rendezvous is on 127.0.0.1 at a port that is free right now."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline legs")
    ap.add_argument("--no-radix", action="store_true", help="skip the 64Mi-key radix microbench")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 PMC child passes (roofline.traffic)")
    ap.add_argument("--partition", default="morton", choices=["morton", "hash"],
                    help="multi-GPU: repartition spatially (default) or keep the id-hash partition")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if "WORLD_SIZE" not in os.environ and args.gpus > 1:
            # typed as a plain command: this process (which has not touched torch or HIP) starts the ranks itself,
            # one per GPU, as a child torch.distributed.run; rank 0's JSON line goes straight to our stdout
            sys.exit(launch_ranks(args.gpus))
        args.gpus = world

    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        backend = os.environ.get("COLLISION_BENCH_BACKEND", "nccl")      # "gloo": rehearsal on a shared GPU
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            local_rank = 0
            torch.cuda.set_device(0)
            dist.init_process_group(backend)

    from collision_amd import hip
    from collision_amd.collision import Collider

    ctx = hip.Context(local_rank)
    extra = {}

    if world == 1:
        cq = hip.CommandQueue(ctx)
        coords, radii = uniform_scene(N_SPHERES)
        coords_buf, radii_buf = hip.Buffer(ctx, hostbuf=coords), hip.Buffer(ctx, hostbuf=radii)
        n_buf, pairs_buf = hip.Buffer(ctx, 4), hip.Buffer(ctx, PAIR_CAPACITY * 8)
        collider = Collider(ctx, N_SPHERES, NGROUPS, GROUP_SIZE)

        def step():
            collider.get_collisions(cq, coords_buf, radii_buf, n_buf, pairs_buf, PAIR_CAPACITY)

        def sync():
            cq.finish()

        total_spheres = N_SPHERES
        parallelism = "single"
    else:
        from collision_amd.multi import DistributedCollider, make_rank_scene

        def make_engine(n_per_rank):
            eng = DistributedCollider(ctx, dist, n_per_rank, group_size=GROUP_SIZE, pair_capacity=PAIR_CAPACITY * 8,
                                      partition=args.partition)
            # r shrinks with the density so that contacts per sphere stay those of config 2
            c, r, g = make_rank_scene(n_per_rank, rank, world, RADIUS * (1e6 / n_per_rank) ** (1.0 / 3.0))
            eng.set_local_spheres(c, r, g)
            return eng

        engine = make_engine(N_PER_RANK_MULTI)
        cq = engine.cq

        def step():
            engine.step()

        def sync():
            engine.synchronize()

        total_spheres = N_PER_RANK_MULTI * world
        parallelism = "hash(id) mod %d arrival, %s repartition, RCCL AABB all-gathers + halo exchange" % (world, args.partition)

    def barrier():
        if dist is not None:
            dist.barrier()

    def timed_region(step_fn, sync_fn):
        """W untimed steps, then exactly K steps between barrier + sync on both sides; max over ranks."""
        for _ in range(args.warmup):
            step_fn()
        sync_fn()
        barrier()
        sync_fn()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step_fn()
        sync_fn()
        barrier()
        sync_fn()
        dt = time.perf_counter() - t0
        if dist is not None:
            import torch
            t = torch.tensor([dt], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    elapsed = timed_region(step, sync)
    ms_per_step = elapsed / args.steps * 1e3
    value = total_spheres / (elapsed / args.steps) / 1e6

    if world == 1:
        pair_count = int(hip.read_buffer(cq, n_buf, np.uint32, 1)[0])
    else:
        pair_count = engine.global_pair_count()
        extra["dist_backend"] = dist.get_backend()
        extra["world_size_seen"] = dist.get_world_size()
        extra["per_rank"] = {"owned_spheres_rank0": engine.stats.get("owned"), "ghost_queries_rank0": engine.stats.get("ghosts"),
                             "partition_slot_records": engine.stats.get("partition_slot"),
                             "halo_slot_records": engine.stats.get("halo_slot"), "repeated_steps": engine.repeats}
        # the same spheres with a coherent arrival: each step's input is what the rank owned after the previous one
        # (adopt_owned: a simulation advancing positions in place).  Nothing has moved, so after the first step every
        # sphere is kept by its rank and the repartition slots travel nearly empty.  NOT the headline workload.
        def coherent_step():
            engine.adopt_owned()
            engine.step()
        dtc = timed_region(coherent_step, engine.synchronize)
        extra["coherent_arrival"] = {"workload": "%d x 2M spheres, input of a step = owned spheres of the previous one" % world,
                                     "ms_per_step": round(dtc / args.steps * 1e3, 4),
                                     "m_spheres_per_s": round(N_PER_RANK_MULTI * world / (dtc / args.steps) / 1e6, 2),
                                     "partition_slot_records": engine.stats.get("partition_slot"),
                                     "pairs_found": engine.global_pair_count()}
        # the same protocol at 1 M spheres per rank (round 1's line), reported beside the config-4 workload
        del engine
        engine1 = make_engine(N_SPHERES)
        dt1 = timed_region(engine1.step, engine1.synchronize)
        extra["weak_1M_per_rank"] = {"workload": "%d x 1M spheres, same protocol" % world, "ms_per_step": round(dt1 / args.steps * 1e3, 4),
                                     "m_spheres_per_s": round(N_SPHERES * world / (dt1 / args.steps) / 1e6, 2),
                                     "pairs_found": engine1.global_pair_count()}
        cq = engine1.cq
    extra["pairs_found"] = pair_count

    result = None
    if rank == 0:
        roofline = None
        rb = None
        if not args.no_radix:
            # the dominant kernel is the same on every rank: rank 0 measures it live at any N
            try:
                rb = radix_microbench(hip, ctx, cq)
            except Exception as exc:            # never lose the bench line of an N>1 run over the micro leg
                if world == 1:
                    raise
                extra["roofline_error"] = repr(exc)
        if rb is not None:
            extra["radix_sort"] = {"n_keys": rb["n_keys"], "key": "u32 (30-bit, Morton-like)", "value": "u32",
                                   "gkeys_per_s": round(rb["gkeys_per_s"], 3), "sort_ms": round(rb["sort_ms"], 4),
                                   "hist_ms": round(rb["hist_ms"], 4),
                                   "scatter_ms_top_digit": round(rb["scatter_ms_top_digit"], 4),
                                   # passes x (scatter bytes + histogram read of the keys)
                                   "algo_bytes_whole_sort": 4 * (16 + 4) * rb["n_keys"],
                                   "whole_sort_gbs": round(4 * (16 + 4) * rb["n_keys"] / rb["sort_ms"] / 1e6, 1)}
            roofline = {"bound": "hbm", "kernel": "radix k_scatter<u32 key, 4-byte value, 16 items, 512 threads> "
                                                   "(64Mi pairs, 8-bit digit, pass 0, tile %d)" % rb["tile"],
                        "achieved": round(rb["scatter_gbs"], 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(rb["scatter_gbs"] / HBM_PEAK_GBS, 4),
                        "algo_bytes_per_launch": rb["algo_bytes_per_launch"],
                        "launch_ms": round(rb["scatter_ms"], 4), "launch_ms_is": "median of the timed launches (one HIP event per launch)",
                        "launch_ms_mean": round(rb["scatter_ms_mean"], 4),
                        "launch_ms_p10_p90": [round(v, 4) for v in rb["scatter_ms_p10_p90"]],
                        "launches_timed": SCATTER_TIMED, "warmup_launches": SCATTER_WARMUP,
                        "launch_ms_first_20_after_idle": round(rb["scatter_ms_cold"], 4),
                        "traffic": None}
            cc = rb.get("copy_ceiling")
            if cc and "float4_copy" in cc:
                def frac(ms):
                    return round(rb["algo_bytes_per_launch"] / ms / 1e6 / HBM_PEAK_GBS, 4)
                roofline["copy_ceiling"] = {
                    "what": "a pass that reads n bytes and writes n bytes, same 1 GiB, same run (col_debug_copy): the plain float4 "
                            "copy (one vector per thread) and the scatter pass's tile shape (64 KB per 512-thread workgroup "
                            "through LDS, coalesced stores) without ranking; the pass pays for its ranking and its scattered runs on top",
                    "float4_copy_ms": round(cc["float4_copy"]["launch_ms"], 4), "float4_copy_frac_of_hbm_peak": frac(cc["float4_copy"]["launch_ms"]),
                    "tile_shape_copy_ms": round(cc["tile_shape_copy"]["launch_ms"], 4),
                    "tile_shape_copy_frac_of_hbm_peak": frac(cc["tile_shape_copy"]["launch_ms"]),
                    "k_scatter_ms_right_after": round(cc["k_scatter_right_after"]["launch_ms"], 4),
                    "k_scatter_frac_of_float4_copy": round(cc["float4_copy"]["launch_ms"] / cc["k_scatter_right_after"]["launch_ms"], 4),
                    "copied_correctly": cc["copied_correctly"]}
            elif cc:
                roofline["copy_ceiling"] = cc
            if world == 1 and not args.no_pmc:
                roofline["traffic"], roofline["traffic_detail"] = pmc_traffic()
                roofline["traffic_source"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child passes of tools/radix_only.py in this "
                                              "run; bytes = (2 * FETCH + WRITE) * 1024")
        if world == 1:
            # per-stage device times of the 1M path (HIP events on the launch stream)
            from collision_amd.stages import stage_times
            extra["stage_ms"] = stage_times(hip, ctx, cq, collider, coords_buf, radii_buf, n_buf, pairs_buf,
                                            PAIR_CAPACITY)
            # count-only mode (collision.cl:203-207: capacity 0, no pair buffer)
            def count_only():
                collider.get_collisions(cq, coords_buf, radii_buf, n_buf, None, 0)
            count_only()
            cq.finish()
            extra["count_only_ms"] = round(time_events(hip, cq, count_only, 20), 4)
            extra["count_only_pairs"] = int(hip.read_buffer(cq, n_buf, np.uint32, 1)[0])
            if not args.no_radix:
                extra["config3_clustered"] = config3_leg(hip, ctx, cq)
                extra["config2_f64"] = config2_f64_leg(hip, ctx, cq)
                extra["radix_sort"]["gkeys_per_s_other_distributions"] = config5_variants(hip, ctx, cq)
                extra["config4_per_rank_size_on_one_gpu"] = single_gpu_leg(hip, ctx, cq, N_PER_RANK_MULTI)
                # config 4's whole scene on ONE GPU: the regime where the path runs out of HBM, not out of the caches
                extra["config4_whole_scene_on_one_gpu"] = single_gpu_leg(hip, ctx, cq, 8 * N_PER_RANK_MULTI, reps=5)
                extra["reference_benchmark_shapes_ms"] = reference_benchmark_shapes(hip, ctx, cq)
        cpu = None
        if not args.no_cpu and world == 1:
            cpu = cpu_baseline(coords, radii)
            extra["cpu_bruteforce_config1"] = cpu_bruteforce_10k()
            extra["cpu_bruteforce_all_cores"] = cpu_bruteforce_all_cores()
        result = {
            "metric": "M spheres/sec end-to-end (bounds->Morton->radix sort->LBVH->refit->traversal)",
            "value": round(value, 2), "unit": "M spheres/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32 coords / u32 keys+ids", "data": "synthetic",
            "config": {"workload": ("BASELINE config 2: %d uniform-random spheres, r=%g, f32, RandomState(4), pair capacity %d"
                                    % (N_SPHERES, RADIUS, PAIR_CAPACITY)) if world == 1 else
                                   ("BASELINE config 4: %d uniform-random spheres (RandomState(4)) arriving hash(id) mod %d "
                                    "partitioned, %d per GPU, r=%.3g (contacts per sphere of config 2), f32"
                                    % (total_spheres, world, N_PER_RANK_MULTI, RADIUS * (1e6 / total_spheres) ** (1.0 / 3.0))),
                       "spheres_total": total_spheres, "group_size": GROUP_SIZE, "parallelism": parallelism},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        result.update(extra)
        print(json.dumps(result))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return result


if __name__ == "__main__":
    main()
