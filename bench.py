#!/usr/bin/env python3
"""Headline benchmark: million spheres/s end to end (bounds -> Morton -> radix sort -> LBVH ->
refit -> traversal) on BASELINE config 2 (1 M uniform-random spheres, r = 0.001, f32), plus the
radix-sort microbench of config 5 (64 Mi uint32 keys + uint32 ids) whose scatter pass carries
the HBM roofline claim.

    python bench.py [--gpus N] [--steps K] [--warmup W]

N = 1: one process.  N > 1: launched by torch.distributed.run, one rank per GPU (RCCL); every
rank holds 1 M spheres of an N x 1 M scene (weak scaling), see collision_amd/multi.py.
Rank 0 prints ONE JSON line.  A "step" is one get_collisions over device-resident inputs.
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
SCATTER_WARMUP, SCATTER_TIMED = 100, 50     # k_scatter launches: untimed, then the timed region


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
        res[name] = time_events(hip, cq, scatter, SCATTER_TIMED)       # steady state: the timed region

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
        "scatter_ms_cold": res["pass0_cold"],
        "algo_bytes_per_launch": algo_bytes,
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
    run()
    cq.finish()
    ms = time_events(hip, cq, run, reps)
    pairs = int(hip.read_buffer(cq, nb, np.uint32, 1)[0])
    return {"workload": "BASELINE config 3: 1M spheres in 8 Gaussian clusters (sigma=%g), r=%g" % (sigma, RADIUS),
            "ms_per_step": round(ms, 4), "m_spheres_per_s": round(N_SPHERES / ms / 1e3, 1), "pairs": pairs,
            "contacts_per_sphere": round(2.0 * pairs / N_SPHERES, 1), "m_pairs_per_s": round(pairs / ms / 1e3, 1)}


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


def pmc_traffic():
    """HBM bytes per k_scatter launch from the committed rocprofv3 PMC passes (separate --pmc
    FETCH_SIZE / WRITE_SIZE runs of tools/radix_only.py, summarised by tools/summarize_prof.py).
    gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reports half the bytes of a coalesced
    streaming read -- checked here on k_hist, which reads exactly n*4 bytes and reports n*2 -- so
    bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.  None if the summary is not present."""
    f = ROOT / "profiles" / "r01_radix64M_pmc.json"
    if not f.exists():
        return None
    summary = json.loads(f.read_text())
    # the 64 Mi-pair instance: k_scatter<u32 key, 4-byte value, 16 items/thread, 512 threads>
    names = [k for k in summary if k.startswith("k_scatter<unsigned int, 4")]
    names.sort(key=lambda k: ("16, 512" not in k, k))
    d = summary[names[0]] if names else None
    if not d or "FETCH_SIZE" not in d or "WRITE_SIZE" not in d:
        return None
    return (2.0 * d["FETCH_SIZE"]["median"] + d["WRITE_SIZE"]["median"]) * 1024.0


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline legs")
    ap.add_argument("--no-radix", action="store_true", help="skip the 64Mi-key radix microbench")
    ap.add_argument("--partition", default="morton", choices=["morton", "hash"],
                    help="multi-GPU: repartition spatially (default) or keep the id-hash partition")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d"
                     % (args.gpus, args.gpus))
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
        engine = DistributedCollider(ctx, dist, N_SPHERES, group_size=GROUP_SIZE, pair_capacity=PAIR_CAPACITY * 4,
                                     partition=args.partition)
        cq = engine.cq
        coords, radii, gids = make_rank_scene(N_SPHERES, rank, world, RADIUS)
        engine.set_local_spheres(coords, radii, gids)

        def step():
            engine.step()

        def sync():
            engine.synchronize()

        total_spheres = N_SPHERES * world
        parallelism = "%s-partition x%d, RCCL AABB all-gather + halo exchange" % (args.partition, world)

    def barrier():
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    sync()
    barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    barrier()
    sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3
    value = total_spheres / (elapsed / args.steps) / 1e6

    if world == 1:
        pair_count = int(hip.read_buffer(cq, n_buf, np.uint32, 1)[0])
    else:
        pair_count = engine.global_pair_count()
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
                        "launch_ms": round(rb["scatter_ms"], 4),
                        "launches_timed": SCATTER_TIMED, "warmup_launches": SCATTER_WARMUP,
                        "launch_ms_first_20_after_idle": round(rb["scatter_ms_cold"], 4),
                        "traffic": pmc_traffic(),
                        "traffic_source": "profiles/r01_radix64M_pmc.json (rocprofv3 --pmc, offline pass)"}
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
                extra["radix_sort"]["gkeys_per_s_other_distributions"] = config5_variants(hip, ctx, cq)
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
            "config": {"workload": "BASELINE config 2: %d uniform-random spheres per GPU, r=%g, f32, "
                                   "RandomState(4), pair capacity %d" % (N_SPHERES, RADIUS, PAIR_CAPACITY),
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
