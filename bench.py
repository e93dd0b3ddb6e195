#!/usr/bin/env python3
"""bench.py -- fp64 supernodal Cholesky factorisation throughput on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--case lapl_3375x3375]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one numeric factorisation (the level loop of mmat.rg:1227-1355) of the workload, on an
arena that already holds P A P^T in HBM when the timed region starts (a fresh pre-filled arena per
step, so nothing is cached from the previous step and no re-fill sits inside the timed region).
value = F_ref * K / t, F_ref = the reference's own BLAS-call flop count.

N > 1: the SAME factorisation is sharded by subtrees of the separator tree (strong scaling): rank g
factors the subtrees below tree level log2(N), the extend-add contributions to the shared top of
the tree are summed with one RCCL all-reduce over the contiguous tail of the arena, then every
rank factors the top levels -- one C-ABI call per rank and step (cholamd_factor_sharded).
`--gpus N` without a launcher (WORLD_SIZE unset) starts the N rank processes itself, as children,
before anything touches the GPU; a world size that differs from --gpus is an error, never a silent
single-GPU run.

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CASES = {
    "lapl_9x9": ("lapl_3_2.mtx", "lapl_3_2_ord_2.txt", "lapl_3_2_clust_2.txt"),
    "lapl_25x25": ("lapl_5_2.mtx", "lapl_5_2_ord_3.txt", "lapl_5_2_clust_3.txt"),
    "lapl_400x400": ("lapl_20_2.mtx", "lapl_20_2_ord_5.txt", "lapl_20_2_clust_5.txt"),
    "lapl_3375x3375": ("lapl_15_3.mtx", "lapl_15_3_ord_5.txt", "lapl_15_3_clust_5.txt"),
}
SUSTAINED_FP64_MFMA_TFLOPS = 53.5  # highest executed fp64 MFMA rate measured on this pool: k_update_mt on one 8192^2 SYRK at K = 432 (scripts/mt_bench.hip)
SUSTAINED_NOTE = ("highest executed v_mfma_f64_16x16x4_f64 rate measured on this pool: the macro-tile update kernel on one 8192^2 SYRK target at K = 432 "
                  "(scripts/mt_bench.hip); register-operand loops on all 256 CUs reach 47.7-50.5 (scripts/mfma_peak.hip, scripts/mfma32_probe.hip; "
                  "profiles/r3/mfma_peak/): 68 % of `peak`; `frac` stays against `peak`")
PROFILE_ROUND = "r4" if os.path.exists(os.path.join(ROOT, "profiles", "r4", "summary.json")) else "r3"
PROFILE_SUMMARY = os.path.join(ROOT, "profiles", PROFILE_ROUND, "summary.json")  # scripts/profile_summary.py over the rocprofv3 passes of this command
# the large-front regime reported beside the metric in the default single-GPU run: key -> (N, levels, tile, mixed)
LARGE_FRONT = (("gen_60_8", (60, 8, 64, False)), ("gen_100_10", (100, 10, 64, False)), ("gen_100_10_mixed", (100, 10, 64, True)))


def stored_rows(plan, r, c, rows):
    """Rows of block (r, c) that its panel stores (row compaction: the kept 16-row tiles)."""
    tm = plan.block_tile_map(r, c)
    return int(sum(min(16, rows - 16 * t) for t in range(len(tm)) if tm[t] >= 0))


def profile_numbers(kernel, case, mixed, options=()):
    """Counter-derived numbers of `kernel` for this workload from the committed rocprofv3 passes (counters cannot be
    read live from inside the process): HBM bytes per launch, average launch duration, MFMA busy fraction."""
    try:
        with open(PROFILE_SUMMARY) as f:
            runs = json.load(f)["runs"]
        key = case + (":mixed" if mixed else "") + ("".join(":" + o for o in sorted(options)) if options else "")
        k = runs[key]["kernels"][kernel]
        return {"hbm_bytes_per_launch": k.get("hbm_bytes_per_launch_corrected"), "avg_launch_us": k.get("avg_launch_us"),
                "mfma_busy_frac": k.get("mfma_busy_frac"), "source": f"profiles/{PROFILE_ROUND}/summary.json: " + runs[key].get("source", "")}
    except Exception:
        return {}


def profile_solve_numbers(case, mixed, n_solves_timed, alg_bytes_factor):
    """The solve's kernels in the committed rocprofv3 passes of this case (a mixed-precision pass runs the refinement after its factorisations): HBM
    bytes and kernel time per solve = sums over the k_solve_* kernels / the solves of the pass (one k_permute_in launch each); the achieved rate
    against HBM peak, and the same time against the ALGORITHMIC bytes of a solve (the factor's non-zeros read once per sweep: two sweeps)."""
    try:
        from cholesky_amd import benchline
        with open(PROFILE_SUMMARY) as f:
            runs = json.load(f)["runs"]
        ks = runs[case + (":mixed" if mixed else "")]["kernels"]
        sel = [v for k, v in ks.items() if k.startswith("k_solve_")]
        n_t, n_c = ks["k_permute_in"]["launches"], ks["k_permute_in"].get("pmc_launches")
        us = sum(v["avg_launch_us"] * v["launches"] for v in sel) / n_t
        byt = sum(v["hbm_bytes_per_launch_corrected"] * v["pmc_launches"] for v in sel if v.get("hbm_bytes_per_launch_corrected") is not None) / n_c
        alg = 2.0 * alg_bytes_factor
        return {"solve_kernels_ms_rocprof": round(us * 1e-3, 3), "hbm_bytes_per_solve_rocprof": int(byt), "hbm_GBs": round(byt / us * 1e-3, 1),
                "hbm_frac_of_peak": round(byt / us * 1e-3 / benchline.PEAK_HBM_GBS, 4), "alg_bytes_per_solve": int(alg), "alg_GBs": round(alg / us * 1e-3, 1),
                "note": "stored panels are dense blocks: the sweeps read every stored entry (structural zeros inside the kept tiles included), "
                        "alg_bytes counts the factor's non-zeros at the factor's element size, twice",
                "solves_in_timed_refinement": n_solves_timed, "source": f"profiles/{PROFILE_ROUND}/summary.json: k_solve_* of this case"}
    except Exception:
        return {}


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(files, flops, budget_s=12.0):
    """The oracle (C restatement of the reference CPU path, OpenBLAS 1 thread per call when one is on the box, exactly as the
    reference links it, mmat.rg:1057) timed on this host on a bounded sample: single thread (`value`), and the level loop with
    the reference's task parallelism -- 3 workers (test_matrices.py:27, -ll:cpu 3) and one worker per host core."""
    from oracle import oracle as orc
    backend = "openblas" if orc.use_openblas() else "own-c-kernels"
    ncores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:  # the share of the host this job may use (cgroup quota), else the GPU box's documented share of 16 cores per GPU
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        share = max(1, int(int(quota) / int(period))) if quota != "max" else 16
    except (OSError, ValueError):
        share = 16
    ncores = max(1, min(ncores, share))
    try:
        O = orc.Oracle(*files)
        O.factor()  # warm-up
        t_acc, n = 0.0, 0
        t_wall = time.perf_counter()
        while time.perf_counter() - t_wall < budget_s and n < 2000:
            t_acc += O.factor()
            n += 1
        name = orc.backend_name()
        variants = {}
        for label, workers in (("workers_3", 3), ("workers_all_cores", ncores)):
            O.factor_parallel(workers)  # thread team start-up
            ta, k = 0.0, 0
            t_wall = time.perf_counter()
            while time.perf_counter() - t_wall < budget_s / 3 and k < 2000:
                ta += O.factor_parallel(workers)
                k += 1
            variants[label] = {"value": round(flops * k / ta * 1e-9, 3), "unit": "GF/s", "cores": workers, "ms": round(ta / k * 1e3, 3), "factorisations": k}
    finally:
        orc.use_own_kernels()
    return {
        "value": round(flops * n / t_acc * 1e-9, 3), "unit": "GF/s", "cores": 1, "kind": "port",
        "sample": f"{n} full numeric factorisations of the same workload (level loop only, {t_acc / n * 1e3:.2f} ms each), "
                  f"oracle/chol_oracle.c with {name}",
        "backend": backend, "cpu": cpu_model(), "host_cores": ncores,
        "task_parallel": variants,
        "task_parallel_note": "the same level loop with OpenMP task parallelism inside a level as Legion extracts it (POTRF / TRSM tasks of a level independent, "
                              "update tasks serialised per target block), BLAS one thread per call; bit-identical factors",
    }


def large_front(ca, torch, stream, specs=LARGE_FRONT, steps=2, warmup=1):
    """Measurement records (benchline.large_front_entry) of the large-front regime on this GPU: whole numeric factorisations of generated
    N^3 Laplacians, each timed with device synchronisation on both sides on a re-filled arena, one more under the library's per-launch HIP
    events for the dominant kernel's share; fp64: one solve, mixed: the fp64 iterative refinement of one right-hand side."""
    from cholesky_amd import benchline
    out, plans = {}, {}
    for key, (gn, glv, gtile, mixed) in specs:
        if (gn, glv, gtile) not in plans:
            plans.clear()  # one large plan alive at a time
            plans[(gn, glv, gtile)] = ca.Problem(gn, gn, gn, glv, gtile).plan()
        plan = plans[(gn, glv, gtile)]
        dev = ca.Device(plan, torch.cuda.current_device())
        a = dev.new_arena_f32() if mixed else dev.new_arena()
        fill, factor = (dev.fill_f32, dev.factor_f32) if mixed else (dev.fill, dev.factor)
        step_s = []
        for i in range(warmup + steps):
            fill(a, stream)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            factor(a, stream)
            torch.cuda.synchronize()
            if i >= warmup:
                step_s.append(time.perf_counter() - t0)
        info = dev.info()
        if info[0] != 0:
            sys.exit(f"bench.py: large_front {key}: factorisation failed: info {info}")
        bvec = torch.from_numpy(ca.Problem(gn, gn, gn, glv, gtile).rhs()).cuda()
        xvec = torch.empty_like(bvec)
        refine = solve = None
        if mixed:
            dev.solve_refine(a, bvec, xvec, 30, 1e-11, stream)  # warm-up (work lists, CSR upload)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            it, rel = dev.solve_refine(a, bvec, xvec, 30, 1e-11, stream)
            torch.cuda.synchronize()
            refine = {"corrections": it, "relres": rel, "ms": round((time.perf_counter() - t0) * 1e3, 3), "tol": 1e-11}
            refine["traffic"] = profile_solve_numbers(f"gen:{gn}:{glv}" + (f":{gtile}" if gtile != 64 else ""), True, it + 1, (plan.alg_bytes // 8) * 4)
        else:
            dev.solve(a, bvec, xvec, stream)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            dev.solve(a, bvec, xvec, stream)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) * 1e3
            r = torch.empty_like(bvec)
            dev.residual(bvec, xvec, r, stream)
            torch.cuda.synchronize()
            solve = {"ms": round(ms, 3), "relres": float(r.norm() / bvec.norm())}
            solve["traffic"] = profile_solve_numbers(f"gen:{gn}:{glv}" + (f":{gtile}" if gtile != 64 else ""), False, 1, plan.alg_bytes)
        fill(a, stream)
        dev.sync(stream)
        dev.set_timing(1)
        factor(a, stream)
        dev.sync(stream)
        timing = dev.get_timing_ex()
        dev.set_timing(0)
        calls, flops = plan.counts()
        case = f"gen:{gn}:{glv}" + (f":{gtile}" if gtile != 64 else "")
        kinds = {"potrf": float(flops[0]), "trsm": float(flops[1]), "update": float(flops[2] + flops[3])}
        fused = timing["trsm"][1] == 0
        dom = max(("potrf", "trsm", "update"), key=lambda k: timing[k][0])
        out[key] = {"case": case, "mixed": mixed, "step_s": step_s, "info": list(info), "event_pair_ms": dev.event_overhead_ms(stream),
                    "plan": {"n": plan.n, "flops": plan.flops, "alg_bytes": plan.alg_bytes, "arena_bytes": plan.arena_doubles * (4 if mixed else 8), "flops_by_kind": kinds},
                    "timing": {k: list(v) for k, v in timing.items()}, "refinement": refine, "solve": solve,
                    "profile": profile_numbers(benchline.kernel_names(mixed, True, fused)[dom].split(" + ")[0], case, mixed)}
        del a, dev, bvec, xvec
        torch.cuda.empty_cache()
    return out


def launch_ranks(n):
    """--gpus n without a launcher: start n fresh rank processes (torch.distributed.run, one per GPU) as children of
    this process, which has not touched the GPU, and return their exit code."""
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--case", default=None,
                    help="a reference fixture (default: the metric's configuration) or gen:N:levels[:tile] = a generated N^3 7-point "
                         "Laplacian with geometric nested dissection (e.g. gen:100:10, BASELINE config 5's matrix; no cpu_baseline)")
    ap.add_argument("--precision", default="fp64", choices=["fp64", "mixed"],
                    help="mixed = fp32 factor (the timed step) + fp64 iterative refinement of one solve (reported in config), BASELINE config 5")
    ap.add_argument("--option", action="append", default=[], metavar="NAME=VALUE",
                    help="cholamd_device_set_option switch for this run (e.g. program=0: level-by-level launches), repeatable; recorded in config")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--rank-arenas", type=int, default=0, help="--gpus N > 1: 1 = every rank's arenas by cholamd_device_alloc_arena (memory only under the rank's own "
                                                               "panels and the shared top: 7.5 of 34.8 GB per rank at 100^3 / world 8); default 0 = plain allocations "
                                                               "(no box with two GPUs has run the VMM arenas under RCCL yet: the driver's multi-GPU run keeps the plain ones)")
    ap.add_argument("--large-front", type=int, default=None, help="1: add the `large_front` object (60^3 fp64, 100^3 fp64 + mixed on this GPU) to the line; "
                                                                   "default: on for the metric's own configuration (no --case / --precision / --option, one GPU), else off")
    ap.add_argument("--sustained", type=int, default=2000, help="extra figure beside the metric: the same step for this many factorisations (0 = skip)")
    ap.add_argument("--in-flight", type=int, default=4, help="extra figure (not the metric): independent factorisations kept this many at a time on device objects / streams of their own; 0 or 1 = skip")
    args = ap.parse_args()
    if args.large_front is None:
        args.large_front = int(args.case is None and args.gpus == 1 and args.precision == "fp64" and not args.option)
    if args.case is None:
        args.case = "lapl_3375x3375"
    if args.gpus < 1 or args.gpus & (args.gpus - 1):
        sys.exit("bench.py: --gpus must be a power of two (the separator tree is cut at level log2(gpus))")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus))  # nothing has touched the GPU in this process

    import torch
    import torch.distributed as dist

    import cholesky_amd as ca
    from cholesky_amd import benchline, parallel

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks; refusing to report a mislabelled run")
    if torch.cuda.device_count() < max(world, 1):  # counting devices does not initialise the GPU
        sys.exit(f"bench.py: --gpus {args.gpus} needs {world} visible GPUs, found {torch.cuda.device_count()}")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device(f"cuda:{local_rank}"))
        if dist.get_world_size() != args.gpus:
            sys.exit(f"bench.py: process group of {dist.get_world_size()} ranks for --gpus {args.gpus}")
    else:
        torch.cuda.set_device(0)

    generated = args.case.startswith("gen:")
    if generated:
        parts = args.case.split(":")
        gn, glv, gtile = int(parts[1]), int(parts[2]), int(parts[3]) if len(parts) > 3 else 64
        files = None
        plan = ca.Problem(gn, gn, gn, glv, gtile).plan()
    else:
        g = os.path.join(ROOT, "tests", "golden", args.case)
        files = [os.path.join(g, f) for f in CASES[args.case]]
        plan = ca.Plan(*files)
    dev = ca.Device(plan, local_rank)
    for kv in args.option:
        k, v = kv.split("=")
        dev.set_option(k, int(v))
    if os.environ.get("CHOLAMD_SAVE_MAPS"):  # profiled runs (scripts/gpu_batch.sh): the address map of this process, so that the frames of a crash
        try:                                  # under the profiler can be attributed to a library afterwards (profiles/r4/README.md, the rc=139 of round 3)
            with open("/proc/self/maps") as src, open(os.environ["CHOLAMD_SAVE_MAPS"], "w") as dst:
                dst.write(src.read())
        except OSError:
            pass
    split = parallel.split_level(world)
    tail_off = parallel.tail_offset(plan, world)  # first panel of the shared top of the tree
    comm = None
    if world > 1:
        dev.set_partition(rank, world)
        comm = parallel.make_comm(dev, world, rank)  # libcholamd's own RCCL communicator (unique id broadcast by the process group)

    mixed = args.precision == "mixed"
    K, W = args.steps, args.warmup
    stream = torch.cuda.current_stream()
    per_arena = plan.arena_doubles * (4 if mixed else 8)
    n_arenas = max(1, min(K + W, int((8e9 if not generated else 4e10) // per_arena)))
    rank_arenas = bool(args.rank_arenas) and world > 1
    if rank_arenas:
        arenas = [dev.alloc_arena(4 if mixed else 8) for _ in range(n_arenas)]
    else:
        arenas = [dev.new_arena_f32() if mixed else dev.new_arena() for _ in range(n_arenas)]
    fill = dev.fill_f32 if mixed else dev.fill

    def refill():
        for a in arenas:
            fill(a, stream)
        dev.sync(stream)

    def step(a):
        # world > 1: local subtree levels, ONE RCCL all-reduce of the arena tail (extend-add
        # contributions to the shared ancestors), then the top levels
        if mixed and world > 1:
            dev.factor_sharded_f32(a, comm, stream)  # fp32 schedule sharded like the fp64 one; exchange and broadcasts on floats
        elif mixed:
            dev.factor_f32(a, stream)
        else:
            parallel.factor_sharded(dev, a, world, tail_off, stream, comm=comm)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    refill()
    done = 0
    for i in range(W):
        if i >= n_arenas:  # a factored arena is not an input: restore before re-using one
            refill()
        step(arenas[done % n_arenas])
        done += 1
    fence()
    if done + K > n_arenas:  # not enough pre-filled arenas for K more steps: restore them outside the timed region
        refill()
        done = 0
    fence()
    if K <= n_arenas:
        t0 = time.perf_counter()
        for i in range(K):
            step(arenas[(done + i) % n_arenas])
        fence()
        dt = time.perf_counter() - t0
    else:
        # arenas too large to hold K of them (generated problems): K steps timed one by one, the re-fill of the
        # arena between two steps outside the timed regions
        dt = 0.0
        for i in range(K):
            a = arenas[i % n_arenas]
            if i >= n_arenas:
                fill(a, stream)
            fence()
            t0 = time.perf_counter()
            step(a)
            fence()
            dt += time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    info = dev.info()  # raises on a negative (internal) code: no line is printed for a failed factorisation
    # every factorisation of the timed region factored the same matrix: the arenas must be bit-identical (a stalled or corrupted launch in the
    # middle of the run would otherwise go unseen: info is the last launch's)
    if world == 1 and K <= n_arenas and not all(bool(torch.equal(arenas[(done + i) % n_arenas], arenas[done % n_arenas])) for i in range(1, K)):
        sys.exit("bench.py: the factors of the timed steps differ from each other (a launch failed or is not deterministic)")
    if world > 1:
        bad = torch.tensor([1.0 if info[0] != 0 else 0.0], device="cuda")
        dist.all_reduce(bad, op=dist.ReduceOp.MAX)
        if bad.item() != 0:
            sys.exit(f"bench.py: factorisation failed on some rank (rank {rank}: info {info})")
    elif info[0] != 0:
        sys.exit(f"bench.py: factorisation failed: info {info}")

    # dominant-kernel roofline: HIP events recorded by the library around every launch, on the
    # stream the kernels run on, in a separate pass over pre-filled arenas
    refine = None
    if mixed and world > 1:  # the other half of the configuration, distributed like the factorisation: the fp32 factor stays on the ranks, every rank
        # sweeps its own subtrees and solves the shared top redundantly, only vectors travel (cholamd_solve_refine_sharded; no gather to rank 0)
        last = arenas[(done + K - 1) % n_arenas] if K <= n_arenas else arenas[(K - 1) % n_arenas]
        if generated:
            bvec = torch.from_numpy(ca.Problem(gn, gn, gn, glv, gtile).rhs()).cuda()
        else:
            bvec = torch.from_numpy(ca.plan.read_vector(os.path.join(os.path.dirname(files[0]), f"B_{plan.n}x1.mtx"), plan.n)).cuda()
        xvec = torch.empty_like(bvec)
        dev.solve_refine_sharded(last, bvec, xvec, comm, 30, 1e-11, stream)  # warm-up (work lists, CSR upload)
        fence()
        t0 = time.perf_counter()
        it, rel = dev.solve_refine_sharded(last, bvec, xvec, comm, 30, 1e-11, stream)
        fence()
        refine = {"corrections": it, "relres": rel, "ms": round((time.perf_counter() - t0) * 1e3, 3), "tol": 1e-11,
                  "distributed": "factor left on the ranks; two vector all-reduces per solve (cholamd_solve_refine_sharded)"}
    if mixed and rank == 0 and world == 1:  # the other half of the configuration: one right-hand side solved to fp64 accuracy with the fp32 factor
        if generated:
            bvec = torch.from_numpy(ca.Problem(gn, gn, gn, glv, gtile).rhs()).cuda()
        else:
            bvec = torch.from_numpy(ca.plan.read_vector(os.path.join(os.path.dirname(files[0]), f"B_{plan.n}x1.mtx"), plan.n)).cuda()
        xvec = torch.empty_like(bvec)
        dev.solve_refine(arenas[(done + K - 1) % n_arenas] if K <= n_arenas else arenas[(K - 1) % n_arenas], bvec, xvec, 30, 1e-11, stream)  # warm-up (work lists, CSR upload)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        it, rel = dev.solve_refine(arenas[(done + K - 1) % n_arenas] if K <= n_arenas else arenas[(K - 1) % n_arenas], bvec, xvec, 30, 1e-11, stream)
        torch.cuda.synchronize()
        refine = {"corrections": it, "relres": rel, "ms": round((time.perf_counter() - t0) * 1e3, 3), "tol": 1e-11}
    solve = None
    if not mixed and rank == 0 and world == 1:  # beside the metric: one right-hand side through the fp64 factor of the last step (the profiler passes of this command then hold the solve's kernels)
        if generated:
            bvec = torch.from_numpy(ca.Problem(gn, gn, gn, glv, gtile).rhs()).cuda()
        else:
            bvec = torch.from_numpy(ca.plan.read_vector(os.path.join(os.path.dirname(files[0]), f"B_{plan.n}x1.mtx"), plan.n)).cuda()
        xvec = torch.empty_like(bvec)
        last = arenas[(done + K - 1) % n_arenas] if K <= n_arenas else arenas[(K - 1) % n_arenas]
        dev.solve(last, bvec, xvec, stream)  # warm-up (work lists)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        dev.solve(last, bvec, xvec, stream)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3
        solve = {"ms": round(ms, 3), "relres": dev.residual(bvec, xvec)}
    # not the metric -- reported beside it: independent factorisations (a batch of matrices, a parameter sweep) kept TWO in
    # (--in-flight) in flight, each on a device object and stream of its own.  A factorisation of this size ends in a dependency chain that keeps a
    # handful of the 256 CUs busy; the next one's leaves fill the rest.  `value` above stays the one-after-the-other rate.
    concurrent = None
    NF = args.in_flight
    if world == 1 and not mixed and not generated and K <= n_arenas and K >= 2 * NF and NF >= 2:
        devs, streams = [dev], [stream]
        for q in range(1, NF):
            dq = ca.Device(plan, local_rank)
            for kv in args.option:
                k, v = kv.split("=")
                dq.set_option(k, int(v))
            devs.append(dq)
            streams.append(torch.cuda.Stream())
        refill()
        for i in range(2 * NF):  # warm-up of the other objects (descriptor upload, counters)
            devs[i % NF].factor(arenas[i], streams[i % NF])
        torch.cuda.synchronize()
        refill()
        t0 = time.perf_counter()
        for i in range(K):
            devs[i % NF].factor(arenas[i], streams[i % NF])
        torch.cuda.synchronize()
        dt2 = time.perf_counter() - t0
        if all(d.info()[0] == 0 for d in devs):
            same = all(bool(torch.equal(arenas[0], arenas[q])) for q in range(1, NF))  # factors of the same matrix by the different objects
            concurrent = {"in_flight": NF, "value": round(plan.flops * K / dt2 * 1e-9, 3), "unit": "GF/s", "ms_per_step": round(dt2 / K * 1e3, 5), "steps": K,
                          "factors_identical": same,
                          "note": f"K independent factorisations, {NF} at a time ({NF} device objects, {NF} HIP streams); throughput of a batch, not the latency of one factorisation"}
        del devs[1:]
    # sustained rate: the same step back to back for >= 2 000 steps (the 20-step figure of the driver's command is a 3.5 ms
    # burst; clocks and caches settle differently over seconds).  Single GPU, the metric's configuration only.
    sustained = None
    NS = args.sustained
    if world == 1 and not mixed and not generated and NS > 0:
        pool = arenas + [dev.new_arena() for _ in range(max(0, min(NS, int(8e9 // per_arena)) - n_arenas))]  # bursts as long as 8 GB of pre-filled arenas allow
        t_s, n_s = 0.0, 0
        while n_s < NS:
            for a in pool:
                fill(a, stream)
            fence()
            k = min(len(pool), NS - n_s)
            t0 = time.perf_counter()
            for i in range(k):
                step(pool[i])
            fence()
            t_s += time.perf_counter() - t0
            n_s += k
        sustained = {"steps": n_s, "value": round(plan.flops * n_s / t_s * 1e-9, 3), "unit": "GF/s", "ms_per_step": round(t_s / n_s * 1e3, 5),
                     "note": f"{n_s} factorisations back to back in bursts of {len(pool)} pre-filled arenas (re-fills between bursts outside the timed regions)"}
        del pool
    # dominant-kernel roofline: HIP events recorded by the library around every launch (and, sharded, around the RCCL exchange and
    # broadcasts), on the stream they run on, in a separate pass over pre-filled arenas -- the SAME call as the timed step
    refill()
    fence()
    dev.set_timing(1)
    reps = min(n_arenas, 20)
    for a in arenas[:reps]:
        step(a)
    dev.sync(stream)
    timing = dev.get_timing_ex()
    dev.set_timing(0)
    ev_ms = dev.event_overhead_ms(stream)  # what an empty (record, record) pair reads on this stream
    rccl_ranks = comm.count() if comm is not None else None
    xvol, eb_x = dev.exchange_volume(), (4 if mixed else 8)
    timing_all = [timing]
    if world > 1:
        timing_all = [None] * world
        dist.all_gather_object(timing_all, timing)

    if rank == 0:
        calls, flops = plan.counts()
        blocks = plan.blocks
        sizes = {int(b[1]): int(b[4] - b[2] + 1) for b in blocks if b[0] == b[1]}
        kernel0 = benchline.kernel_names(mixed, generated, all(t["trsm"][1] == 0 for t in timing_all))[benchline.dominant(timing_all)[0]].split(" + ")[0]
        rec = {
            "world": world, "steps": K, "warmup": W, "dt_s": dt, "case": args.case, "generated": generated, "mixed": mixed,
            "options": args.option + (["rank_arenas=1"] if rank_arenas else []), "info": list(info),
            "plan": {"n": plan.n, "levels": plan.levels, "nsep": plan.nsep, "flops": plan.flops, "alg_bytes": plan.alg_bytes,
                     "flops_by_kind": {"potrf": float(flops[0]), "trsm": float(flops[1]), "update": float(flops[2] + flops[3])},
                     "piv_entries": sum(n * (n + 1) // 2 for n in sizes.values()),
                     "anc_entries": sum(stored_rows(plan, int(b[0]), int(b[1]), int(b[4] - b[2] + 1)) * sizes[int(b[1])] for b in blocks if b[0] != b[1])},
            "workload": (f"generated {gn}^3 7-point Laplacian, geometric nested dissection ({glv} levels, tile {gtile}), cholamd_generate_laplacian" if generated else
                         f"tests/{args.case} (reference fixture: 7-point Laplacian, nested-dissection ordering from the fixture files)"),
            "data": "synthetic (generated Laplacian, ordering and clusters)" if generated else "reference fixture (matrix, ordering and cluster files from the reference's tests/, copied as data); no random data anywhere",
            "timing": [{k: list(v) for k, v in t.items()} for t in timing_all], "reps": reps, "event_pair_ms": ev_ms,
            "rccl_ranks": rccl_ranks,
            "exchange": (None if world == 1 else
                         f"owner-directed exchange of the arena tail: {xvol[3]} column blocks, grouped ncclSend / ncclRecv to their owners + rank-ordered sum "
                         f"(rank 0 receives {xvol[0] * eb_x / 1e6:.2f} MB, sends {xvol[1] * eb_x / 1e6:.2f} MB of a {xvol[2] * eb_x / 1e6:.2f} MB tail)" if xvol[3] > 0 else
                         f"one in-place ncclAllReduce of the {xvol[2] * eb_x / 1e6:.2f} MB arena tail (top levels replicated)"),
            "profile": profile_numbers(kernel0, args.case, mixed, args.option) if world == 1 else {},
            "refinement": refine, "solve": solve, "concurrent": concurrent, "sustained": sustained,
            "fp64_mfma_sustained_measured": SUSTAINED_FP64_MFMA_TFLOPS,
            "fp64_mfma_sustained_note": SUSTAINED_NOTE,
        }
        if not args.no_cpu_baseline and not generated:  # rank 0's host, at every world size
            rec["cpu_baseline"] = cpu_baseline(files, plan.flops)
        if args.large_front and world == 1:
            del arenas[:]
            torch.cuda.empty_cache()
            rec["large_front"] = large_front(ca, torch, stream)
        print(json.dumps(benchline.assemble(rec)))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
