"""The JSON line of bench.py as a PURE function of a measurement record (no torch, no GPU): unit-tested on the CPU for world
1 / 2 / 8 records (tests/test_benchline.py), so that the first multi-GPU driver run prints a line that carries `roofline`
and `cpu_baseline` like the single-GPU one (BASELINE.json north_star: "reported at 1, 2, 4 and 8 GPUs as absolute GF/s and
as fraction of roofline, next to the reference's own CPU path ... in the same run").

Record (`rec`, plain Python types):
  world, steps, warmup, dt_s (max over ranks of the timed region), case, generated, mixed, options, info
  plan = {n, levels, nsep, flops, alg_bytes, flops_by_kind = {potrf, trsm, update}, piv_entries, anc_entries}
  workload (text), data (text)
  timing = per rank: {kind: [ms over `reps` factorisations, launches]} for kind in potrf / trsm / update / other / exchange / bcast
  reps, event_pair_ms
  rccl_ranks (ncclCommCount of the library's communicator, None at world 1), exchange = text
  profile = profile_numbers() of the dominant kernel ({} when no committed profile matches), cpu_baseline, concurrent,
  refinement, sustained (optional dicts, passed through)
  large_front (optional) = {key: record of large_front_entry()}: the large-front regime (generated 60^3 / 100^3 Laplacians, fp64 and
  mixed) measured in the same run on one GPU -- the fixtures of the metric never reach a front that is "a real dense contraction"
"""

PEAK_FP64_TFLOPS = 78.6    # MI355X dense fp64 (vector = matrix) peak, public spec; the microarch guide lists no fp64 row
PEAK_FP32_TFLOPS = 157.3   # MI355X_MICROARCH.md: f32-input MFMA = the fp32 vector rate
PEAK_HBM_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
KINDS = ("potrf", "trsm", "update", "other")


def kernel_names(mixed, generated, fused):
    if mixed:
        return {"potrf": "k32_potrf", "trsm": "k32_trsm_wt", "update": "k32_update_mt + k32_update", "other": "k_program"}
    return {"potrf": "k_potrf_trsm" if fused else "k_potrf_rr", "trsm": "k_trsm_rr", "update": "k_update_mt + k_update" if generated else "k_update", "other": "k_program"}


def dominant(timing):
    """(kind, rank) of the compute kernel kind with the largest time on any rank: the job waits for its slowest rank."""
    best = None
    for r, t in enumerate(timing):
        for k in KINDS:
            ms = t.get(k, [0.0, 0])[0]
            if best is None or ms > best[0]:
                best = (ms, k, r)
    return best[1], best[2]


def large_front_entry(rec):
    """One entry of the line's `large_front` object, from a measurement record:
      case, mixed, steps, step_s (list: wall-clock of each timed factorisation, device-synchronised on both sides),
      plan = {n, flops, alg_bytes, arena_bytes, flops_by_kind}, timing = {kind: [ms, launches]} of ONE factorisation (HIP events of the
      library around each launch), event_pair_ms, info, refinement (mixed: {corrections, relres, ms}), solve (fp64: {ms, relres}),
      profile = profile_numbers() of the dominant kernel ({} when no committed counter pass matches)
    value = F_ref / mean step (GF/s); the dominant kernel kind's ALGORITHMIC flops (the reference's BLAS calls of that kind) over its summed
    launch time against the dense MFMA peak of the arithmetic type."""
    mixed = bool(rec.get("mixed"))
    steps = [float(t) for t in rec["step_s"]]
    if not steps or min(steps) <= 0.0:
        raise ValueError("large_front: no timed step")
    plan, timing = rec["plan"], rec["timing"]
    mean_s = sum(steps) / len(steps)
    peak = PEAK_FP32_TFLOPS if mixed else PEAK_FP64_TFLOPS
    kinds = dict(plan["flops_by_kind"])
    fused = timing.get("trsm", [0.0, 0])[1] == 0
    if fused:
        kinds["potrf"] = kinds["potrf"] + kinds["trsm"]
    dom = max((k for k in ("potrf", "trsm", "update")), key=lambda k: timing.get(k, [0.0, 0])[0])
    ms, n_launch = timing[dom]
    kernel_s = max((ms - float(rec.get("event_pair_ms", 0.0)) * n_launch) * 1e-3, 1e-9)
    achieved = kinds[dom] / kernel_s * 1e-12
    prof = rec.get("profile") or {}
    out = {
        "case": rec["case"], "dtype": "f32 factor + f64 iterative refinement" if mixed else "f64", "n": plan["n"],
        "value": round(plan["flops"] / mean_s * 1e-9, 3), "unit": "GF/s", "ms_per_step": round(mean_s * 1e3, 4), "steps": len(steps),
        "F_ref_flops": plan["flops"], "B_alg_bytes": plan["alg_bytes"], "arena_bytes": plan.get("arena_bytes"), "factor_info": list(rec.get("info", (0, 0))),
        "flops_note": "value = F_ref / time: F_ref counts the reference's BLAS calls on its filled cluster blocks (dense); the level schedule leaves out what is "
                      "structurally zero inside the LEAVES' blocks (option leaf_envelope: identical factors), so the executed flops are fewer",
        "whole_step_frac_of_peak": round(plan["flops"] / mean_s * 1e-12 / peak, 5),
        "roofline": {"bound": "mfma", "kernel": kernel_names(mixed, True, fused)[dom].split(" + ")[0], "kind": dom,
                     "achieved": round(achieved, 4), "peak": peak, "unit": "TFLOP/s", "frac": round(achieved / peak, 5),
                     "kernel_ms_per_step": round(kernel_s * 1e3, 4), "launches_per_step": n_launch, "alg_flops_of_kind": kinds[dom],
                     "kernel_ms_per_step_events_raw": {k: round(timing.get(k, [0.0, 0])[0], 4) for k in KINDS},
                     "mfma_busy_frac_rocprof": prof.get("mfma_busy_frac"), "hbm_bytes_per_launch_rocprof": prof.get("hbm_bytes_per_launch"),
                     "avg_launch_us_rocprof": prof.get("avg_launch_us"), "profile_source": prof.get("source")},
    }
    if rec.get("refinement") is not None:
        out["refinement"] = rec["refinement"]
    if rec.get("solve") is not None:
        out["solve"] = rec["solve"]
    return out


def assemble(rec):
    world, K, W, dt = int(rec["world"]), int(rec["steps"]), int(rec["warmup"]), float(rec["dt_s"])
    plan, timing, reps = rec["plan"], rec["timing"], max(int(rec["reps"]), 1)
    mixed, generated = bool(rec.get("mixed")), bool(rec.get("generated"))
    ev_ms = float(rec.get("event_pair_ms", 0.0))
    if len(timing) != world:
        raise ValueError(f"timing of {len(timing)} ranks for a world of {world}")
    eb = 4.0 if mixed else 8.0
    kinds = dict(plan["flops_by_kind"])
    kinds["other"] = float(plan["flops"])
    program = any(t.get("other", [0.0, 0])[1] > 0 for t in timing)  # the whole factorisation as ONE launch (k_program), timed as "other"
    fused = all(t.get("trsm", [0.0, 0])[1] == 0 for t in timing)
    if fused:  # POTRF and TRSM of a column-block step in one launch (k_potrf_trsm), timed as "potrf"
        kinds["potrf"] = kinds["potrf"] + kinds["trsm"]
    piv, anc = float(plan["piv_entries"]), float(plan["anc_entries"])
    panel_bytes = {"potrf": 2 * eb * piv + (2 * eb * anc + eb * piv if fused else 0.0), "trsm": 2 * eb * anc + eb * piv, "update": None,
                   "other": 2 * eb * (piv + anc)}
    dom, dom_rank = dominant(timing)
    ms, n_launch = timing[dom_rank][dom]
    launches_per_step = n_launch / reps
    avg_s_raw = ms / max(n_launch, 1) * 1e-3
    avg_s = max(avg_s_raw - ev_ms * 1e-3, 1e-9)
    # ALGORITHMIC flops of that kernel kind over the whole job (the reference's BLAS calls of that kind, every rank's share) per launch
    # of the slowest rank, over that rank's average launch duration: what the job gets out of the kernel.  Against world x the peak of
    # one GPU.  (Replicated top levels execute the same call on every rank: counted once -- algorithmic, not executed, flops.)
    flops_per_launch = kinds[dom] / max(launches_per_step, 1e-9)
    achieved = flops_per_launch / avg_s * 1e-12
    peak1 = PEAK_FP32_TFLOPS if mixed else PEAK_FP64_TFLOPS
    peak = peak1 * world
    value = float(plan["flops"]) * K / dt * 1e-9
    names = kernel_names(mixed, generated, fused)
    prof = rec.get("profile") or {}

    def per_step_max(kind):
        return max(t.get(kind, [0.0, 0])[0] for t in timing) / reps

    exchange_ms = per_step_max("exchange") if world > 1 else 0.0
    bcast_ms = per_step_max("bcast") if world > 1 else 0.0
    if world > 1:
        traffic, traffic_note = None, ("no counter pass exists for an N-GPU run (rocprofv3 --pmc is collected on the 1-GPU box; the driver "
                                       "owns the multi-GPU node): HBM bytes per launch are reported for n_gpus = 1 only")
        par = (f"subtree-sharded x{world}: levels below tree level log2({world}) per rank, {rec.get('exchange', 'RCCL exchange of the arena tail')}, "
               "then the top levels (replicated, or distributed by column blocks with ncclBroadcast: option dist_top)")
    else:
        traffic = prof.get("hbm_bytes_per_launch")
        traffic_note = ("HBM bytes per launch of that kernel, (2*FETCH_SIZE + WRITE_SIZE)*1024 from the committed rocprofv3 --pmc passes of this command ("
                        + prof.get("source", "no profile committed for this case") + "); to compare with alg_bytes_per_launch")
        par = "single GPU"
    out = {
        "metric": "fp64 factorization GF/s (numeric level loop, F_ref flops) on 3-D Laplacian" if not mixed else
                  "fp32 factorization GF/s (numeric level loop, F_ref flops) on 3-D Laplacian; solution refined to fp64 accuracy",
        "value": round(value, 3), "unit": "GF/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": round(dt / K * 1e3, 5), "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f32 factor + f64 iterative refinement" if mixed else "f64",
        "data": rec["data"],
        "config": {"workload": rec["workload"], "n": plan["n"], "levels": plan["levels"], "separators": plan["nsep"], "F_ref_flops": plan["flops"],
                   "B_alg_bytes": plan["alg_bytes"], "parallelism": par, "rccl_ranks": rec.get("rccl_ranks"),
                   "precision": "mixed" if mixed else "fp64", "options": list(rec.get("options", [])), "factor_info": list(rec.get("info", (0, 0)))},
        "roofline": {"bound": "mfma", "kernel": names[dom], "achieved": round(achieved, 5),
                     "peak": peak, "unit": "TFLOP/s", "frac": round(achieved / peak, 6),
                     "peak_note": f"{world} x {peak1} TFLOP/s" if world > 1 else None,
                     "traffic": traffic, "traffic_note": traffic_note,
                     "alg_bytes_per_launch": plan["alg_bytes"] / max(launches_per_step, 1e-9),
                     "alg_bytes_note": "SURVEY 8(d): B_alg = 8 (nnz(tril A) + nnz(L)) per factorisation / launches of the kernel per factorisation; dense_panel_bytes_per_launch = what the stored panels (row-compacted; the structural zeros inside the kept tiles included) make those launches move",
                     "dense_panel_bytes_per_launch": (panel_bytes[dom] / max(launches_per_step, 1e-9)) if panel_bytes[dom] else None,
                     "avg_launch_us": round(avg_s * 1e6, 2), "avg_launch_us_events_raw": round(avg_s_raw * 1e6, 2), "event_pair_overhead_us": round(ev_ms * 1e3, 2),
                     "slowest_rank": dom_rank,
                     "avg_launch_us_rocprof": prof.get("avg_launch_us") if world == 1 else None,
                     "mfma_busy_frac_rocprof": prof.get("mfma_busy_frac") if world == 1 else None,
                     "launches_per_step": launches_per_step,
                     "alg_flops_per_launch": flops_per_launch,
                     "exchange_ms": round(exchange_ms, 5) if world > 1 else None,
                     "bcast_ms": round(bcast_ms, 5) if world > 1 else None,
                     "exchange_note": ("HIP events on the launch stream around the RCCL extend-add exchange (reduce of the arena tail) and around the grouped "
                                       "ncclBroadcasts of the distributed top levels; per step, max over ranks (a rank that arrives early waits inside the collective)") if world > 1 else None,
                     "whole_step_frac_of_peak": round(value * 1e-3 / peak, 6),
                     "whole_step_alg_GBs": round(plan["alg_bytes"] * K / dt * 1e-9, 3), "hbm_peak_GBs": PEAK_HBM_GBS * world,
                     "kernel_ms_per_step_events_raw": {k: round(per_step_max(k), 5) for k in KINDS},
                     "launch_structure": "one program launch per factorisation (resident workgroups, job queue, followers)" if program else
                                         "per level and column-block step: fused POTRF+TRSM launch + update launch(es)"},
    }
    for key in ("fp64_mfma_sustained_measured", "fp64_mfma_sustained_note"):
        if rec.get(key) is not None and not mixed:
            out["roofline"][key] = rec[key]
    if rec.get("refinement") is not None:
        out["config"]["refinement"] = rec["refinement"]
    if rec.get("solve") is not None:
        out["config"]["solve"] = rec["solve"]
    if rec.get("sustained") is not None:
        out["sustained"] = rec["sustained"]
    if rec.get("concurrent") is not None:
        out["concurrent"] = rec["concurrent"]
    if rec.get("cpu_baseline") is not None:  # rank 0's host, whatever the world size
        out["cpu_baseline"] = rec["cpu_baseline"]
    if rec.get("large_front"):  # the large-front regime beside the metric (never in `value`)
        out["large_front"] = {k: large_front_entry(v) for k, v in rec["large_front"].items()}
        out["large_front"]["note"] = ("generated N^3 7-point Laplacians (cholamd_generate_laplacian: gen:N:levels), one GPU, same run: whole numeric factorisation "
                                      "timed per step with device synchronisation on both sides, F_ref flops; `roofline` = the dominant kernel kind's algorithmic "
                                      "flops over its summed launch time (HIP events of the library around every launch) against the dense MFMA peak")
    return out
