"""bench.py's JSON line as a pure function of a measurement record (cholesky_amd/benchline.py): the N > 1 line carries `roofline` and
`cpu_baseline` like the single-GPU one (BASELINE.json north_star "reported at 1, 2, 4 and 8 GPUs as
absolute GF/s and as fraction of roofline, next to the reference's own CPU path ... in the same run")."""
import json

import pytest

from cholesky_amd import benchline

PLAN = {"n": 3375, "levels": 5, "nsep": 31, "flops": 1.48552e8, "alg_bytes": 2932064,
        "flops_by_kind": {"potrf": 3.35e7, "trsm": 5.88e7, "update": 5.63e7}, "piv_entries": 400000, "anc_entries": 1300000}


def record(world, program=False):
    timing = []
    for r in range(world):
        if program:
            t = {"potrf": [0.0, 0], "trsm": [0.0, 0], "update": [0.0, 0], "other": [3.6, 20], "exchange": [0.0, 0], "bcast": [0.0, 0]}
        else:  # level-by-level launches of a sharded run; rank 1 is the slow one
            t = {"potrf": [2.0 + 0.5 * (r == 1), 140], "trsm": [0.0, 0], "update": [1.0, 120], "other": [0.0, 0],
                 "exchange": [0.8, 20], "bcast": [0.0, 0]}
        timing.append(t)
    return {"world": world, "steps": 20, "warmup": 5, "dt_s": 20 * 2.1e-4, "case": "lapl_3375x3375", "generated": False, "mixed": False,
            "options": [], "info": [0, 0], "plan": PLAN, "workload": "tests/lapl_3375x3375", "data": "reference fixture",
            "timing": timing, "reps": 20, "event_pair_ms": 0.0047, "rccl_ranks": world if world > 1 else None,
            "exchange": "one ncclReduce per owned piece", "profile": {"hbm_bytes_per_launch": 5.8e7, "avg_launch_us": 173.6, "source": "x"} if world == 1 else {},
            "cpu_baseline": {"value": 24.2, "unit": "GF/s", "cores": 1, "kind": "port", "sample": "400 factorisations"}}


@pytest.mark.parametrize("world", [1, 2, 8])
def test_line_carries_roofline_and_cpu_baseline(world):
    out = benchline.assemble(record(world, program=(world == 1)))
    json.dumps(out)  # serialisable
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in out, key
    assert out["n_gpus"] == world and out["vs_baseline"] is None and out["scaling"] == "strong"
    rf = out["roofline"]
    assert rf["achieved"] is not None and rf["achieved"] > 0
    assert rf["frac"] is not None and rf["frac"] == pytest.approx(rf["achieved"] / rf["peak"], rel=1e-3)
    assert rf["peak"] == pytest.approx(78.6 * world)
    assert out["cpu_baseline"]["kind"] == "port" and out["cpu_baseline"]["cores"] == 1
    assert out["value"] == pytest.approx(1.48552e8 / 2.1e-4 * 1e-9, rel=1e-6)
    if world == 1:
        assert rf["kernel"] == "k_program" and rf["traffic"] == 5.8e7 and rf["exchange_ms"] is None
        assert out["config"]["rccl_ranks"] is None
        assert rf["achieved"] == pytest.approx(1.48552e8 / ((3.6 / 20 - 0.0047) * 1e-3) * 1e-12, rel=1e-3)
    else:
        assert rf["kernel"] == "k_potrf_trsm" and rf["slowest_rank"] == 1
        assert rf["traffic"] is None and "no counter pass" in rf["traffic_note"]
        assert rf["exchange_ms"] == pytest.approx(0.04) and out["config"]["rccl_ranks"] == world
        # POTRF + fused TRSM flops of the whole job over the slowest rank's time in that kernel
        t_kernel = (2.5 / 140 - 0.0047) * 1e-3 * 7
        assert rf["achieved"] == pytest.approx((3.35e7 + 5.88e7) / t_kernel * 1e-12, rel=1e-3)


def test_rank_count_must_match():
    rec = record(2)
    rec["timing"] = rec["timing"][:1]
    with pytest.raises(ValueError):
        benchline.assemble(rec)


def lf_record(mixed):
    # one factorisation's event times: the macro-tile update kernel carries the step
    timing = {"potrf": [12.0, 40], "trsm": [9.0, 30], "update": [400.0, 41], "other": [0.0, 0], "exchange": [0.0, 0], "bcast": [0.0, 0]}
    rec = {"case": "gen:100:10", "mixed": mixed, "step_s": [0.48, 0.47], "info": [0, 0], "event_pair_ms": 0.005,
           "plan": {"n": 1000000, "flops": 2.0e13, "alg_bytes": 8.1e10, "arena_bytes": 3.3e10 if not mixed else 1.65e10,
                    "flops_by_kind": {"potrf": 1.0e11, "trsm": 9.0e11, "update": 1.9e13}},
           "timing": timing, "profile": {"mfma_busy_frac": 0.61, "hbm_bytes_per_launch": 2.4e9, "avg_launch_us": 9800.0, "source": "profiles/r4/summary.json: x"}}
    if mixed:
        rec["refinement"] = {"corrections": 3, "relres": 1.3e-13, "ms": 64.0, "tol": 1e-11}
    else:
        rec["solve"] = {"ms": 22.0, "relres": 3e-15}
    return rec


@pytest.mark.parametrize("mixed", [False, True])
def test_large_front_entry(mixed):
    e = benchline.large_front_entry(lf_record(mixed))
    json.dumps(e)
    assert e["value"] == pytest.approx(2.0e13 / 0.475 * 1e-9, rel=1e-6) and e["unit"] == "GF/s" and e["ms_per_step"] == pytest.approx(475.0)
    rf = e["roofline"]
    assert rf["kernel"] == ("k32_update_mt" if mixed else "k_update_mt") and rf["kind"] == "update"
    assert rf["peak"] == (157.3 if mixed else 78.6)
    # the update kind's algorithmic flops over its summed launch time (event-pair reading taken off every launch)
    assert rf["achieved"] == pytest.approx(1.9e13 / ((400.0 - 41 * 0.005) * 1e-3) * 1e-12, rel=1e-3)
    assert rf["frac"] == pytest.approx(rf["achieved"] / rf["peak"], rel=1e-3)
    assert rf["mfma_busy_frac_rocprof"] == 0.61 and rf["hbm_bytes_per_launch_rocprof"] == 2.4e9
    assert ("refinement" in e) == mixed and ("solve" in e) == (not mixed)
    with pytest.raises(ValueError):
        benchline.large_front_entry(dict(lf_record(mixed), step_s=[]))


def test_line_carries_large_front():
    rec = record(1, program=True)
    rec["large_front"] = {"gen_100_10": lf_record(False), "gen_100_10_mixed": lf_record(True)}
    out = benchline.assemble(rec)
    json.dumps(out)
    lf = out["large_front"]
    assert set(lf) == {"gen_100_10", "gen_100_10_mixed", "note"}
    assert lf["gen_100_10"]["dtype"] == "f64" and lf["gen_100_10_mixed"]["refinement"]["relres"] < 1e-10
    assert out["value"] == pytest.approx(1.48552e8 / 2.1e-4 * 1e-9, rel=1e-6)  # the metric is untouched
    assert "large_front" not in benchline.assemble(record(1, program=True))
