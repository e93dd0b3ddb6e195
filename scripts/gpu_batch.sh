#!/bin/bash
# One gpurun call's worth of checks (development aid): GPU test suite, microbenchmarks, the bench line, rocprofv3
# kernel stats and PMC passes.  Usage: scripts/gpu_batch.sh <tag> [steps...]; outputs under gpurun_out/<tag>/.
# A step that times out stops the batch (no further GPU step after a kill).
tag=$1; shift
steps=${@:-"tests chol16 bench stats pmc"}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
[ -z "$GRAFT_REPO_ROOT" ] && out=$(pwd)/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
run() { # name, seconds, command...   (progress lines go to stderr: stdout may be redirected into a result file)
  local name=$1 secs=$2; shift 2
  echo "== $name $(date +%T)" >&2
  timeout -k 10 $secs "$@"; local rc=$?
  echo "== $name rc=$rc" >&2
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping"; exit 1; fi
  return $rc
}
for s in $steps; do case $s in
tests)  run tests 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; tail -5 $out/pytest.log ;;
chol16) run chol16 60 ./scripts/chol16_bench > $out/chol16.txt 2>&1; cat $out/chol16.txt ;;
bench)  run bench 300 python bench.py --steps 200 --warmup 10 > $out/bench.json 2> $out/bench.err; cat $out/bench.json; tail -3 $out/bench.err ;;
stats)  (cd /tmp && run stats 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o s -- python3 $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --no-cpu-baseline > $out/bench_under_rocprof.json 2> $out/stats.err); find $out/stats -name "*kernel_stats.csv" -exec head -8 {} \; ;;
pmc)    for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
          n=$(echo $grp | cut -d' ' -f1)
          (cd /tmp && run pmc_$n 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/pmc_$n -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 2 --no-cpu-baseline --sustained 0 > $out/pmc_$n.json 2> $out/pmc_$n.err) || tail -5 $out/pmc_$n.err
        done ;;
gen)    for c in gen:40:6 gen:60:8; do n=$(echo $c | tr ':' '_')
          (cd /tmp && run stats_$n 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$n -o s -- python3 $GRAFT_REPO_ROOT/bench.py --case $c --steps 5 --warmup 1 > $out/bench_$n.json 2> $out/stats_$n.err); cat $out/bench_$n.json
          (cd /tmp && run pmc_$n 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_$n -o p -- python3 $GRAFT_REPO_ROOT/bench.py --case $c --steps 3 --warmup 1 > $out/pmc_$n.json 2> $out/pmc_$n.err) || tail -5 $out/pmc_$n.err
        done ;;
mixed)  for c in gen:60:8 gen:100:10; do n=$(echo $c | tr ':' '_')
          run mixed_$n 600 python bench.py --case $c --precision mixed --steps 3 --warmup 1 > $out/bench_mixed_$n.json 2> $out/bench_mixed_$n.err; cat $out/bench_mixed_$n.json; tail -3 $out/bench_mixed_$n.err
        done ;;
mixedtests) run mixedtests 900 python -m pytest tests/test_gpu_mixed.py -x -q > $out/pytest_mixed.log 2>&1; tail -15 $out/pytest_mixed.log ;;
prog)   run prog 180 python scripts/prog_check.py > $out/prog_check.txt 2>&1; cat $out/prog_check.txt ;;
proggen) run proggen 240 python scripts/prog_check.py gen:12,12,12,4,16 gen:20,20,20,5,64 gen:10,9,8,5,8 > $out/prog_gen.txt 2>&1; cat $out/prog_gen.txt ;;
trace)  run trace 120 python scripts/prog_trace.py lapl_3375x3375 > $out/prog_trace.txt 2>&1; cat $out/prog_trace.txt
        run trace0 120 python scripts/prog_trace.py lapl_3375x3375 follow=0 > $out/prog_trace_nofollow.txt 2>&1; cat $out/prog_trace_nofollow.txt ;;
profiles) # every rocprofv3 pass behind profiles/rN/summary.json: key | bench arguments | steps
        SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU"
        while IFS='|' read -r key bargs nsteps mops traffic; do
          [ -z "$key" ] && continue
          [ -n "$PROFILE_KEYS" ] && ! echo " $PROFILE_KEYS " | grep -q " $key " && continue
          d=$out/$key; mkdir -p $d
          export CHOLAMD_SAVE_MAPS=$d/proc_maps.txt   # bench.py writes /proc/self/maps there: a crash under the profiler can be attributed to a library
          (cd /tmp && run stats_$key 500 rocprofv3 --kernel-trace --stats --output-format csv -d $d/stats -o s -- python3 $GRAFT_REPO_ROOT/bench.py $bargs --steps $nsteps --warmup 2 --no-cpu-baseline --sustained 0 > $d/bench_under_rocprof.json 2> $d/stats.err) || tail -3 $d/stats.err
          (cd /tmp && run pmc_$key 500 rocprofv3 --kernel-trace --pmc $SQ $mops GRBM_GUI_ACTIVE --output-format csv -d $d/pmc_SQ -o p -- python3 $GRAFT_REPO_ROOT/bench.py $bargs --steps $nsteps --warmup 2 --no-cpu-baseline --sustained 0 > $d/pmc_SQ.json 2> $d/pmc_SQ.err) || tail -3 $d/pmc_SQ.err
          if [ "$traffic" = "1" ]; then for c in FETCH_SIZE WRITE_SIZE; do
            (cd /tmp && run ${c}_$key 500 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d/pmc_$c -o p -- python3 $GRAFT_REPO_ROOT/bench.py $bargs --steps $nsteps --warmup 2 --no-cpu-baseline --sustained 0 > $d/pmc_$c.json 2> $d/pmc_$c.err) || tail -3 $d/pmc_$c.err
          done; fi
          rm -f $d/*/*_kernel_trace.csv $d/*/*agent_info.csv   # per-dispatch traces are large; the stats / counter CSVs are what is kept
        done <<'LIST'
lapl_3375|--case lapl_3375x3375 --in-flight 0|20|SQ_INSTS_VALU_MFMA_MOPS_F64|1
lapl_3375_levels|--case lapl_3375x3375 --option program=0 --in-flight 0|20|SQ_INSTS_VALU_MFMA_MOPS_F64|1
gen_40_6|--case gen:40:6|5|SQ_INSTS_VALU_MFMA_MOPS_F64|0
gen_60_8|--case gen:60:8|3|SQ_INSTS_VALU_MFMA_MOPS_F64|1
gen_60_8_mixed|--case gen:60:8 --precision mixed|3|SQ_INSTS_VALU_MFMA_MOPS_F32|1
gen_100_10_mixed|--case gen:100:10 --precision mixed|2|SQ_INSTS_VALU_MFMA_MOPS_F32|1
gen_100_10|--case gen:100:10|2|SQ_INSTS_VALU_MFMA_MOPS_F64|1
LIST
        ;;
super)  for sb in 1 2 3 4; do for c in gen:40:6 gen:60:8; do
          run super_${sb}_$c 300 python bench.py --case $c --steps 3 --warmup 1 --option super_blocks=$sb 2> /dev/null | python3 -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('super_blocks=$sb', '$c', d['value'], 'GF/s', d['ms_per_step'], 'ms', d['roofline']['kernel_ms_per_step_events_raw'])"
        done; done
        for sb in 1 3; do run supermixed_$sb 400 python bench.py --case gen:100:10 --precision mixed --steps 2 --warmup 1 --option super_blocks=$sb 2> /dev/null | python3 -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('mixed super_blocks=$sb gen:100:10', d['value'], 'GF/s', d['ms_per_step'], 'ms', d['config']['refinement'])"; done ;;
*) echo "unknown step $s" ;;
esac; done
echo "batch done"
