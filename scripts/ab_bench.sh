#!/bin/bash
# Development aid: interleaved A/B of the bench line between libcholamd.so builds (one gpurun call, one box):
#   scripts/ab_bench.sh OUT ROUNDS NAME=path/to/libcholamd.so ...   (NAME=default: the in-tree build)
out=$1; rounds=$2; shift 2
mkdir -p $out
for r in $(seq 1 $rounds); do
  for kv in "$@"; do
    name=${kv%%=*}; lib=${kv#*=}
    if [ "$lib" = "default" ]; then unset CHOLAMD_LIB; else export CHOLAMD_LIB=$(pwd)/$lib; fi
    timeout -k 10 300 python bench.py --steps 200 --warmup 10 --no-cpu-baseline --sustained 0 --in-flight 0 --large-front 0 ${AB_ARGS} > $out/ab_${name}_$r.json 2> $out/ab_${name}_$r.err || { echo "$name round $r failed"; tail -3 $out/ab_${name}_$r.err; exit 1; }
    python3 -c "import json; d=json.load(open('$out/ab_${name}_$r.json')); print('$name', $r, d['value'], 'GF/s', round(d['ms_per_step']*1e3,2), 'us')"
  done
done
