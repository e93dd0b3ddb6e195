set -x
for k in 144 432 512; do for ts in 64 128; do for bl in 1 4; do timeout -k 5 60 ./scripts/mt_bench 8192 $k $bl $ts || exit 1; done; done; done
timeout -k 5 60 ./scripts/mt_bench 8000 150 1 128
timeout -k 5 60 ./scripts/mt_bench 8000 150 1 64
