set -x
CHOLAMD_TRACE_FOLLOW=gpurun_out/follow_g.txt timeout -k 5 120 python scripts/prog_trace.py > gpurun_out/trace_g.txt 2>&1
