TRACE_JOBS=96:140 timeout -k 5 120 python scripts/prog_trace.py > gpurun_out/trace_i.txt 2>&1
