timeout 300 python tools/prof_workload.py high_entropy 64 2>&1 | tail -8
