BROTLI_AMD_LIB=$PWD/tools/scratch/lib_prof.so timeout 300 python tools/prof_workload.py high_entropy 8 4 2>&1 | grep "kernel ms\|ticks total\|per command"
BROTLI_AMD_LIB=$PWD/tools/scratch/lib_prof.so timeout 300 python tools/prof_workload.py high_entropy 8 16 2>&1 | grep "kernel ms\|ticks total\|per command"
