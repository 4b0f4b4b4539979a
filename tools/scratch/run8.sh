BROTLI_AMD_LIB=$PWD/tools/scratch/lib_proflean.so timeout 300 python tools/prof_workload.py long_backref 8 4 2>&1 | grep "kernel ms\|lean cmds" | head -2
BROTLI_AMD_LIB=$PWD/tools/scratch/lib_proflean.so timeout 300 python tools/prof_workload.py long_backref 8 16 2>&1 | grep "kernel ms\|lean cmds" | head -2
BROTLI_AMD_LIB=$PWD/tools/scratch/lib_prof.so timeout 300 python tools/prof_workload.py long_backref 8 16 2>&1 | grep "kernel ms\|ticks total\|per command\|lean exits" | head -8
