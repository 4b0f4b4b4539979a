BROTLI_AMD_LIB=$PWD/tools/scratch/lib_profspec.so timeout 300 python tools/prof_workload.py high_entropy 64 2>&1 | grep "spec rounds" | head -1
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -1
