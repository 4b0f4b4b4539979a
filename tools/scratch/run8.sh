BROTLI_AMD_LIB=$PWD/tools/scratch/lib_proflean.so timeout 300 python tools/prof_fixture.py 2>&1 | grep -v "^ticks\|^per command\|^fast\|amdgpu.ids\|lean exits" | tail -4
BROTLI_AMD_LIB=$PWD/tools/scratch/lib_proflean.so timeout 300 python tools/prof_workload.py long_backref 64 2>&1 | grep "lean cmds" | tail -1
