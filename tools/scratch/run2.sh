timeout 300 python tools/prof_workload.py long_backref 256 2>&1 | grep -v "^ticks\|^per command\|^fast\|amdgpu.ids" | tail
