#!/bin/bash
# tools/abc.sh <workload> <steps> <lib>...: alternating runs of several builds on the same GPU box
WL=$1; STEPS=$2; shift 2
for round in 1 2; do
  for L in "$@"; do
    BROTLI_AMD_LIB=$L timeout 300 python bench.py --workload $WL --steps $STEPS --warmup 1 --no-cpu-baseline --no-extra 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$L', d['value'], d['ms_per_step'])"
  done
done
