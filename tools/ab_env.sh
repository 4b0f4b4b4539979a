#!/bin/bash
# A/B of engine modes of one build on the same GPU box: tools/ab_env.sh <workload> <steps> <mode>...   (mode: default | scan | split | norec)
WL=$1; STEPS=$2; shift 2
for round in 1 2; do
  for M in "$@"; do
    if [ "$M" = default ]; then unset BROTLI_AMD_ENGINE; else export BROTLI_AMD_ENGINE=$M; fi
    timeout 300 python bench.py --workload $WL --steps $STEPS --warmup 1 --no-cpu-baseline --no-extra 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$M', d['value'], d['ms_per_step'])"
  done
done
