#!/bin/bash
# A/B of two builds of the library on the same GPU box: tools/ab.sh <workload> <libA> <libB> [steps]
# (alternating runs; prints ms per step of each)
WL=$1; A=$2; B=$3; STEPS=${4:-5}
for round in 1 2; do
  for L in "$A" "$B"; do
    BROTLI_AMD_LIB=$L timeout 300 python bench.py --workload $WL --steps $STEPS --warmup 1 --no-cpu-baseline --no-extra 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$L', d['value'], d['ms_per_step'])"
  done
done
