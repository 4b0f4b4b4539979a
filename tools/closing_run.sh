#!/bin/bash
# The round's closing measurements on the GPU box: tools/closing_run.sh <tag>  ->  gpurun_out/closing_<tag>/...
# (the three workloads' rocprofv3 summaries and PMC passes, the per-phase table of the path engine, the default bench line)
TAG=${1:-r04}
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$REPO"
O=gpurun_out/closing_$TAG
mkdir -p $O
timeout 900 tools/collect_profiles.sh ${TAG} longbackref_256x4MiB > $O/collect_c3.log 2>&1
timeout 900 tools/collect_profiles.sh ${TAG}_highentropy highentropy_256x4MiB > $O/collect_c4.log 2>&1
timeout 900 tools/collect_profiles.sh ${TAG}_alice29x1024 alice29x1024 > $O/collect_c2.log 2>&1
cd "$REPO"
for WL in longbackref_256x4MiB highentropy_256x4MiB alice29x1024; do
  timeout 300 python bench.py --workload $WL --steps 10 --warmup 2 --no-extra > $O/bench_$WL.txt 2>&1
done
if [ -f tools/scratch/lib_scanprof.so ]; then
  for WL in longbackref_256x4MiB highentropy_256x4MiB; do
    BROTLI_AMD_LIB=tools/scratch/lib_scanprof.so timeout 300 python bench.py --workload $WL --steps 2 --warmup 1 --no-cpu-baseline --no-extra > $O/phases_$WL.txt 2>&1
  done
fi
if [ -f tools/scratch/lib_waveprof.so ]; then
  BROTLI_AMD_LIB=tools/scratch/lib_waveprof.so timeout 300 python bench.py --workload longbackref_256x4MiB --steps 2 --warmup 1 --no-cpu-baseline --no-extra > $O/waves_longbackref_256x4MiB.txt 2>&1
fi
( time timeout 900 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err
tail -1 $O/bench_default.json | cut -c1-600
tail -4 $O/bench_default.err
