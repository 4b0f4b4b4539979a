#!/bin/bash
# tools/copy_gang.sh <tag>: what tools/gang_run.sh left under gpurun_out/ into profiles/ (the tracked copies the design cites)
TAG=${1:-r05}
cd "$(dirname "$0")/.."
for WL in longbackref_1x1024MiB longbackref_1x64MiB longbackref_32x4MiB longbackrefmix_200; do
  cp gpurun_out/prof_${TAG}_$WL/summary.txt profiles/${TAG}_bench_$WL.txt
  cp gpurun_out/prof_${TAG}_$WL/pmc.json profiles/pmc_${TAG}_$WL.json
done
{
  echo "# tools/gang_run.sh $TAG: gangs of blocks on one stream (DESIGN 2e) against one block a stream, the same box, bench.py --workload <id> --steps 3 --warmup 1;"
  echo "# every run checks every output by SHA-256 before and after its timed steps"
  cat gpurun_out/gang_$TAG/gang_ab.txt
} > profiles/${TAG}_gang_ab.txt
{
  echo "# tools/gang_run.sh $TAG: a -DBROTLI_AMD_GANG_STATS build (tools/build_variant.sh gangstats -DBROTLI_AMD_GANG_STATS), BROTLI_AMD_GANG_STATS=1: the first stream's gang,"
  echo "# counters summed over its blocks; ticks are s_memtime's (the shader clock)"
  cut -c1-4000 gpurun_out/gang_$TAG/gang_stats.txt
} > profiles/${TAG}_gang_stats.txt
ls -la profiles | grep -E "gang|1x1024|1x64|32x4"
