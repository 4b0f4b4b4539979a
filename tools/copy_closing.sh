#!/bin/bash
# tools/copy_closing.sh <tag>: what tools/closing_run.sh left under gpurun_out/ into profiles/ (the tracked copies the design cites)
TAG=${1:-r04}
cd "$(dirname "$0")/.."
cp gpurun_out/prof_$TAG/summary.txt profiles/${TAG}_bench_longbackref_256x4MiB.txt
cp gpurun_out/prof_${TAG}_highentropy/summary.txt profiles/${TAG}_bench_highentropy_256x4MiB.txt
cp gpurun_out/prof_${TAG}_alice29x1024/summary.txt profiles/${TAG}_bench_alice29x1024.txt
cp gpurun_out/prof_$TAG/pmc.json profiles/pmc_$TAG.json
cp gpurun_out/prof_${TAG}_highentropy/pmc.json profiles/pmc_${TAG}_highentropy.json
cp gpurun_out/prof_${TAG}_alice29x1024/pmc.json profiles/pmc_${TAG}_alice29x1024.json
tail -1 gpurun_out/closing_$TAG/bench_default.json > profiles/${TAG}_bench_default.json
{
  echo "# -DBROTLI_AMD_PROFILE_SCAN build of HEAD (round ${TAG#r}), bench.py --workload longbackref_256x4MiB --steps 2: block 0's stream, wave 0's clock per region"
  grep -v '^{' gpurun_out/closing_$TAG/phases_longbackref_256x4MiB.txt | grep -v amdgpu.ids
  echo
  echo "# the same build, --workload highentropy_256x4MiB (regions of long literal runs)"
  grep -v '^{' gpurun_out/closing_$TAG/phases_highentropy_256x4MiB.txt | grep -v amdgpu.ids
  echo
  echo "# -DBROTLI_AMD_PROFILE_WAVES build, longbackref_256x4MiB: every wave's ticks in front of each barrier of the engine (numbered in source order)"
  grep -v '^{' gpurun_out/closing_$TAG/waves_longbackref_256x4MiB.txt | grep -v amdgpu.ids
} > profiles/${TAG}_path_engine_phases.txt
ls -la profiles | grep $TAG
