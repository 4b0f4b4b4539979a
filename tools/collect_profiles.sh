#!/bin/bash
# rocprofv3 evidence for bench.py's roofline object, run on the GPU box:
#   tools/collect_profiles.sh <tag> [workload]      ->  gpurun_out/prof_<tag>/{summary.txt, pmc.json}
# Three separate rocprofv3 runs of the same bench command (kernel trace; FETCH_SIZE; WRITE_SIZE), plus one with the
# SQ instruction-mix counters, as MI355X_MICROARCH.md prescribes (PMC passes carry --kernel-trace only).
set -u
TAG=${1:-r01}
WL=${2:-longbackref_256x4MiB}
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
CMD="python $REPO/bench.py --workload $WL --steps 5 --warmup 1 --no-cpu-baseline --no-extra"
run() {  # name, rocprofv3 options...
  local name=$1; shift
  rm -rf "/tmp/rp_$name"
  timeout 600 rocprofv3 "$@" -d "/tmp/rp_$name" -o "$name" -- $CMD > "$OUT/$name.log" 2>&1
  find "/tmp/rp_$name" -name '*.db' | head -1
}
DB_T=$(run trace --kernel-trace --stats)
DB_F=$(run fetch --kernel-trace --pmc FETCH_SIZE)
DB_W=$(run write --kernel-trace --pmc WRITE_SIZE)
DB_S=$(run sq --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_ANY)
DB_M=$(run vmem --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_WAVES SQ_BUSY_CYCLES)
python "$REPO/tools/rocpd_summary.py" $DB_T $DB_F $DB_W $DB_S $DB_M > "$OUT/summary.txt" 2>&1
python - "$WL" "$DB_T" "$DB_F" "$DB_W" > "$OUT/pmc.json" <<'EOF'
import json, sqlite3, sys
wl, dbt, dbf, dbw = sys.argv[1:5]
# (the decode launches only: a batch of more streams than CUs is preceded by a probe launch of the same kernel that decodes nothing --
# a few per cent of a decode launch's time and traffic; anything below half of the largest value is not a decode launch)
def avg(db, counter):
    try:
        v = [r[0] for r in sqlite3.connect(db).execute("select value from counters_collection where counter_name=? and kernel_name like 'brotli_amd_decode%kernel%'", (counter,))]
        v = [x for x in v if x >= 0.5 * max(v)]
        return sum(v) / len(v)
    except Exception:
        return None
def kavg(db):
    v = [r[0] for r in sqlite3.connect(db).execute("select duration from kernels where name like 'brotli_amd_decode%kernel%'")]
    v = [x for x in v if x >= 0.5 * max(v)]
    return (sum(v) / len(v), len(v))
f, w = avg(dbf, "FETCH_SIZE"), avg(dbw, "WRITE_SIZE")
k = kavg(dbt)
kname = [r[0] for r in sqlite3.connect(dbt).execute("select distinct name from kernels where name like 'brotli_amd_decode%kernel%'")]
out = {"workload": wl, "kernel": kname[0].split("(")[0] if kname else "brotli_amd_decode_kernel",   # (brotli_amd_decode_gang_kernel: the launches that give every stream a gang of blocks)
       "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of bench.py, tools/collect_profiles.sh",
       "kernel_avg_ns": k[0], "kernel_launches": k[1], "fetch_size_kb_raw": f, "write_size_kb": w,
       "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B read requests at 64 B); byte and 16-byte accesses of this kernel uncalibrated",
       "traffic_bytes_per_launch": None if f is None or w is None else int(2 * f * 1024 + w * 1024)}
print(json.dumps(out, indent=1))
EOF
tail -3 "$OUT/trace.log"
cat "$OUT/pmc.json"
