#!/bin/bash
# One rocprofv3 PMC pass (instruction mix) + kernel time for each library variant given (run on the GPU box):
#   tools/pmc_variants.sh <workload> <variant name>...      (variants: tools/scratch/lib_<name>.so, see tools/build_variant.sh)
set -u
WL=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export TMPDIR=/tmp
cd /tmp
for v in "$@"; do
  rm -rf /tmp/pv_$v
  BROTLI_AMD_LIB=$REPO/tools/scratch/lib_$v.so timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY -d /tmp/pv_$v -o pv -- python $REPO/bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline --no-extra > /tmp/pv_$v.log 2>&1
  DB=$(find /tmp/pv_$v -name '*.db' | head -1)
  python - "$DB" "$v" <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
t = [x for x in tabs if x.startswith('counters_collection')]
vals = {r[0]: r[1] for r in db.execute("select counter_name, avg(value) from %s where kernel_name like 'brotli_amd_decode_kernel%%' group by counter_name" % t[0])}
k = db.execute("select avg(end - start) from kernels where name like 'brotli_amd_decode_kernel%'").fetchone()[0] if 'kernels' in tabs else None
tot = sum(vals.get(c, 0) for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_BRANCH", "SQ_INSTS_SMEM"))
print("%-10s kernel %8.3f ms  VALU %7.1fM SALU %7.1fM LDS %6.1fM BRANCH %6.1fM SMEM %5.1fM  sum %7.1fM  ACTIVE_ANY %7.1fM" % (sys.argv[2], (k or 0) / 1e6, vals.get("SQ_INSTS_VALU", 0) / 1e6, vals.get("SQ_INSTS_SALU", 0) / 1e6,
      vals.get("SQ_INSTS_LDS", 0) / 1e6, vals.get("SQ_INSTS_BRANCH", 0) / 1e6, vals.get("SQ_INSTS_SMEM", 0) / 1e6, tot / 1e6, vals.get("SQ_ACTIVE_INST_ANY", 0) / 1e6))
PY
done
