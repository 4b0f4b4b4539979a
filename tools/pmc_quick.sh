#!/bin/bash
# Quick PMC passes over one bench workload (run on the GPU box):  tools/pmc_quick.sh <workload> <counter set>...
# Each further argument is one rocprofv3 pass: a quoted, space-separated list of counters.  Prints per-kernel averages.
set -u
WL=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export TMPDIR=/tmp
cd /tmp
i=0
for set in "$@"; do
  i=$((i+1))
  rm -rf /tmp/pq_$i
  timeout 600 rocprofv3 --kernel-trace --pmc $set -d /tmp/pq_$i -o pq -- python $REPO/bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline --no-extra > /tmp/pq_$i.log 2>&1
  DB=$(find /tmp/pq_$i -name '*.db' | head -1)
  python - "$DB" <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
t = [x for x in tabs if x.startswith('counters_collection')]
for row in db.execute("select counter_name, avg(value), count(*) from %s where kernel_name like 'brotli_amd_decode_kernel%%' group by counter_name" % t[0]):
    print("  %-28s avg %16.0f  (%d dispatches)" % row)
PY
done
