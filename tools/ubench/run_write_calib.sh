#!/bin/bash
# tools/ubench/run_write_calib.sh: FETCH_SIZE / WRITE_SIZE per kernel of write_calib against the 256 MiB each of them moves
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
export TMPDIR=/tmp
cd /tmp
for C in WRITE_SIZE FETCH_SIZE; do
  rm -rf /tmp/wc_$C
  timeout 300 rocprofv3 --kernel-trace --pmc $C -d /tmp/wc_$C -o wc -- $REPO/tools/ubench/bin/write_calib > /tmp/wc_$C.log 2>&1
  DB=$(find /tmp/wc_$C -name '*.db' | head -1)
  python - "$DB" $C <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
t = [x for x in tabs if x.startswith('counters_collection')][0]
cols = [r[1] for r in db.execute("pragma table_info(%s)" % t)]
print(sys.argv[2], "(KB) per dispatch, / 262144 KB:")
for row in db.execute("select dispatch_id, kernel_name, value from %s where counter_name=? order by dispatch_id" % t, (sys.argv[2],)):
    print("  %3d %-40s %12.0f  %.3f" % (row[0], row[1][:40], row[2], row[2] / 262144.0))
PY
done
