#!/usr/bin/env python3
"""Reads the GT lines a -DBROTLI_AMD_GANG_TRACE=<invocation> build prints (csrc/brotli_path_engine.h: GT(k); the host prints them in BrotliAmdBatchWait):
one line a region of that invocation of the first stream's gang, wave 0's stamp of the 100 MHz clock all CUs share (s_memrealtime) at every hand-over.
python tools/gang_trace.py <bench.py's output>  ->  a line a region, K shader clocks (2.37 GHz) from the first region's arrival:
arr    the stream arrives at the region (the region before's walk has published the entry)
walk   walk done          det    details done (every wave's)
state  the region before's state is here          res    resolve done, this region's state published
exw1   the execute's first wait is over (the output of the regions before the one or two still under way)
ab     wave 0's share of (a) and (b) done          class  the dependent copies' ranges; everybody's (a) and (b) in memory
free   the levels that do not wait for the regions before are done          exw2   the second wait is over (the region before's output)
lag0 / lag  the first / the last level of the copies that lag is done          out    the region's output is complete and published
built  this engine's tables of the region were built
then the numbers of levels without / with lagging and whether some copies went in order, and what the region's arrival, resolve and output
came after the region before's (the three chains)."""
import re
import sys
rows = []
for l in open(sys.argv[1]):
    m = re.match(r'GT (\d+) blk (\d+) m (\d+) ndep (\d+) :(.*)', l)
    if m:
        rows.append((int(m.group(1)), int(m.group(2)), int(m.group(3)), int(m.group(4)), list(map(int, m.group(5).split()))))
n = max(r[0] for r in rows) + 1
rows = rows[-n:]   # (the last launch's)
base = min(r[4][0] for r in rows)
f = 23.7 / 1000    # 10 ns -> K clocks at 2.37 GHz
order = [0, 1, 2, 3, 4, 5, 8, 10, 11, 9, 12, 14, 6, 7]
names = ['arr', 'walk', 'det', 'state', 'res', 'exw1', 'ab', 'class', 'free', 'exw2', 'lag0', 'lag', 'out', 'built']
print('region block commands dependent | ' + ' '.join(names) + ' | levels free/lagging, in order | arrival, resolve, output behind the region before\'s | output behind the second wait')
prev = None
for k, blk, m, nd, t in rows:
    r = lambda i: round((t[i] - base) * f, 1) if t[i] else 0
    vals = ' '.join('%s=%s' % (a, r(i)) for a, i in zip(names, order))
    d = lambda i: round((t[i] - prev[i]) * f, 1) if prev else 0
    w = t[13] & 0xffffffff
    print(k, blk, m, nd, '|', vals, '|', w & 255, (w >> 8) & 255, (w >> 16) & 1, '|', d(0), d(4), d(6), '|', round((t[6] - t[9]) * f, 1) if t[9] else 0)
    prev = t
