#!/usr/bin/env python3
"""Reads the wall-clock stamp dump of ONE ID-output launch (development builds only: -DPQPS_STAMPS, the shim
writes the file named by PQPS_STAMPS_FILE after every launch) and prints where the time of the launch went:
when the scan tiles ended, when the expanders started, settled (counts seen / sums in front known) and finished.

    PQPS_STAMPS_FILE=/tmp/st.bin python scripts/ab_scan.py --rows 100000000 --queries Q_A --reps 1
    python scripts/stamps.py /tmp/st.bin
"""
import sys

import numpy as np

d = np.fromfile(sys.argv[1], dtype=np.uint64)
groups, tpg, lagreq = int(d[0]), int(d[1]), int(d[2])
quads = (groups - lagreq) // 4 if groups > lagreq else 0
lag = groups - 4 * quads
e = d[4:4 + groups * 8].reshape(groups, 8).astype(np.int64)
t = d[4 + groups * 8:].astype(np.int64)
tv = t[t > 0]
t0 = min(tv.min(), e[:, 0][e[:, 0] > 0].min())
if int(d[3]) > 0:                                                 # the launch's first instruction (block 0): times below are from there
    print(f"first tile ended {(tv.min() - int(d[3])) / 100.0:.1f} us after the launch's first instruction")
    t0 = int(d[3])
us = lambda x: (x - t0) / 100.0                                   # wall_clock64: 100 MHz
print(f"groups {groups} tiles/group {tpg} trailing {lag}; tiles end: first {us(tv.min()):.1f}, p50 {us(np.median(tv)):.1f}, "
      f"p99 {us(np.percentile(tv, 99)):.1f}, last {us(tv.max()):.1f} us")
start, own, res, done, polls = e[:, 0], e[:, 1], e[:, 2], e[:, 3], e[:, 4]
for kind, sl in (("among tiles", slice(0, groups - lag)), ("trailing", slice(groups - lag, groups))):
    if sl.stop <= sl.start:
        continue
    print(f"{kind}: start {us(start[sl].min()):.1f}..{us(start[sl].max()):.1f}, settled last {us(res[sl].max()):.1f}, "
          f"done p50 {us(np.median(done[sl])):.1f} p90 {us(np.percentile(done[sl], 90)):.1f} last {us(done[sl].max()):.1f}, "
          f"polls mean {polls[sl].mean():.2f} max {polls[sl].max()}")
    for name, a, b in (("start->settled", start, res), ("settled->done", res, done), ("start->done", start, done)):
        x = (b[sl] - a[sl]) / 100.0
        print(f"  {name:16s} mean {x.mean():6.2f} p50 {np.median(x):6.2f} p90 {np.percentile(x, 90):6.2f} max {x.max():6.2f} us")
    # how long after the group's own last tile
    last_tile = np.array([t[g * tpg:(g + 1) * tpg].max() for g in range(sl.start, sl.stop)])
    x = (res[sl] - last_tile) / 100.0
    print(f"  own tiles end -> settled: mean {x.mean():6.2f} p50 {np.median(x):6.2f} p90 {np.percentile(x, 90):6.2f} max {x.max():6.2f} us")
sel = list(range(0, groups, max(groups // 8, 1))) + list(range(max(groups - 6, 0), groups))
for g in sel:
    tg = t[g * tpg:(g + 1) * tpg]
    tg = tg[tg > 0]
    print(f"  g {g:5d}: tiles end {us(tg.max()) if len(tg) else -1:6.1f}  start {us(start[g]):6.1f} counts seen {us(own[g]) if own[g] else -1:6.1f} "
          f"settled {us(res[g]):6.1f} done {us(done[g]):6.1f} polls {polls[g]}"
          + (f"  [wave 0: words in LDS {us(e[g, 5]):6.1f}," + (f" first lists in {us(e[g, 7]):6.1f}," if e[g, 7] > 0 else "") + f" IDs out {us(e[g, 6]):6.1f}]" if e[g, 5] > 0 else ""))
