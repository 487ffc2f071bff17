import numpy as np, sys
d = np.fromfile(sys.argv[1], dtype=np.uint64)
groups, tpg, lagreq = int(d[0]), int(d[1]), int(d[2])
quads = (groups - lagreq) // 4 if groups > lagreq else 0
lag = groups - 4 * quads
e = d[4:4 + groups * 8].reshape(groups, 8).astype(np.int64)
t = d[4 + groups * 8:].astype(np.int64)
tv = t[t > 0]
t0 = min(tv.min(), e[:, 0][e[:, 0] > 0].min())
us = lambda x: (x - t0) / 100.0
print(f"groups {groups} tpg {tpg} lag {lag}; tiles end: first {us(tv.min()):.1f} last {us(tv.max()):.1f} us")
start, own, res, done, polls = e[:, 0], e[:, 1], e[:, 2], e[:, 3], e[:, 4]
for kind, sl in (("among tiles", slice(0, groups - lag)), ("trailing", slice(groups - lag, groups))):
    if sl.stop <= sl.start: continue
    print(f"{kind}: start {us(start[sl].min()):.1f}..{us(start[sl].max()):.1f}, resolved last {us(res[sl].max()):.1f}, done last {us(done[sl].max()):.1f}, polls mean {polls[sl].mean():.2f} max {polls[sl].max()}")
    for name, a, b in (("start->resolved", start, res), ("resolved->done", res, done), ("start->done", start, done)):
        x = (b[sl] - a[sl]) / 100.0
        print(f"  {name:16s} mean {x.mean():6.2f} p50 {np.median(x):6.2f} p90 {np.percentile(x,90):6.2f} max {x.max():6.2f} us")
sel = list(range(0, groups, max(groups // 8, 1))) + list(range(groups - 5, groups))
for g in sel:
    tg = t[g * tpg:(g + 1) * tpg]; tg = tg[tg > 0]
    print(f"  g {g:5d}: tiles end {us(tg.max()) if len(tg) else -1:6.1f}  start {us(start[g]):6.1f} own {us(own[g]) if own[g] else -1:6.1f} res {us(res[g]):6.1f} done {us(done[g]):6.1f} polls {polls[g]}")
ph = e[:, 5:8].astype(np.float64)
q = slice(0, groups - lag)
if q.stop > 0:
    life = (done[q] - res[q]) / 100.0
    print(f"phase cycles per group (among tiles): mask fetch {ph[q,0].mean():.0f}  transpose+scan {ph[q,1].mean():.0f}  rank loop+flush {ph[q,2].mean():.0f}; resolved->done {life.mean():.1f} us")
