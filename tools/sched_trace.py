"""Occupancy of the lanes over a run, from the lines ChainScheduler writes under ZE_SCHED_TRACE=<file> (one per burst: scheduler id,
host time, B, live chains, steps run, requests waiting, passes of the round not yet enqueued, prefilled requests not yet joined; one per prefill pass: id, time, P, chains, rows).
usage: ZE_SCHED_TRACE=/tmp/t.txt python bench.py ... ; python tools/sched_trace.py /tmp/t.txt [bucket seconds]"""
import sys
from collections import defaultdict

rows = [l.split() for l in open(sys.argv[1]) if l.strip()]
dt = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
lanes = sorted({r[0] for r in rows})
# the trace holds every run of the process (warm-up steps included): keep the last run of each lane = after its last long gap
by = defaultdict(list)
passes = defaultdict(list)
for r in rows:
    if r[2] == "G":
        print("scheduler", r[0], "pass enqueued at host +%.3f s; on the GPU it started at +%.3f s and ended at +%.3f s (since the scheduler was created)" % (float(r[3]), float(r[4]), float(r[5])))
        continue
    if r[2] == "P":
        passes[r[0]].append((float(r[1]), int(r[3]), int(r[4])))
    else:
        by[r[0]].append((float(r[1]), int(r[3]), int(r[4]), int(r[5]), int(r[6]), int(r[7])))
t_end = max(v[-1][0] for v in by.values())
span = float(sys.argv[3]) if len(sys.argv) > 3 else 20.0
t0 = t_end - span
print("lanes:", lanes, "| last %.1f s of the trace, buckets of %.2f s: per lane steps / mean live chains / waiting at the end of the bucket" % (span, dt))
nb = int(span / dt) + 1
for b in range(nb):
    cells = []
    for ln in lanes:
        pts = [p for p in by[ln] if t0 + b * dt <= p[0] < t0 + (b + 1) * dt]
        st = sum(p[2] for p in pts)
        ps = [p for p in passes[ln] if t0 + b * dt <= p[0] < t0 + (b + 1) * dt]
        cells.append("%4d st %5.0f live %4d wait %2d passes %6d rows" % (st, (sum(p[1] * p[2] for p in pts) / st) if st else 0, pts[-1][3] if pts else 0,
                                                                 len(ps), sum(p[2] for p in ps)))
    print("%6.2f s | " % (b * dt) + " | ".join(cells))
