# coding: utf-8
"""Round 5: where a train step's wall time goes, from a rocprofv3 kernel trace (scripts/r5_timeline.sh): per step -- the
stretch one queue runs alone (forward, losses, optimiser), the stretch two queues overlap (backward), each queue's busy
time and idle gaps inside backward, the kernels on the critical (busier) queue.  Usage: r5_timeline.py <kernel_trace.csv>"""
import csv
import sys
from collections import defaultdict


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n[:70]


rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), short(r["Kernel_Name"])))
rows.sort()
# steps: split at clip_adam launches (one per step)
ends = [i for i, r in enumerate(rows) if r[3].startswith("clip_adam")]
if len(ends) < 3:
    print("fewer than 3 steps in the trace")
    sys.exit(0)
steps = [(ends[i] + 1, ends[i + 1] + 1) for i in range(len(ends) - 1)]
lo, hi = steps[len(steps) // 2]          # a step from the middle of the run
S = rows[lo:hi]
t0, t1 = S[0][0], max(r[1] for r in S)
print("step: %d launches, %.3f ms wall (first start -> last end)" % (len(S), (t1 - t0) / 1e6))
byq = defaultdict(list)
for r in S:
    byq[r[2]].append(r)
for q, L in sorted(byq.items()):
    busy = sum(e - s for s, e, _, _ in L)
    print("  queue %d: %4d launches, busy %.3f ms, span %.3f .. %.3f ms" % (q, len(L), busy / 1e6, (L[0][0] - t0) / 1e6, (max(e for _, e, _, _ in L) - t0) / 1e6))
# overlap window = span of the second-busiest queue
qs = sorted(byq, key=lambda q: -sum(e - s for s, e, _, _ in byq[q]))
if len(qs) >= 2:
    side = byq[qs[1]]
    w0, w1 = side[0][0], max(e for _, e, _, _ in side)
    print("backward window (span of the second queue): %.3f .. %.3f ms = %.3f ms" % ((w0 - t0) / 1e6, (w1 - t0) / 1e6, (w1 - w0) / 1e6))
    for q in qs[:2]:
        L = [r for r in byq[q] if r[1] > w0 and r[0] < w1]
        busy = sum(min(e, w1) - max(s, w0) for s, e, _, _ in L)
        print("  queue %d inside it: busy %.3f ms (%.0f %%), %d launches" % (q, busy / 1e6, 100.0 * busy / (w1 - w0), len(L)))
        agg = defaultdict(lambda: [0, 0])
        for s, e, _, n in L:
            agg[n][0] += e - s
            agg[n][1] += 1
        for n, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:8]:
            print("      %-70s %4d  %.3f ms" % (n, c, t / 1e6))
    main = byq[qs[0]]
    pre = [r for r in main if r[1] <= w0]
    post = [r for r in main if r[0] >= w1]
    print("before the window (forward + losses): %.3f ms wall, busy %.3f ms, %d launches" % ((w0 - t0) / 1e6, sum(e - s for s, e, _, _ in pre) / 1e6, len(pre)))
    agg = defaultdict(lambda: [0, 0])
    for s, e, _, n in pre:
        agg[n][0] += e - s
        agg[n][1] += 1
    for n, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:8]:
        print("      %-70s %4d  %.3f ms" % (n, c, t / 1e6))
    print("after the window (join, clip + Adam): %.3f ms wall, busy %.3f ms" % ((t1 - w1) / 1e6, sum(e - s for s, e, _, _ in post) / 1e6))
    # gaps on the main queue
    gaps = sorted(((main[i + 1][0] - main[i][1]), main[i][3], main[i + 1][3]) for i in range(len(main) - 1))
    tot = sum(g for g, _, _ in gaps if g > 0)
    print("idle gaps between consecutive launches on the main queue: total %.3f ms; largest:" % (tot / 1e6))
    for g, a, b in gaps[-6:][::-1]:
        print("      %.1f us between %s -> %s" % (g / 1e3, a[:40], b[:40]))
    if len(sys.argv) > 2 and sys.argv[2] == "dumpall":  # ... and of the stretch before it
        print("launches before the window (us from the step's start | length | kernel):")
        for s, e, q, n in S:
            if e <= w0:
                print("  %9.1f %7.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, n[:90]))
    if len(sys.argv) > 2 and sys.argv[2] in ("dump", "dumpall"):     # every launch of the overlap window: start us, length us, queue, kernel
        print("launches of the step from the window's start (us from the step's start | length | queue | kernel):")
        for s, e, q, n in S:
            if e > w0:
                print("  %9.1f %7.1f  q%d  %s" % ((s - t0) / 1e3, (e - s) / 1e3, qs.index(q) + 1 if q in qs else q, n[:60]))
