"""Timeline summary of one train step from a rocprofv3 --kernel-trace CSV: per hardware queue busy time in 0.5 ms
buckets, the share of time with 0 / 1 / 2+ kernels in flight, and (with --list A B, microseconds from the step start) the
run-length-compressed kernel sequence of a window.  This is how the round-2 serialisation of the encoder branch was found.

    python scripts/trace_timeline.py gpurun_out/.../NNN_kernel_trace.csv [--list 6000 9000]
"""
import collections
import csv
import math
import re
import sys


def main():
    path = sys.argv[1]
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if "adamw_kernel" in r["Kernel_Name"]]
    seg = rows[idx[-3] + 1: idx[-1] + 1]          # two optimiser steps (D, G) = one train step
    t0 = int(seg[0]["Start_Timestamp"])
    t1 = max(int(r["End_Timestamp"]) for r in seg)
    print(f"{len(seg)} kernels, {(t1 - t0) / 1e6:.3f} ms")
    ev = []
    for r in seg:
        ev += [(int(r["Start_Timestamp"]), 1), (int(r["End_Timestamp"]), -1)]
    ev.sort()
    cur, last, hist = 0, t0, collections.Counter()
    for t, d in ev:
        hist[min(cur, 3)] += t - last
        last, cur = t, cur + d
    for k in sorted(hist):
        print(f"  {k}{'+' if k == 3 else ' '} kernels in flight: {hist[k] / (t1 - t0) * 100:5.1f} %  {hist[k] / 1e6:.3f} ms")
    byq = collections.defaultdict(list)
    for r in seg:
        byq[r["Queue_Id"]].append(r)
    qs = sorted(byq)
    print("queue busy ms:", {q: round(sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in byq[q]) / 1e6, 2) for q in qs})
    print("t(ms)  busy % per queue", qs)
    for bi in range(math.ceil((t1 - t0) / 5e5)):
        lo, hi = t0 + bi * 5e5, t0 + (bi + 1) * 5e5
        cells = []
        for q in qs:
            s = sum(max(0, min(int(r["End_Timestamp"]), hi) - max(int(r["Start_Timestamp"]), lo)) for r in byq[q])
            cells.append(f"{s / 5e3:5.0f}")
        print(f"{bi * 0.5:5.1f}", " ".join(cells))
    if "--list" in sys.argv:
        a, b = (float(v) for v in sys.argv[sys.argv.index("--list") + 1: sys.argv.index("--list") + 3])
        out = []
        for r in seg:
            n = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"])[:48]
            st, en = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
            if out and out[-1][0] == r["Queue_Id"] and out[-1][1] == n:
                out[-1][3] = en; out[-1][4] += 1
            else:
                out.append([r["Queue_Id"], n, st, en, 1])
        for q, n, st, en, c in out:
            if a <= st <= b:
                print(f"q{q} {st:8.0f}-{en:8.0f} x{c:<3d} {n}")


if __name__ == "__main__":
    main()
