"""Kernel sequence of ONE replay of a periodically replayed graph from a rocprofv3 --kernel-trace csv (the period is found as
the shortest repetition of kernel names at the end of the trace).   branch_kernels.py <kernel_trace.csv>"""
import collections, csv, re, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
names = [re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"]) for r in rows]
names = [re.sub(r"\(.*", "", n)[:64] for n in names]
per = None
for n in range(20, 2000):
    if names[-n:] == names[-2 * n:-n]:
        per = n
        break
if per is None:
    sys.exit("no repetition found")
seg, nm = rows[-per:], names[-per:]
t0 = int(seg[0]["Start_Timestamp"])
agg = collections.defaultdict(lambda: [0, 0.0])
for r, n in zip(seg, nm):
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%8.1f %6.1f  %-64s grid %s" % ((st - t0) / 1e3, (en - st) / 1e3, n, r.get("Grid_Size", r.get("Grid_Size_X", "?"))))
    agg[n][0] += 1
    agg[n][1] += (en - st) / 1e3
print("# %d kernels, span %.1f us, summed %.1f us" % (per, (max(int(r["End_Timestamp"]) for r in seg) - t0) / 1e3, sum(v[1] for v in agg.values())))
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("# %8.1f us %4d x %6.1f  %s" % (t, c, t / c, n))
