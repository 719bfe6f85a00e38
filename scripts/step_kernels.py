"""Kernel mix of ONE replayed step from a rocprofv3 --kernel-trace csv: the kernels between two consecutive generator
optimiser launches (every second adamw_kernel), `back` steps from the end.   step_kernels.py <kernel_trace.csv> [back]"""
import collections, csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ad = [i for i, r in enumerate(rows) if "adamw_kernel" in r["Kernel_Name"]]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
end = ad[-1 - 2 * (back - 1)]
start = ad[-3 - 2 * (back - 1)] + 1
step = rows[start:end + 1]
wall = (max(int(r["End_Timestamp"]) for r in step) - int(step[0]["Start_Timestamp"])) / 1e3
agg = collections.defaultdict(lambda: [0, 0.0])
for r in step:
    k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).replace("void ", "")
    k = re.sub(r"\(.*", "", k)[:90]
    agg[k][0] += 1
    agg[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print(f"# one replayed step: {len(step)} kernels, {sum(v[1] for v in agg.values()) / 1e3:.3f} ms of kernel time in {wall / 1e3:.3f} ms")
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{t:9.1f} us {n:5d} x {t / n:7.1f} us  {k}")
