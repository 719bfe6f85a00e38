"""Per-replay kernel mix of the captured train step: difference of two `rocprofv3 --kernel-trace --stats` runs of bench.py
with different step counts (warm-up, capture and one-time allocations cancel).
    python scripts/per_replay_counts.py statsA.csv stepsA statsB.csv stepsB"""
import csv, re, sys
a, na, b, nb = sys.argv[1], int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
def load(p):
    return {r["Name"]: (int(r["Calls"]), float(r["TotalDurationNs"])) for r in csv.DictReader(open(p))}
A, B = load(a), load(b)
rows = []
for k in set(A) | set(B):
    ca, ta = A.get(k, (0, 0.0)); cb, tb = B.get(k, (0, 0.0))
    n = (cb - ca) / (nb - na); t = (tb - ta) / (nb - na) / 1e3
    if abs(n) > 1e-9:
        rows.append((t, n, re.sub(r"\(anonymous namespace\)::", "", k)[:110]))
rows.sort(reverse=True)
print(f"# kernels per replay: {sum(r[1] for r in rows):.1f}   kernel time per replay: {sum(r[0] for r in rows) / 1e3:.3f} ms")
for t, n, k in rows:
    print(f"{t:9.1f} us {n:7.1f} x {t / n if n else 0:7.1f} us  {k}")
