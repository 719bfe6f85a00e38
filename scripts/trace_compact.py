"""rocprofv3 kernel-trace CSV -> compact per-kernel rows (step, queue, start us, end us, grid, short name) of the last N steps.
    python scripts/trace_compact.py TRACE.csv N > out.tsv"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2
idx = [i for i, r in enumerate(rows) if "adamw_kernel" in r["Kernel_Name"]]
first = idx[-2 * n - 1] + 1
t0 = int(rows[first]["Start_Timestamp"])
step = 0
for i in range(first, idx[-1] + 1):
    r = rows[i]
    name = re.sub(r"\(anonymous namespace\)::|void |at::native::", "", r["Kernel_Name"])[:60]
    grid = r.get("Grid_Size", r.get("Grid_Size_X", "?"))
    wg = r.get("Workgroup_Size", r.get("Workgroup_Size_X", "?"))
    print(f"{step}\tq{r['Queue_Id']}\t{(int(r['Start_Timestamp']) - t0) / 1e3:.1f}\t{(int(r['End_Timestamp']) - t0) / 1e3:.1f}\t"
          f"{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:.1f}\t{grid}/{wg}\t{r.get('LDS_Block_Size', '?')}\t{name}")
    if i in idx and (idx.index(i) - (len(idx) - 2 * n - 1)) % 2 == 0:
        step += 1
