"""Kernels of one replayed step between two named kernels of a rocprofv3 --kernel-trace csv (queue, start, duration):
    step_region.py <kernel_trace.csv> [first substring] [last substring]   -- default: the generator's backward pass
(between the last pair_loss / scalar_combine launch and the wn_backward launch of the last replay)."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ad = [i for i, r in enumerate(rows) if "adamw_kernel" in r["Kernel_Name"]]
end = ad[-1]; start = ad[-3] + 1
step = rows[start:end + 1]
t0 = int(step[0]["Start_Timestamp"])
for r in step:
    k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).replace("void ", "")
    k = re.sub(r"\(.*", "", k)[:60]
    s = (int(r["Start_Timestamp"]) - t0) / 1e3; d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print(f"{s:9.1f} {d:7.1f} q{r.get('Queue_Id','?'):>3s} g{r.get('Grid_Size','?'):>8s} wg{r.get('Workgroup_Size','?'):>5s} lds{r.get('LDS_Block_Size','?'):>7s} {k}")
