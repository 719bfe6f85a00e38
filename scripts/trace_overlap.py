"""Which queues ran when: from a rocprofv3 --kernel-trace csv, the last full step's kernels per queue (first start, last end,
busy time), and the time two or more queues were busy at once.  usage: trace_overlap.py <kernel_trace.csv> [steps back]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ad = [i for i, r in enumerate(rows) if "adamw_kernel" in r["Kernel_Name"]]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
# a step = from after the G optimiser launch of step n-1 (every second adamw) to the G optimiser launch of step n
end = ad[-1 - 2 * (back - 1)]
start = ad[-3 - 2 * (back - 1)] + 1
step = rows[start:end + 1]
t0 = int(step[0]["Start_Timestamp"])
print(f"step: {len(step)} kernels, {(int(step[-1]['End_Timestamp']) - t0) / 1e3:.1f} us")
byq = collections.defaultdict(list)
for r in step:
    byq[(r.get("Queue_Id"), r.get("Stream_Id", ""))].append(r)
for q, rs in sorted(byq.items(), key=lambda kv: int(kv[1][0]["Start_Timestamp"])):
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs)
    print(f"queue {q}: {len(rs):4d} kernels  first start {(int(rs[0]['Start_Timestamp']) - t0) / 1e3:8.1f} us  last end "
          f"{(max(int(r['End_Timestamp']) for r in rs) - t0) / 1e3:8.1f} us  busy {busy / 1e3:8.1f} us   first: {rs[0]['Kernel_Name'][:60]}")
ev = []
for r in step:
    ev.append((int(r["Start_Timestamp"]), 1)); ev.append((int(r["End_Timestamp"]), -1))
ev.sort()
lvl, last, hist = 0, ev[0][0], collections.Counter()
for t, d in ev:
    hist[lvl] += t - last; last = t; lvl += d
print("time with k kernels in flight (us):", {k: round(v / 1e3, 1) for k, v in sorted(hist.items())})
# idle gaps > 15 us (no kernel in flight), with the kernels on either side
lvl, gaps, prev_end = 0, [], None
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in step)
cur_end, cur_name = iv[0][1], iv[0][2]
for s_, e_, n_ in iv[1:]:
    if s_ > cur_end + 15000:
        gaps.append(((cur_end - t0) / 1e3, (s_ - cur_end) / 1e3, cur_name[:50], n_[:50]))
    if e_ > cur_end:
        cur_end, cur_name = e_, n_
print("idle gaps > 15 us: (at us, length us, kernel before, kernel after)")
for g in gaps:
    print("  %8.1f  %6.1f  %s | %s" % g)
