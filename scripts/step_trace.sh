#!/bin/bash
# Kernel trace of the graph-replayed bf16 train step + its per-queue timeline (scripts/trace_timeline.py) and kernel mix
# (scripts/step_kernels.py).  Run on the GPU box from the repository root:  bash scripts/step_trace.sh OUTDIR [extra bench args]
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/${1:-gpurun_out/trace}
shift
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/k -- python3 $R/bench.py --steps 6 --warmup 2 --profile-steps 0 --no-cpu-baseline --no-f32 "$@" > $O/bench.json 2> $O/k.log || exit 1
T=$(find $O/k -name "*kernel_trace.csv")
(cd $R && python3 scripts/trace_timeline.py $T --list 0 99999 > $O/timeline.txt && python3 scripts/step_kernels.py $T 3 > $O/step_kernels.txt && python3 scripts/trace_compact.py $T 2 > $O/compact.tsv) || exit 1
python3 - "$T" "$O/trace_all.tsv.gz" <<'PY'
import csv, gzip, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
with gzip.open(sys.argv[2], "wt") as f:
    for r in rows:
        n = re.sub(r"\(anonymous namespace\)::|void |at::native::", "", r["Kernel_Name"])[:70]
        f.write(f"q{r['Queue_Id']}\t{(int(r['Start_Timestamp']) - t0) / 1e3:.1f}\t{(int(r['End_Timestamp']) - t0) / 1e3:.1f}\t{r.get('Grid_Size', '?')}\t{n}\n")
PY
rm -rf $O/k
echo done
