#!/bin/bash
# Round-5 evidence, run on the GPU box from the repository root:  bash scripts/r05_evidence.sh pmc | bench
# pmc:   three rocprofv3 --pmc passes over the graph-replaying bench + two over one instrumented eager step (per layer shape)
# bench: the default bench line, the kernel-trace stats of the same command, the kernel mix of one replayed step
# Results land in gpurun_out/r05/evidence/; the files to be judged are then copied into profiles/ (names r05_*).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05/evidence
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 3 --warmup 1 --profile-steps 0 --no-cpu-baseline --no-f32"
if [ "$1" = "pmc" ]; then
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/a -- $B > $O/a.log 2>&1 || exit 1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/b -- $B > $O/b.log 2>&1 || exit 1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/c -- $B > $O/c.log 2>&1 || exit 1
  (cd $R && python3 scripts/pmc_summary.py $O/a $O/b $O/c 3 $O/r05_pmc_summary.json $O/r05_pmc_raw > $O/summary.log) || exit 1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/sb -- python3 $R/scripts/pmc_by_shape.py run $O/records.json > $O/sb.log 2>&1 || exit 1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/sc -- python3 $R/scripts/pmc_by_shape.py run $O/records_c.json > $O/sc.log 2>&1 || exit 1
  (cd $R && python3 scripts/pmc_by_shape.py join $O/records.json $O/sb $O/sc $O/by_shape.json > $O/r05_pmc_by_shape.txt) || exit 1
  rm -rf $O/a $O/b $O/c $O/sb $O/sc
else
  (cd $R && python3 bench.py > $O/r05_bench_bf16.json 2> $O/bench.err) || exit 1
  # kernel stats of the bf16 step alone (graph replays + generator inference; no fp32 pass, no instrumented pass) ...
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/k -- python3 $R/bench.py --no-cpu-baseline --no-f32 --profile-steps 0 > $O/r05_bench_under_rocprof.json 2> $O/k.log || exit 1
  cp $(find $O/k -name "*kernel_stats.csv") $O/r05_train_step_bf16_kernel_stats.csv
  (cd $R && python3 scripts/step_kernels.py $(find $O/k -name "*kernel_trace.csv") 3 > $O/r05_step_kernels.txt) || exit 1
  rm -rf $O/k
  # ... and of the default command as the driver runs it (bf16 replays, instrumented pass, fp32 parity pass)
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/k2 -- python3 $R/bench.py --no-cpu-baseline > $O/r05_bench_default_under_rocprof.json 2> $O/k2.log || exit 1
  cp $(find $O/k2 -name "*kernel_stats.csv") $O/r05_bench_default_kernel_stats.csv
  rm -rf $O/k2
fi
echo done
