"""Summarise two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) into HBM bytes per launch per kernel family.
usage: pmc_traffic.py <fetch_dir> <write_dir> <out.json>
Counters are in KiB; FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950 (128-B requests counted as 64 B)."""
import collections, csv, glob, json, sys


def per_kernel(d, counter):
    f = glob.glob(d + "/*/*counter_collection.csv")[0]
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        n = r["Kernel_Name"]
        fam = ("conv_bf16_kernel" if ("conv_bf16_kernel" in n or "conv_bf16_multi_kernel" in n) else
               "conv_wgrad_bf16_kernel" if ("conv_wgrad_bf16_kernel" in n or "conv_wgrad_bf16_one_kernel" in n) else
               "conv_direct_kernel" if "conv_direct_kernel" in n else "conv_wgrad_kernel" if "conv_wgrad_kernel" in n else
               "conv_igemm_kernel<2>" if "conv_igemm_kernel<2" in n else "conv_splitk_epilogue_kernel" if "splitk_epilogue" in n
               else "conv_cout1_kernels" if "conv_cout1" in n else "assemble_batch_kernel" if "assemble_batch" in n else None)
        if fam is None:
            continue
        agg[fam][0] += 1
        agg[fam][1] += float(r["Counter_Value"])
    return agg


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes, --kernel-trace only) around "
                 "`python3 bench.py --steps 1 --warmup 1 --profile-steps 0 --no-cpu-baseline --no-f32 --no-graph`; FETCH_SIZE doubled per "
                 "MI355X_MICROARCH.md (gfx950 counts 128-B requests as 64 B for wide coalesced reads), counters are in KiB; "
                 "summarised by scripts/pmc_traffic.py"}
for fam in fetch:
    n = fetch[fam][0]
    fk, wk = fetch[fam][1] / n, write[fam][1] / max(1, write[fam][0])
    out[fam] = {"launches": n, "fetch_kib_raw_per_launch": round(fk, 1), "write_kib_per_launch": round(wk, 1),
                "hbm_bytes_per_launch": int(round((2 * fk + wk) * 1024, -4))}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
