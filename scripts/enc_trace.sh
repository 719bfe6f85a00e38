# kernel sequence of one replay of the side branch (scripts/encoder_branch.py) from a rocprofv3 kernel trace
mkdir -p gpurun_out/prof_enc
cd /tmp && export TMPDIR=/tmp ENC_ONLY=1
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_enc -- python3 $GRAFT_REPO_ROOT/scripts/encoder_branch.py > $GRAFT_REPO_ROOT/gpurun_out/prof_enc.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<PY
import csv,glob,re,collections
f=glob.glob("gpurun_out/prof_enc/*/*kernel_trace.csv")[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r["Start_Timestamp"]))
names=[re.sub(r"\(anonymous namespace\)::|void ","",r["Kernel_Name"])[:60] for r in rows]
per=None
for n in range(50,1200):
    if names[-n:]==names[-2*n:-n]: per=n; break
seg=rows[-per:]
t0=int(seg[0]["Start_Timestamp"])
out=open("gpurun_out/enc_seq.txt","w")
agg=collections.defaultdict(lambda:[0,0.0])
for r in seg:
    st,en=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    n=re.sub(r"\(anonymous namespace\)::|void ","",r["Kernel_Name"])[:70]
    out.write("%8.1f %6.1f  %s grid=%s\n"%((st-t0)/1e3,(en-st)/1e3,n, r.get("Grid_Size_X","")))
    k=n.split("(")[0][:50]; agg[k][0]+=1; agg[k][1]+=(en-st)/1e3
print("kernels per replay", per, "span us", (int(seg[-1]["End_Timestamp"])-t0)/1e3)
for k,(c,t) in sorted(agg.items(), key=lambda kv:-kv[1][1]): print("%-52s %4d %8.1f us  avg %5.1f"%(k,c,t,t/c))
PY
rm -rf gpurun_out/prof_enc
