mkdir -p gpurun_out
python -m pytest tests/test_gpu_conv_bf16.py -x -q -m gpu -k bwd_weight > gpurun_out/wg3_tests.log 2>&1; tail -2 gpurun_out/wg3_tests.log
B="python bench.py --steps 30 --warmup 3 --no-cpu-baseline --profile-steps 0 --no-f32"
$B > gpurun_out/wg3_a1.json 2>gpurun_out/wg3.err && STEGAN_BF16_WGRAD_NO_HALF=1 $B > gpurun_out/wg3_b1.json 2>>gpurun_out/wg3.err && $B > gpurun_out/wg3_a2.json 2>>gpurun_out/wg3.err && STEGAN_BF16_WGRAD_NO_HALF=1 $B > gpurun_out/wg3_b2.json 2>>gpurun_out/wg3.err
python - <<PY
import json
for f in ["wg3_a1","wg3_b1","wg3_a2","wg3_b2"]:
    try:
        d=json.loads(open("gpurun_out/%s.json"%f).read().strip().splitlines()[-1]); print(f, d["ms_per_step"], d["value"])
    except Exception as e: print(f, "ERR", e)
PY
