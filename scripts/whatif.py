"""What-if timings of the single-graph train step with loss terms switched off (which arm of the step bounds it?).
    python scripts/whatif.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ste_gan_amd import ops
from ste_gan_amd.train import GraphedTrainStep

ops.set_compute_dtype("bf16")
dev = torch.device("cuda", 0)
VARIANTS = [("full step", {}),
            ("no encoder losses (SU, phoneme)", dict(loss_speech_unit_error=False, loss_phoneme_error=False)),
            ("no multi-TD", dict(loss_multi_td_error=False)),
            ("no side branch at all", dict(loss_speech_unit_error=False, loss_phoneme_error=False, loss_multi_td_error=False)),
            ("no feature matching", dict(loss_feat_match_error=False))]
for name, off in VARIANTS:
    cfg = bench.make_cfg("train_step")
    for k, v in off.items():
        setattr(cfg.train, k, v)
    step = GraphedTrainStep(bench.build_step(cfg, dev), bench.synthetic_batch(dev, 1000), warmup=2)
    batch = bench.synthetic_batch(dev, 1000)
    for _ in range(5):
        step(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        step(batch)
    torch.cuda.synchronize()
    print(f"{name:34s} {(time.perf_counter() - t0) / 200 * 1e3:.3f} ms/step", flush=True)
    step.close()
    del step
    torch.cuda.empty_cache()
