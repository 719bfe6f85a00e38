"""GPU time of the three phases of the train step (the three hipGraphs of the torch.distributed replay, 1-rank RCCL group,
no exchange calls): phase_d = G forward + side branch + discriminator step; phase_g = D optimiser + generator-side D passes +
generator backward; phase_opt = G optimiser.  Event pairs around each replay, averaged over 30 steps."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
import bench
from ste_gan_amd import ops
from ste_gan_amd.train import GraphedTrainStep

ops.set_compute_dtype(os.environ.get("PHASE_DTYPE", "bf16"))
dev = torch.device("cuda", 0)
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29556", HSA_ENABLE_IPC_MODE_LEGACY="0")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
cfg = bench.make_cfg("train_step")
if os.environ.get("PHASE_NO_TD"):            # what-if runs: which branch bounds phase_d?
    cfg.train.loss_multi_td_error = False
if os.environ.get("PHASE_NO_ENC"):
    cfg.train.loss_speech_unit_error = False
    cfg.train.loss_phoneme_error = False
if os.environ.get("PHASE_NO_ADV"):
    cfg.train.loss_adversarial = False
eager = bench.build_step(cfg, dev)
batch = bench.synthetic_batch(dev, seed=1000)
step = GraphedTrainStep(eager, batch, warmup=2)
assert step.graphs
for _ in range(3):
    step(batch)
torch.cuda.synchronize()
n = 30
ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(n)]
for i in range(n):
    ev[i][0].record()
    for k, g in enumerate(step.graphs):
        g.replay()
        ev[i][k + 1].record()
torch.cuda.synchronize()
tot = [sum(ev[i][k].elapsed_time(ev[i][k + 1]) for i in range(n)) / n for k in range(3)]
print("phase_d %.3f ms  phase_g %.3f ms  phase_opt %.3f ms  sum %.3f ms" % (tot[0], tot[1], tot[2], sum(tot)), flush=True)
os._exit(0)
