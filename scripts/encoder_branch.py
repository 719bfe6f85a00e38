"""The D-independent side branch of the train step alone (multi-TD loss + EMG-encoder losses and their gradient w.r.t. the
generated EMG): graph-replay time, and -- under `rocprofv3 --kernel-trace` -- its kernel sequence.
    ENC_ONLY=1 drops the multi-TD loss, TD_ONLY=1 the encoder."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ste_gan_amd import ops
from ste_gan_amd.constants import DataType

ops.set_compute_dtype(os.environ.get("PHASE_DTYPE", "bf16"))
dev = torch.device("cuda", 0)
cfg = bench.make_cfg("train_step")
step = bench.build_step(cfg, dev)
batch = bench.synthetic_batch(dev, seed=1000)
t = cfg.train
x_t = batch[DataType.REAL_EMG] if DataType.REAL_EMG in batch else batch[list(batch.keys())[0]]
su, ph = batch[DataType.SPEECH_UNITS], batch[DataType.PHONEMES]
for k, v in batch.items():
    print(k, tuple(v.shape), v.dtype, file=sys.stderr)
x_gen = torch.tanh(torch.randn_like(x_t))


def branch():
    xp = x_gen.detach().requires_grad_(True)
    loss = None
    if not os.environ.get("ENC_ONLY"):
        loss = t.loss_multi_td_weight * step.multi_td_loss(x_t, xp)
    if not os.environ.get("TD_ONLY"):
        eo = step.emg_encoder_loss(xp, su, ph)
        term = t.loss_speech_unit_weight * eo.speech_unit_loss + t.loss_phoneme_weight * eo.phoneme_loss
        loss = term if loss is None else loss + term
    (g,) = torch.autograd.grad(loss, xp)
    return g


s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3):
        branch()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s):
    out = branch()
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 50
for _ in range(n):
    g.replay()
torch.cuda.synchronize()
print("side branch: %.3f ms per replay" % (1e3 * (time.perf_counter() - t0) / n), flush=True)
