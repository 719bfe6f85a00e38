"""Race hunting: replay the captured train step WITHOUT per-step synchronisation in bursts, check the loss scalars after
every burst, and when a NaN appears report which tensors hold non-finite values."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ste_gan_amd import ops
from ste_gan_amd.constants import DataType
from ste_gan_amd.train import GraphedTrainStep

ops.set_compute_dtype(os.environ.get("PROBE_DTYPE", "bf16"))
dev = torch.device("cuda", 0)
if os.environ.get("PROBE_GROUP"):            # 1-rank RCCL group: GraphedTrainStep replays three graphs around the exchanges
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29577", HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
cfg = bench.make_cfg("train_step")
if os.environ.get("PROBE_NO_TD"):
    cfg.train.loss_multi_td_error = False
if os.environ.get("PROBE_NO_ENC"):
    cfg.train.loss_speech_unit_error = False
    cfg.train.loss_phoneme_error = False
eager = bench.build_step(cfg, dev)
batch = bench.synthetic_batch(dev, seed=1000)
step = eager if os.environ.get("PROBE_EAGER") else GraphedTrainStep(eager, batch, warmup=2)
if os.environ.get("PROBE_EAGER"):
    class _W:
        def __init__(self, s): self.s = s; self.last = None; self.static = batch
        def __call__(self, b): self.last = self.s(b); return self.last
        read_stats = staticmethod(type(eager).read_stats)
    step = _W(eager)
bursts, per = int(os.environ.get("PROBE_BURSTS", "60")), int(os.environ.get("PROBE_PER", "10"))
for b in range(bursts):
    for _ in range(per):
        step(batch)
    torch.cuda.synchronize()
    st = step.read_stats(step.last)
    bad = [k for k, v in st.items() if isinstance(v, float) and not math.isfinite(v)]
    if os.environ.get("PROBE_TRACE"):
        print(b * per, {k: round(v, 3) for k, v in st.items() if isinstance(v, float) and k in ("generator", "discriminator", "feature_matching", "multi_td")}, flush=True)
    if bad:
        nf = lambda t: int((~torch.isfinite(t)).sum())
        print(f"burst {b}: non-finite {bad}")
        print("  static x_t", nf(step.static[DataType.REAL_EMG]), "units", nf(step.static[DataType.SPEECH_UNITS].float()))
        print("  x_pred", nf(step.last["x_pred"]))
        print("  G params", sum(nf(p) for p in eager.netG.parameters()), "D params", sum(nf(p) for p in eager.netD.parameters()))
        print("  G grads", sum(nf(p.grad) for p in eager.netG.parameters()), "D grads", sum(nf(p.grad) for p in eager.netD.parameters()))
        print("  stats", {k: v for k, v in st.items() if isinstance(v, float)})
        sys.stdout.flush()
        os._exit(1)
print("no NaN in", bursts * per, "steps", flush=True)
os._exit(0)
