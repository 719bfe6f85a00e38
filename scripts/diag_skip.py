"""What would a replayed step cost without some of its launches?  (diagnostic: the results of such a step are WRONG)

    python scripts/diag_skip.py stegan_conv1d_bwd_weight_hx,stegan_act_bwd ...

The product library has no such switch (round 4 had STEGAN_SKIP / STEGAN_WHATIF inside ste_gan_amd: removed in round 5, bench.py
refuses to run with them set).  This script wraps the loaded ctypes handle FROM OUTSIDE the package: the named C-ABI entry
points become no-ops that report success, then the bench step is captured and timed."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ste_gan_amd import _lib, ops
from ste_gan_amd.train import GraphedTrainStep


class _Skipping:
    def __init__(self, h, names):
        self._h, self._names = h, set(names)

    def __getattr__(self, name):
        if name in self._names:
            return lambda *a, **k: 0
        return getattr(self._h, name)


def main():
    names = [n for a in sys.argv[1:] for n in a.split(",") if n]
    ops.set_compute_dtype(os.environ.get("DIAG_DTYPE", "bf16"))
    handle = _lib.lib()
    for n in names:
        getattr(handle, n)                      # typo guard
    if names:
        _lib._lib = _lib._libs[_lib._operand] = _Skipping(handle, names)
    dev = torch.device("cuda", 0)
    eager = bench.build_step(bench.make_cfg("train_step"), dev)
    batch = bench.synthetic_batch(dev, seed=1000)
    step = GraphedTrainStep(eager, batch, warmup=2)
    step.load_batch(batch)
    for _ in range(5):
        step(None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 100
    for _ in range(n):
        step(None)
    torch.cuda.synchronize()
    print(f"skipped={','.join(names) or '-'}  {1e3 * (time.perf_counter() - t0) / n:.3f} ms/step (results are wrong when anything is skipped)")


if __name__ == "__main__":
    main()
