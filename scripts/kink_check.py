"""Is a dx mismatch a leaky-ReLU kink flip?  Lists activations whose sign differs between the GPU forward and the oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import stegan_oracle as O, shapes as oshapes
from oracle.detfill import det_input, fill_state_dict
from ste_gan_amd.models.discriminator import DiscriminatorSmall
sd = fill_state_dict(oshapes.discriminator_shapes(True))
d = DiscriminatorSmall(8); d.load_state_dict(sd); d = d.cuda().eval()
xf = det_input("gl2.fake", (3, 96, 8), "tanh")
of = d(xf.cuda())
rf = O.discriminator_forward(sd, xf, True, training=False)
for i, (a, b) in enumerate(zip(of, rf)):
    for j, (fa, fb) in enumerate(zip(a[:-1], b[:-1])):
        fa = fa.detach().cpu()
        flip = (fa * fb) < 0
        rel = (fa - fb).abs().max() / fb.abs().max()
        if flip.any():
            idx = flip.nonzero()[0]
            print(f"disc {i} fmap {j}: {int(flip.sum())} sign flips, e.g. gpu {float(fa[tuple(idx)]):.3e} vs ref {float(fb[tuple(idx)]):.3e}; max|y| {float(fb.abs().max()):.3e}, fwd max rel err {float(rel):.2e}")
        elif rel > 1e-5:
            print(f"disc {i} fmap {j}: fwd max rel err {float(rel):.2e} (no flips)")
print("done")
