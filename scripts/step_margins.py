"""How far is the fp32 train step from the golden fixture, quantity by quantity?  (sets the bounds of tests/test_gpu_train_step.py)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np, torch
import cases
import test_gpu_train_step as T

name = "step_c48_b3"
g = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
cfg, netG, netD, optG, optD, step = T.build(name)


def head(sd):
    return np.stack([np.pad(v.flatten()[:4].double().cpu().numpy(), (0, max(0, 4 - v.numel()))) for v in sd])


for s in range(cases.STEP_CASES[name][4]):
    out = step(T.to_batch(cases.step_batch(name, s)))
    t = f"s{s}."
    for label, got, key in (("g_grad_norm", T.norms([p.grad for p in netG.parameters()]), "g_grad_norm"),
                            ("d_grad_norm", T.norms([p.grad for p in netD.parameters()]), "d_grad_norm"),
                            ("g_param_norm", T.norms(netG.state_dict().values()), "g_param_norm"),
                            ("d_param_norm", T.norms(netD.state_dict().values()), "d_param_norm")):
        ref = g[t + key]
        rel = np.abs(got - ref) / np.maximum(np.abs(ref), 1e-3 * np.abs(ref).max())
        print(t + label, "max rel", float(rel.max()), "at", int(rel.argmax()), "ref there", float(ref[rel.argmax()]), "max ref", float(ref.max()))
    for label, got, key in (("g_grad_head", head([p.grad for p in netG.parameters()]), "g_grad_head"),
                            ("d_grad_head", head([p.grad for p in netD.parameters()]), "d_grad_head"),
                            ("g_param_head", head(netG.state_dict().values()), "g_param_head"),
                            ("d_param_head", head(netD.state_dict().values()), "d_param_head")):
        ref = g[t + key]
        if ref.shape != got.shape:
            print(t + label, "shape", ref.shape, got.shape); continue
        err = np.abs(got - ref)
        sc = np.abs(ref).max(axis=1, keepdims=True) + 1e-12
        print(t + label, "max abs", float(err.max()), "max rel-to-row-scale", float((err / np.maximum(sc, 1e-6 * np.abs(ref).max())).max()))
