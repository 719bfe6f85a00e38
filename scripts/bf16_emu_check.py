"""Measure the bf16 train step against the oracle run with bf16-rounded operands (oracle.bf16_operands): prints the error
statistics from which the bounds of tests/test_gpu_bf16_emulation.py were chosen.  usage: bf16_emu_check.py [small|c3]"""
import copy
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests/golden"))
import cases                                                                                   # noqa: E402
from oracle import stegan_oracle as O                                                          # noqa: E402


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def report(tag, st, out, netG, netD, o, g_keys, d_keys):
    from test_gpu_full_size import _losses
    got = np.array(_losses(st)); want = np.array([o["loss_d"], o["loss_g"], o["loss_td"], o["loss_su"], o["loss_ph"], o["loss_fm"]])
    print(tag, "losses rel", np.abs(got - want) / np.abs(want))
    print(tag, "x_pred max abs", float((out["x_pred"].cpu() - o["x_pred"]).abs().max()))
    for name, net, keys, grads in (("G", netG, g_keys, o["g_grads"]), ("D", netD, d_keys, o["d_grads"])):
        ps = dict(net.named_parameters())
        errs = np.array([rel(ps[k].grad, grads[k]) for k in keys])
        nrm = np.array([abs(float(ps[k].grad.double().norm()) / max(float(grads[k].double().norm()), 1e-30) - 1) for k in keys])
        worst = np.argsort(-errs)[:4]
        print(tag, name, "grad rel L2: max %.3e median %.3e | norm rel: max %.3e" % (errs.max(), np.median(errs), nrm.max()),
              [(keys[i], "%.2e" % errs[i], tuple(grads[keys[i]].shape)) for i in worst])


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "small"
    from ste_gan_amd import ops
    ops.set_compute_dtype("bf16")
    if which == "small":
        from test_gpu_train_step import build, to_batch
        from oracle import shapes as oshapes
        from oracle.detfill import fill_state_dict
        name = next(iter(cases.STEP_CASES))
        ch, (ms, nl, ff), B, T, steps = cases.STEP_CASES[name]
        g = fill_state_dict(oshapes.generator_shapes("SPEECH_UNITS", ch)); d = fill_state_dict(oshapes.discriminator_shapes(True))
        e = fill_state_dict(oshapes.encoder_shapes(ms, 3, nl, dim_feedforward=ff))
        dt = {k: v for k, v in d.items() if not O.is_sn_buffer(d, k)}
        og, od = O.adamw_init(g), O.adamw_init(dt)
        cfg, netG, netD, optG, optD, step = build(name)
        for s in range(steps):
            b = cases.step_batch(name, s)
            with O.bf16_operands():
                o = O.train_step(g, d, e, og, od, b, keep_grads=True)
            out = step(to_batch(b)); st = step.read_stats(out)
            report(f"s{s}", st, out, netG, netD, o, list(g), list(dt))
    else:
        from test_gpu_full_size import _batch, _build, _inputs, _state_dicts
        inp = _inputs("c3", 64, 12, False)
        g, d, e = _state_dicts("SPEECH_UNITS")
        netG, netD, optG, optD, step = _build("SPEECH_UNITS", g, d, e)
        out = step(_batch(inp, False)); st = step.read_stats(out)
        g, d, e = _state_dicts("SPEECH_UNITS")
        dt = {k: v for k, v in d.items() if not O.is_sn_buffer(d, k)}
        import time; t0 = time.time()
        with O.bf16_operands():
            o = O.train_step(g, d, e, O.adamw_init(g), O.adamw_init(dt), inp, keep_grads=True)
        print("oracle (emulated) step: %.1f s" % (time.time() - t0))
        report("c3", st, out, netG, netD, o, list(g), list(dt))


if __name__ == "__main__":
    main()
