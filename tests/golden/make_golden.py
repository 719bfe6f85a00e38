#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own modules (build container only).

Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
Needs /root/reference (read-only; never present on the GPU box -- tests read only the .npz).
Harness-side shims, no reference file is modified or copied:
  1. a stub ``omegaconf`` module (the reference imports it only for annotations),
  2. ``layer.self_attn.batch_first = False`` on the encoder's custom attention so torch 2.10's
     ``nn.TransformerEncoder.forward`` takes its plain per-layer loop (torch 2.0.1 behaviour).
Weights come from oracle.detfill (pure function of key + shape) pushed in via load_state_dict,
inputs from tests/golden/cases.py, so fixtures hold outputs only.
"""
import os
import sys
import tempfile
import types
import warnings

sys.dont_write_bytecode = True
warnings.filterwarnings("ignore")

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

_stub = types.ModuleType("omegaconf")


class _DictConfig(dict):
    __getattr__ = dict.__getitem__


_stub.DictConfig = _DictConfig
_stub.OmegaConf = _DictConfig
sys.modules["omegaconf"] = _stub
sys.path.insert(0, "/root/reference")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

import ste_gan  # noqa: E402  (reference package)
from ste_gan.losses.emg_encoder_loss import EMGEncoderLoss  # noqa: E402
from ste_gan.losses.time_domain_loss import MultiTimeDomainFeatureLoss  # noqa: E402
from ste_gan.models.discriminator import Discriminator, DiscriminatorSmall  # noqa: E402
from ste_gan.models.emg_encoder import EMGEncoderTransformer  # noqa: E402
from ste_gan.models.generator import EMGGeneratorGanTTS  # noqa: E402

import cases  # noqa: E402
from oracle import shapes as oshapes  # noqa: E402
from oracle.detfill import det_input, fill_state_dict, shapes_of  # noqa: E402

torch.manual_seed(0)
torch.set_num_threads(8)


def load_det(module, expect_shapes):
    got = shapes_of(module.state_dict())
    assert list(got.items()) == list(expect_shapes.items()), "oracle/shapes.py disagrees with the reference state_dict"
    module.load_state_dict(fill_state_dict(got))
    return module


def stats(t):
    t = t.detach().double().flatten()
    return np.array([t.sum(), t.abs().sum(), (t * t).sum()], np.float64)


def head(t, n=8):
    f = t.detach().flatten()[:n].double().numpy()
    return np.pad(f, (0, n - len(f)))


def grad_summary(named):
    return (np.array([float(g.double().norm()) for _, g in named], np.float64),
            np.stack([head(g, 4) for _, g in named]))


def make_encoder(ms, nl, ff):
    # dim_feedforward is hard-wired to 3072 in the reference constructor (emg_encoder.py:62-64);
    # small cases rebuild the layer stack with a smaller FFN through the same layer class.
    enc = EMGEncoderTransformer(8, 256, 48, model_size=ms, num_transformer_layers=nl)
    if ff != 3072:
        from ste_gan.layers.transformer import TransformerEncoderLayer
        layer = TransformerEncoderLayer(d_model=ms, nhead=8, relative_positional=True,
                                        relative_positional_distance=100, dim_feedforward=ff, dropout=0.2)
        enc.transformer = torch.nn.TransformerEncoder(layer, nl)
    for layer in enc.transformer.layers:
        layer.self_attn.batch_first = False          # shim 2
    enc = load_det(enc, oshapes.encoder_shapes(ms, 3, nl, dim_feedforward=ff))
    return enc.eval()


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB  " + " ".join(f"{k}{tuple(np.shape(v))}" for k, v in arrs.items()))


# ----------------------------------------------------------------------------------------------
def gen_cases():
    for name, (ft, ch, B, T) in cases.GEN_CASES.items():
        g = EMGGeneratorGanTTS(ft, 256 if ft == "SPEECH_UNITS" else 25, 17, 8, channels=ch)
        load_det(g, oshapes.generator_shapes(ft, ch))
        feats, sess, spk = cases.gen_inputs(name)
        g.train()
        out = g(feats, sess, spk)
        probe = det_input(name + ".probe", out.shape)
        (out * probe).sum().backward()
        gn, gh = grad_summary([(k, p.grad) for k, p in g.named_parameters()])
        with torch.inference_mode():
            out_gen = g.generate(feats, sess, spk)
        assert torch.equal(out_gen, out.detach())
        save(name, out=out.detach().numpy(), grad_norm=gn, grad_head=gh)


def disc_cases():
    for name, (small, B, T) in cases.DISC_CASES.items():
        d = (DiscriminatorSmall if small else Discriminator)(8)
        load_det(d, oshapes.discriminator_shapes(small))
        x = cases.disc_input(name).requires_grad_(True)
        out1 = d(x)
        loss = 0
        for sc in out1:
            loss = loss + F.mse_loss(sc[-1], torch.ones_like(sc[-1]))
            for fm in sc[:-1]:
                loss = loss + fm.abs().mean()
        loss.backward()
        gn, gh = grad_summary([(k, p.grad) for k, p in d.named_parameters()])
        out2 = d(x.detach())
        sd = d.state_dict()
        sn = {k: sd[k].numpy() for k in sd if k.endswith("weight_u") or (k.endswith("weight_v") and k[:-1] + "u" in sd)}
        save(name,
             logits1=np.concatenate([sc[-1].detach().flatten().numpy() for sc in out1]),
             logits2=np.concatenate([sc[-1].detach().flatten().numpy() for sc in out2]),
             fmap_shapes=np.array([list(fm.shape) + [0] * (4 - fm.dim()) for sc in out1 for fm in sc], np.int64),
             fmap_stats=np.stack([stats(fm) for sc in out1 for fm in sc]),
             fmap_head=np.stack([head(fm) for sc in out1 for fm in sc]),
             loss=np.float64(loss.item()), dx=x.grad.numpy(), grad_norm=gn, grad_head=gh,
             **{"sn." + k: v for k, v in sn.items()})


def td_cases():
    for name, (B, T) in cases.TD_CASES.items():
        m = MultiTimeDomainFeatureLoss(8)
        xr, xg = cases.td_inputs(name)
        xg = xg.requires_grad_(True)
        loss, vals = m.time_domain_loss(xr, xg)
        loss.backward()
        feats0 = m.time_domain_losses[0].calculate_time_domain_features(xr)
        save(name, loss=np.float64(loss.item()), scale_losses=np.array([v.item() for v in vals], np.float64),
             dx=xg.grad.numpy(), feats0=feats0.detach().numpy())


def tdopt_cases():
    """TimeDomainFeatureLoss with non-default options, the default loss on a very long row, and AverageFilter alone."""
    from ste_gan.layers.average_filter import AverageFilter
    from ste_gan.losses.time_domain_loss import TimeDomainFeatureLoss
    for name, (B, T, win, shift, pad, aw) in cases.TDOPT_CASES.items():
        m = TimeDomainFeatureLoss(8, win, shift, apply_padding_windowing=pad, average_filter_window_size=aw)
        xr, xg = det_input(name + ".real", (B, T, 8), "tanh"), det_input(name + ".gen", (B, T, 8), "tanh").requires_grad_(True)
        loss = m.time_domain_loss(xr, xg)
        loss.backward()
        save(name, loss=np.float64(loss.item()), dx=xg.grad.numpy(),
             feats=m.calculate_time_domain_features(xr).detach().numpy(),
             low=m.double_average(xr).detach().numpy())
    name, B, T = cases.TD_LONG
    m = MultiTimeDomainFeatureLoss(8)
    xr, xg = det_input(name + ".real", (B, T, 8), "tanh"), det_input(name + ".gen", (B, T, 8), "tanh").requires_grad_(True)
    loss, vals = m.time_domain_loss(xr, xg)
    loss.backward()
    save(name, loss=np.float64(loss.item()), scale_losses=np.array([v.item() for v in vals], np.float64),
         dx=xg.grad.numpy())
    for name, (B, Cc, T, w, pad) in cases.AVG_CASES.items():
        x = det_input(name + ".x", (B, Cc, T)).requires_grad_(True)
        y = AverageFilter(Cc, w, pad)(x)
        gy = det_input(name + ".gy", tuple(y.shape))
        y.backward(gy)
        save(name, y=y.detach().numpy(), dx=x.grad.numpy())


def valid_cases():
    """Restatement of the validation pass, ste_gan/train.py:292-350, over the reference modules (CPU: the fp16 autocast
    around it disables itself) on a list of synthetic batches standing in for valid_loader."""
    from ste_gan.train_utils import mean_error, phoneme_accuracy, phoneme_accuracy_no_silence
    for name, (ch, (ms, nl, ff), _) in cases.VALID_CASES.items():
        netG = load_det(EMGGeneratorGanTTS("SPEECH_UNITS", 256, 17, 8, channels=ch),
                        oshapes.generator_shapes("SPEECH_UNITS", ch))
        enc = make_encoder(ms, nl, ff)
        multi_td = MultiTimeDomainFeatureLoss(8)
        enc_loss = EMGEncoderLoss(enc)
        with torch.no_grad():
            netG.eval()
            td_errors, su_errors, phoneme_errors, wave_errors = [], [], [], []
            n = ok = sil = ok_ns = 0
            for b in cases.valid_batches(name):
                x_t = b["x_real"]
                x_pred_t = netG(b["feats"], b["sess"], torch.zeros_like(b["sess"]))
                wave_errors.append(F.mse_loss(x_pred_t, x_t).item())
                td_errors.append(multi_td(x_t, x_pred_t).item())
                eo = enc_loss(x_pred_t, b["su"], b["phonemes"])
                su_errors.append(eo.speech_unit_loss.item())
                phoneme_errors.append(eo.phoneme_loss.item())
                n += eo.num_phones; ok += eo.num_correct_phones
                sil += eo.num_silence_phones; ok_ns += eo.num_correct_phones_no_silence
        save(name, speech_unit=np.float64(mean_error(su_errors)), multi_td=np.float64(mean_error(td_errors)),
             phoneme=np.float64(mean_error(phoneme_errors)), waveform=np.float64(mean_error(wave_errors)),
             phoneme_accuracy_avg=np.float64(phoneme_accuracy(n, ok)),
             phoneme_accuracy_avg_no_sil=np.float64(phoneme_accuracy_no_silence(n, ok_ns, sil)),
             counters=np.array([n, sil, ok, ok_ns], np.int64),
             per_batch=np.array([wave_errors, td_errors, su_errors, phoneme_errors], np.float64))


def dataset_cases():
    """The reference's EMGDataset (data/emg_dataset.py) over the synthetic on-disk corpus of cases.write_corpus, built with
    the flags data/loader.py:28-48 passes: which utterances each partition keeps (and in what order), the id -> index
    maps, the per-item indices and lengths, the strict length check and the overlap check."""
    import json
    from pathlib import Path
    from ste_gan.data.emg_dataset import EMGDataset
    rec = {}
    with tempfile.TemporaryDirectory() as td:
        root = Path(td)
        cases.write_corpus(root, strict_ok=True)
        L = cases.CORPUS_EMG_LENGTH
        tr = EMGDataset(root, "train", strict=True, filter_by_length=True, only_include_voiced=True, train_emg_length=L)
        va = EMGDataset(root, "valid", session_id_to_idx=tr.session_id_to_idx, speaking_mode_id_to_idx=tr.speaking_mode_id_to_idx,
                        only_include_voiced=True, filter_by_length=True, train_emg_length=L, strict=True)
        te = EMGDataset(root, "test", session_id_to_idx=tr.session_id_to_idx, speaking_mode_id_to_idx=tr.speaking_mode_id_to_idx,
                        only_include_voiced=True, filter_by_length=False, train_emg_length=L, strict=True)
        EMGDataset.check_no_data_overlap([tr, va, te])
        allm = EMGDataset(root, "train", strict=False, filter_by_length=False, only_include_voiced=False, train_emg_length=L)
        sil = EMGDataset(root, "train", strict=False, filter_by_length=True, only_include_voiced=False,
                         only_include_silent=True, train_emg_length=L)
        for tag, ds in (("train", tr), ("valid", va), ("test", te), ("train_all", allm), ("train_silent", sil)):
            items = [ds[i] for i in range(len(ds))]
            rec[tag] = dict(utt_ids=list(ds.utt_ids), num_filtered=int(ds.num_filtered), emg_lengths=list(ds.emg_lengths),
                            session_id_to_idx=dict(ds.session_id_to_idx), speaking_mode_id_to_idx=dict(ds.speaking_mode_id_to_idx),
                            session_index=[int(it["SESSION_INDEX"]) for it in items],
                            speaking_mode_index=[int(it["SPEAKING_MODE_IDX"]) for it in items],
                            unit_frames=[int(it["SPEECH_UNITS"].shape[0]) for it in items],
                            mfcc_frames=[int(it["MFCCS"].shape[0]) for it in items],
                            emg_checksum=[float(it["REAL_EMG"].double().sum()) for it in items],
                            num_sessions=ds.num_sessions, num_speaking_modes=ds.num_speaking_modes,
                            num_emg_channels=ds.num_emg_channels)
        # strict mode on a corpus whose EMG is 8 samples longer than 16 x frames (utterance u05): __getitem__ asserts
        root2 = root / "nonstrict"
        cases.write_corpus(root2, strict_ok=False)
        ds = EMGDataset(root2, "train", strict=True, filter_by_length=True, only_include_voiced=True, train_emg_length=L)
        bad = []
        for i in range(len(ds)):
            try:
                ds[i]
            except AssertionError:
                bad.append(ds.utt_ids[i])
        rec["strict_violations"] = bad
        try:
            EMGDataset.check_no_data_overlap([tr, tr])
            rec["overlap_raises"] = False
        except ValueError:
            rec["overlap_raises"] = True
    with open(os.path.join(HERE, "dataset_corpus.json"), "w") as fp:
        json.dump(rec, fp, indent=1, sort_keys=True)
    print(f"  dataset_corpus.json: {sorted(rec)}")


def enc_cases():
    for name, (ms, nl, ff, B, T) in cases.ENC_CASES.items():
        enc = make_encoder(ms, nl, ff)
        emg, su, ph = cases.enc_inputs(name)
        emg = emg.requires_grad_(True)
        lossmod = EMGEncoderLoss(enc)
        out = lossmod(emg, su, ph)
        (out.speech_unit_loss + out.phoneme_loss).backward()
        save(name, su_pred=out.speech_unit_pred.detach().numpy(), ph_pred=out.phoneme_pred.detach().numpy(),
             su_loss=np.float64(out.speech_unit_loss.item()), ph_loss=np.float64(out.phoneme_loss.item()),
             counters=np.array([out.num_phones, out.num_silence_phones, out.num_correct_phones,
                                out.num_correct_phones_no_silence], np.int64),
             argmax=out.phoneme_pred.argmax(-1).numpy(), demg=emg.grad.numpy())


def step_cases():
    """Restatement of ste_gan/train.py:165-268 over the reference modules (train.py itself needs
    omegaconf + tensorboard and a dataset).  CPU: autocast/GradScaler are no-ops."""
    for name, (ch, (ms, nl, ff), B, T, steps) in cases.STEP_CASES.items():
        netG = load_det(EMGGeneratorGanTTS("SPEECH_UNITS", 256, 17, 8, channels=ch),
                        oshapes.generator_shapes("SPEECH_UNITS", ch))
        netD = load_det(DiscriminatorSmall(8), oshapes.discriminator_shapes(True))
        enc = make_encoder(ms, nl, ff)
        multi_td = MultiTimeDomainFeatureLoss(8)
        enc_loss = EMGEncoderLoss(enc)
        optG = ste_gan.OPTIMIZER(netG.parameters())
        optD = ste_gan.OPTIMIZER(netD.parameters())
        rec = {}
        relu_in = []
        for m in netG.modules():
            if isinstance(m, torch.nn.ReLU):
                m.register_forward_hook(lambda mod, inp, out: relu_in.append(float(inp[0].detach().abs().min())))
        for step in range(steps):
            b = cases.step_batch(name, step)
            netD.zero_grad(); netG.zero_grad()
            netG.train()
            x_t = b["x_real"]
            x_pred = netG(b["feats"], b["sess"], torch.zeros_like(b["sess"]))
            D_fake_det = netD(x_pred.detach())
            D_real = netD(x_t)
            loss_D = 0
            for sc in D_fake_det:
                loss_D += F.mse_loss(sc[-1], torch.zeros_like(sc[-1]))
            for sc in D_real:
                loss_D += F.mse_loss(sc[-1], torch.ones_like(sc[-1]))
            loss_D.backward()
            d_gn, d_gh = grad_summary([(k, p.grad) for k, p in netD.named_parameters()])
            optD.step()
            loss_G = 0
            D_fake = netD(x_pred)
            D_real = netD(x_t)
            adv = 0
            for sc in D_fake:
                adv += F.mse_loss(sc[-1], torch.ones_like(sc[-1]))
            loss_G += adv
            td = multi_td(x_t, x_pred)
            loss_G += 15.0 * td
            eo = enc_loss(x_pred, b["su"], b["phonemes"])
            loss_G += 1.0 * eo.speech_unit_loss
            loss_G += 1.0 * eo.phoneme_loss
            fm = 0
            for i in range(len(D_fake)):
                for j in range(len(D_fake[i]) - 1):
                    fm += F.l1_loss(D_fake[i][j], D_real[i][j].detach())
            loss_G += 7.0 * fm
            loss_G.backward()
            g_gn, g_gh = grad_summary([(k, p.grad) for k, p in netG.named_parameters()])
            optG.step()
            margin_g = min(relu_in); relu_in.clear()
            d_maps = [fm.detach() for out in (D_fake, D_real) for sc in out for fm in sc[:-1]]
            margin_d = min(float(fm.abs().min()) for fm in d_maps)
            # The discriminator's widths are fixed (discriminator.py:54-60,76-81), so even this 3 x 80-sample fixture has
            # ~10^6 leaky-ReLU units per pass with outputs of scale ~0.05: the smallest |output| is ~scale / units ~ 1e-8
            # whatever the seed -- a 1e-6 margin cannot be had for D.  What CAN be asserted: the units close enough to the
            # kink for an fp32 summation order to flip their slope are a vanishing share, so a flip moves a gradient NORM
            # by far less than the tests' bounds (the tests bound norms and L2 errors, never single entries); and the
            # build's own step is bit-reproducible (tests/test_gpu_reproducible.py), so a flip cannot come and go.
            d_units = sum(fm.numel() for fm in d_maps)
            d_near = sum(int((fm.abs() < 1e-6).sum()) for fm in d_maps)
            print(f"  step {step}: kink margins  G relu inputs {margin_g:.2e}   D lrelu outputs {margin_d:.2e} "
                  f"({d_near} of {d_units} D units within 1e-6 of the kink)")
            assert margin_g > 1e-6, "a generator pre-activation sits on the ReLU kink: pick another case"
            assert d_near <= 1e-4 * d_units, "too many discriminator units on the leaky-ReLU kink: pick another case"
            rec[f"s{step}.kink"] = np.array([margin_g, margin_d, d_near, d_units], np.float64)
            tag = f"s{step}."
            rec[tag + "losses"] = np.array([loss_D.item(), loss_G.item(), adv.item(), td.item(),
                                            eo.speech_unit_loss.item(), eo.phoneme_loss.item(), fm.item()], np.float64)
            rec[tag + "counters"] = np.array([eo.num_phones, eo.num_silence_phones, eo.num_correct_phones,
                                              eo.num_correct_phones_no_silence], np.int64)
            rec[tag + "x_pred"] = x_pred.detach().numpy()
            rec[tag + "g_grad_norm"], rec[tag + "g_grad_head"] = g_gn, g_gh
            rec[tag + "d_grad_norm"], rec[tag + "d_grad_head"] = d_gn, d_gh
            rec[tag + "g_param_norm"], rec[tag + "g_param_head"] = grad_summary(list(netG.state_dict().items()))
            rec[tag + "d_param_norm"], rec[tag + "d_param_head"] = grad_summary(list(netD.state_dict().items()))
        save(name, **rec)


def collate_cases():
    """ste_gan_collate (data/collate.py:22-61) on synthetic utterances: the train partition under a fixed torch seed
    (its start frames come from the global torch RNG, collate.py:118-121), and the valid partition."""
    from ste_gan.data.collate import ste_gan_collate
    for name, (frames, extra, emg_len, n_mfcc, seed) in cases.COLLATE_CASES.items():
        rec = {}
        for part in ("train", "valid"):
            torch.manual_seed(seed)
            out = ste_gan_collate(cases.collate_samples(name), partition=part, emg_train_length=emg_len,
                                  chunksize=emg_len, hopsize=ste_gan.HOPSIZE)
            for k, v in out.items():
                # large float outputs are kept as a position-weighted checksum (cases.poscheck) to keep fixtures small
                big = v.is_floating_point() and v.numel() > 50_000
                rec[f"{part}.{k}" + (".poscheck" if big else "")] = cases.poscheck(v) if big else v.numpy()
        save(name, **rec)


if __name__ == "__main__":
    which = sys.argv[1:] or ["gen", "disc", "td", "tdopt", "enc", "step", "valid", "collate", "dataset"]
    for w in which:
        {"gen": gen_cases, "disc": disc_cases, "td": td_cases, "tdopt": tdopt_cases, "enc": enc_cases, "step": step_cases,
         "valid": valid_cases, "collate": collate_cases, "dataset": dataset_cases}[w]()
    leaked = [os.path.join(dp, d) for dp, dn, _ in os.walk("/root/reference") for d in dn if d == "__pycache__"]
    assert not leaked, f"__pycache__ written into the reference tree: {leaked}"
