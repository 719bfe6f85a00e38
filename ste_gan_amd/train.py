"""STE-GAN trainer for MI355X -- drop-in for ``ste_gan/train.py``.

``train(cfg, model_directory, checkpoint, torch_device, debug, emg_enc_ckpt)``, ``main`` and
``parse_args`` keep the reference's signatures and CLI flags (train.py:39-46,505-574).  The hot loop
(train.py:165-268) lives in :class:`GANTrainStep`:

* same dependencies and values -- G forward; D(fake.detach()), D(real); LSGAN D loss; optD step; then with the
  UPDATED D: D(fake), D(real); adversarial + multi-TD + speech-unit + phoneme + feature-matching losses;
  optG step; the spectral-norm power iteration runs on each of the four D forwards.  The D-independent terms
  (multi-TD, EMG encoder) and their gradient w.r.t. the generated EMG are evaluated on a side HIP stream beside the
  discriminator step rather than after it (they read nothing it writes), and enter the generator's backward pass
  as a second root;
* every arithmetic op is a libstegan_hip.so kernel; no ``.item()`` inside the step -- the loss scalars and the
  phoneme counters are read back once, by :meth:`GANTrainStep.read_stats`;
* work the reference does only to throw away is skipped: D's weight gradients inside ``loss_G.backward()``
  (zeroed by the next ``zero_grad``) and the frozen encoder's weight gradients;
* one process per GPU (``torchrun --nproc-per-node N -m ste_gan_amd.train ...``): :func:`main` initialises
  ``torch.distributed`` from the launcher's environment, :func:`train` broadcasts rank 0's weights, shards every
  epoch's batch plan over the ranks (``data.loader.plan_epoch``) and leaves validation, TensorBoard and checkpoint
  files to rank 0; inside the step the two flat gradient arenas (G, D) are all-reduced over RCCL on a side HIP
  stream, the D-gradient exchange overlapping the multi-TD and EMG-encoder losses, which do not depend on D
  (SURVEY.md section 5.8).

Arithmetic.  ``cfg.train.mixed_precision`` selects, as in the reference (train.py:151,181,204: fp16 autocast +
GradScaler on CUDA), the reduced-precision path: here the convolution operands are rounded to bf16 while they are
staged (``ops.set_compute_dtype("bf16")``: fp32 accumulation, activations, master weights and optimiser state, no
loss scaling needed -- bf16 keeps fp32's exponent range).  With the flag off the step runs on the exact fp32 MFMA
kernels: that is the parity path -- on the CPU the reference's autocast / GradScaler disable themselves, so its
own parity path is plain fp32 too.
"""
from __future__ import annotations

import argparse
import itertools
import logging
import os
import random
import sys
import time
from pathlib import Path
from typing import Dict, Optional

import numpy as np
import torch
import torch.distributed as dist

from . import constants as K
from . import ops
from .constants import DataType
from .losses.emg_encoder_loss import EMGEncoderLoss, EMGEncoderLossOutput
from .losses.gan_losses import lsgan_pair_loss, feature_matching_loss, lsgan_loss
from .losses.time_domain_loss import MultiTimeDomainFeatureLoss
from .models.discriminator import init_emg_discriminators
from .models.emg_encoder import load_emg_encoder
from .models.generator import init_emg_generator
from .optim import ExponentialLR, FlatParams, FusedAdamW
from .parallel import FlatGradAllReduce
from .train_utils import (add_eval_hyperparams_to_parser, create_ste_gan_model_name, load_config, mean_error,
                          phoneme_accuracy, phoneme_accuracy_no_silence)
from .utils.common import load_latest_checkpoint


def crop_to_hop(chunk_size: int) -> int:
    """The generator emits 16 EMG samples per Soft-SU frame, so real EMG is cropped to (chunk//16)*16.

    The reference's collate keeps ``chunk_size`` real samples (data/collate.py:44,81-83) and its own step
    then fails for chunk sizes that are not multiples of 16."""
    return (chunk_size // K.HOPSIZE) * K.HOPSIZE


class GANTrainStep:
    """One iteration of ``ste_gan/train.py:165-268`` on one GPU (one rank of a data-parallel job)."""

    def __init__(self, cfg, netG, netD, emg_encoder_loss: Optional[EMGEncoderLoss], multi_td_loss,
                 optG: FusedAdamW, optD: FusedAdamW, process_group=None, scaler=None):
        """``scaler`` (optim.GradScaler): the reference's loss scaling (train.py:151,198-199,266-268), decided on the device.
        The fp16 operand mode gets one by default (torch's defaults: 65536, x2 every 2000 clean iterations, /2 after a
        skipped step); in the fp32 / bf16 modes a power-of-two scale is exact and none is made."""
        self.cfg = cfg
        if scaler is None and ops.COMPUTE_DTYPE == "f16":
            from .optim import GradScaler
            scaler = GradScaler(optG.flat.flat.device)
        self.scaler = scaler
        self.netG, self.netD = netG, netD
        self.emg_encoder_loss, self.multi_td_loss = emg_encoder_loss, multi_td_loss
        self.optG, self.optD = optG, optD
        self.speech_feature_type = cfg.model.speech_feature_type
        self.d_params = [p for p in netD.parameters()]
        self.comm = FlatGradAllReduce(process_group) if dist.is_available() and dist.is_initialized() else None
        self.last: Dict[str, torch.Tensor] = {}
        import os
        self.overlap_encoder = os.environ.get("STEGAN_OVERLAP_ENCODER", "1") != "0"
        # feature-matching gradients are added inside the discriminator convs' activation backward instead of by one
        # autograd accumulation kernel per feature map (ops.fm_fold_begin)
        self.fold_fm = os.environ.get("STEGAN_FOLD_FM", "1") != "0"
        # the generator step's backward pass through the discriminators runs before the side branch is joined (_p_g2)
        self.early_d_backward = os.environ.get("STEGAN_EARLY_D_BWD", "1") != "0"
        self._side = None
        # the side branch in two halves, each forked and joined inside its phase: forward + losses beside the discriminator
        # step (phase_d), backward beside the D update and the generator-side discriminator passes (phase_g).  Set by
        # GraphedTrainStep under torch.distributed, where the phases are separate hipGraphs around the gradient exchanges
        self.split_side = False
        # torch.distributed with the exchanges INSIDE one captured graph (GraphedTrainStep, "single"): the side branch is not
        # joined at the end of phase_d -- it runs on beside the D-gradient exchange, which only the main stream awaits
        self.join_side_after_phase_d = True

    def _side_stream(self):
        if self._side is None:
            # (experiment hook: a priority for the encoder branch's stream; captured graphs showed no effect, see DESIGN)
            self._side = torch.cuda.Stream(priority=int(os.environ.get("STEGAN_SIDE_PRIORITY", "0")))
        return self._side

    def _set_d_grad(self, flag: bool):
        for p in self.d_params:
            p.requires_grad_(flag)

    def __call__(self, batch_dict: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        """One train step = three phases with the two gradient exchanges between them (no-ops in a single process).  The
        phases keep their state in ``self._st`` so that GraphedTrainStep can capture each of them as its own hipGraph under
        torch.distributed (collectives stay outside the graphs)."""
        try:
            self.phase_d(batch_dict)
            self.exchange_d()
            self.phase_g()
            self.exchange_g()
            return self.phase_opt()
        except BaseException:
            self.abort()
            raise

    def abort(self) -> None:
        """A phase raised: leave no state behind that a later step would trust (the open zero pool, gradients already
        accumulated into the arenas, weight gradients parked for side streams, the loss kernels' attached gradient scale).
        Must never hide the exception it is cleaning up after: a sticky HIP error makes ``synchronize`` raise again, and on
        a stream that is being captured (a failed graph capture) a fill would be RECORDED, not executed -- there the arenas
        are only marked dirty, so the next eager ``zero_grad`` fills them."""
        capturing = torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()
        if capturing:
            self._join_captured_forks()
        for fn in (ops.zero_pool_abort, ops.wgrad_abort):
            try:
                fn()
            except Exception as e:                                   # noqa: BLE001 -- the original exception is what matters
                logging.warning(f"GANTrainStep.abort: {fn.__name__} raised {type(e).__name__}: {e}")
        self._st = None
        if self.scaler is not None:
            self.scaler.detach()
        for opt in (self.optG, self.optD):
            try:
                if capturing:
                    opt.flat.clean = False
                    if opt.pack is not None:
                        opt.pack.clean = False
                else:
                    opt.discard_grads()
            except Exception as e:                                   # noqa: BLE001
                logging.warning(f"GANTrainStep.abort: discarding gradients raised {type(e).__name__}: {e}")

    def _join_captured_forks(self) -> None:
        """A phase raised INSIDE a graph capture: every stream the step forked into the capture (side branch, discriminator
        forks, weight-gradient streams, the communication stream) is joined back into the capturing stream, so that the
        capture can END -- hipStreamEndCapture refuses a capture with unjoined work and leaves the stream capturing, after
        which no further capture can begin in the process (seen with the injected failure of
        tests/test_gpu_train_step.py: "capturing stream has unjoined work", then "Cannot register the state during capturing
        stage").  Only streams that ARE capturing are waited for: an event of a stream outside the capture would violate
        its isolation."""
        from .models.discriminator import _MultiDisc
        cands = [getattr(self, "_side", None), getattr(_MultiDisc, "_side", None), getattr(_MultiDisc, "_third", None)]
        cands += list(ops._WG_SIDE) + list(ops._SINK_STREAMS.values())
        if self.comm is not None:
            cands.append(getattr(self.comm, "_stream", None))
        main = torch.cuda.current_stream()
        seen = {main.cuda_stream}
        for s_ in cands:
            if s_ is None or s_.cuda_stream in seen:
                continue
            seen.add(s_.cuda_stream)
            try:
                with torch.cuda.stream(s_):
                    forked = torch.cuda.is_current_stream_capturing()
                if forked:
                    main.wait_stream(s_)
            except Exception as e:                                   # noqa: BLE001 -- best effort; the original error follows
                logging.warning(f"GANTrainStep.abort: joining a forked stream raised {type(e).__name__}: {e}")

    def exchange_d(self):
        """SUM all-reduce of the discriminator's gradient arena (side stream, RCCL); awaited in phase_g."""
        self._d_handle = None
        if self.comm is not None and self.cfg.train.loss_adversarial:
            self.optD.flat.gather_grads()      # any gradient autograd did not write through its arena view
            self._d_handle = self.comm.launch(self.optD.flat.grad)

    def exchange_g(self):
        self._g_handle = None
        if self.comm is not None:
            self.optG.flat.gather_grads()
            self._g_handle = self.comm.launch(self.optG.flat.grad)

    def phase_d(self, batch_dict: Dict[str, torch.Tensor]) -> None:
        """Generator forward, the D-independent losses (side stream), the discriminator step up to its gradients."""
        if self.scaler is not None:
            self.scaler.attach()               # host side: the loss kernels launched from here on scale the gradients they write
        self._p_fwd(batch_dict)
        st = self._st
        # ---- D-independent generator losses (multi-TD, frozen EMG encoder) and THEIR gradient w.r.t. the generated EMG.
        # The reference computes them inside the generator step (train.py:213-256), after the discriminator update, but
        # they read nothing the discriminator step writes.  Here they run -- forward AND backward, down to d loss / d x_pred
        # -- on a side HIP stream forked right after the generator forward, beside the whole discriminator step; the
        # generator step then hands that gradient to autograd together with the discriminator-side losses
        # (autograd.backward([loss, x_pred], [1, g_side])), so the generator is still traversed once and its parameters
        # receive the same sums.  The encoder branch works on T_su = 12 frames: ~130 short, latency-bound launches that
        # fill a fraction of the chip, 2.3 ms when they run alone.  (The first arrangement forked the branch inside the
        # generator step and relied on loss_G.backward() to overlap the two sub-graphs; the round-2 kernel trace showed the
        # captured graph running them one after the other: 0.8 ms of encoder forward and 1.5 ms of encoder backward with
        # every other queue idle.)
        main_stream = st["main_stream"]
        side = self._side_stream() if self.overlap_encoder else None
        if side is not None:
            side.wait_stream(main_stream)                      # x_pred, the batch
            torch.cuda.set_stream(side)
        split = side is not None and self.split_side
        try:
            if split:
                self._p_side_fwd()
            else:
                self._p_side()
        finally:
            if side is not None:
                torch.cuda.set_stream(main_stream)
        self._p_disc()
        if side is not None and ((self.comm is not None and self.join_side_after_phase_d) or split):
            # under torch.distributed this phase ends with the side stream joined: the gradient exchange that follows is
            # issued from the main stream, and a captured phase (GraphedTrainStep) must end with every forked stream joined.
            # The branch's BACKWARD pass then runs beside the first half of phase_g (split_side); without the split the
            # phase lasts until the whole branch is done -- 1.3 ms after the discriminator's gradients are ready.
            main_stream.wait_stream(side)
            side = None
        st["side"] = side
        st["side_bwd_pending"] = split

    def _p_fwd(self, batch_dict: Dict[str, torch.Tensor]) -> None:
        """Start of a step: zero scratch / gradient arenas, generator forward (train.py:182)."""
        netG = self.netG
        ops.zero_pool_begin(batch_dict[DataType.REAL_EMG].device)     # one fill for the step's zero-initialised scratch
        self.optD.zero_grad()
        self.optG.zero_grad()
        speech_units_t = batch_dict[DataType.SPEECH_UNITS]
        s_t1 = speech_units_t if self.speech_feature_type == DataType.SPEECH_UNITS else batch_dict[self.speech_feature_type]
        ops.mark("step start")
        netG.train()
        x_pred_t = netG(s_t1, batch_dict[DataType.SESSION_INDEX], batch_dict[DataType.SPEAKING_MODE_INDEX])   # train.py:182
        ops.mark("G forward done")
        self._st = dict(out={}, x_pred_t=x_pred_t, x_t=batch_dict[DataType.REAL_EMG], side=None,
                        main_stream=torch.cuda.current_stream(), loss_side=None, g_side=None,
                        speech_units_t=speech_units_t, phoneme_targets=batch_dict[DataType.PHONEMES])

    def _p_side_fwd(self) -> None:
        """The D-independent generator losses (forward), on the CURRENT stream ...

        Every term is computed WITH its loss weight folded into the kernel (value and gradient), as a plain summand whose
        upstream gradient is one; the weighted total and the unweighted values the reference logs come out of ONE
        ``ops.scalar_combine`` launch instead of a chain of 0-dim torch mul / add kernels (train.py:213-256)."""
        t = self.cfg.train
        st = self._st
        out, x_pred_t, x_t = st["out"], st["x_pred_t"], st["x_t"]
        speech_units_t, phoneme_targets = st["speech_units_t"], st["phoneme_targets"]
        xp = x_pred_t.detach().requires_grad_(True)        # leaf: this branch's backward stops at the generated EMG
        # one alias of the leaf per loss term that reads it: their gradients meet in ONE hand-written sum (ops.fan_out)
        users = [bool(t.loss_multi_td_error),
                 bool((t.loss_speech_unit_error or t.loss_phoneme_error) and self.emg_encoder_loss is not None),
                 bool(t.loss_waveform_error)]
        alias = iter(ops.fan_out(xp, sum(users))) if sum(users) > 1 else iter([xp] * 3)
        xp_td, xp_enc, xp_wave = (next(alias) if u else None for u in users)
        terms = []                                          # (key in ``out``, weighted 0-dim loss, weight)
        ops.mark("side: start")
        # A term that is enabled with weight 0.0 is still LOGGED with its real value, as the reference does (train.py:213-262
        # logs the term itself): its kernel runs with weight 1 and the term is neither a summand of the total nor a root of
        # the backward pass (_w_eff / _live).
        if t.loss_multi_td_error:
            w = float(t.loss_multi_td_weight)
            terms.append(("multi_td", self.multi_td_loss(x_t, xp_td, weight=_w_eff(w), unit_grad=True), w))
        ops.mark("side: TD done")
        if (t.loss_speech_unit_error or t.loss_phoneme_error) and self.emg_encoder_loss is not None:
            w_su = float(t.loss_speech_unit_weight) if t.loss_speech_unit_error else 0.0
            w_ph = float(t.loss_phoneme_weight) if t.loss_phoneme_error else 0.0
            enc_out: EMGEncoderLossOutput = self.emg_encoder_loss(xp_enc, speech_units_t, phoneme_targets,
                                                                  su_weight=_w_eff(w_su) if t.loss_speech_unit_error else 0.0,
                                                                  ph_weight=_w_eff(w_ph) if t.loss_phoneme_error else 0.0,
                                                                  unit_grad=True)
            if t.loss_speech_unit_error:
                terms.append(("speech_unit", enc_out.speech_unit_loss, w_su))
            if t.loss_phoneme_error:
                terms.append(("phoneme", enc_out.phoneme_loss, w_ph))
            out["phoneme_counters"] = enc_out._counters
        if t.loss_waveform_error:
            w = float(t.loss_waveform_weight)
            terms.append(("waveform", ops.multi_pair_loss([xp_wave], [x_t], 1, _w_eff(w), unit_grad=True), w))
        ops.mark("side: encoder forward + losses done")
        loss_side = None
        if terms:
            n = len(terms)
            rows = [[(1.0 if w != 0.0 else 0.0) for _, _, w in terms]] + \
                   [[(1.0 / _w_eff(w) if j == i else 0.0) for j in range(n)] for i, (_, _, w) in enumerate(terms)]
            vals = ops.scalar_combine([v for _, v, _ in terms], rows)
            loss_side = vals[0]
            for (key, _, _), v in zip(terms, vals[1:]):
                out[key] = v
        st["loss_side"], st["xp"], st["side_roots"] = loss_side, xp, [v for _, v, w in terms if w != 0.0]

    def _p_side_bwd(self) -> None:
        """... and their gradient w.r.t. the generated EMG (the backward pass of the side branch)."""
        st = self._st
        xp, roots = st.pop("xp"), st.pop("side_roots")
        g_side = None
        if roots:
            one = ops.one_like(xp)
            (g_side,) = torch.autograd.grad(roots, xp, grad_outputs=[one] * len(roots))
        ops.mark("side: backward done")
        st["g_side"] = g_side

    def _p_side(self) -> None:
        self._p_side_fwd()
        self._p_side_bwd()

    def _p_disc(self) -> None:
        """The discriminator step up to its finalized gradients (train.py:189-199)."""
        t = self.cfg.train
        netD = self.netD
        st = self._st
        out, x_pred_t, x_t = st["out"], st["x_pred_t"], st["x_t"]
        if t.loss_adversarial:
            if hasattr(netD, "forward_pair_logits"):
                # train.py:190-196 on the concatenated batch: fake | real logits against zeros | ones, unsliced, one launch
                with ops.act_fuse_scope():       # only the logits carry gradient: the layers may chain their lrelu'
                    loss_D = lsgan_pair_loss(netD.forward_pair_logits(x_pred_t.detach(), x_t), 0.0, 1.0, unit_grad=True)
                ops.mark("D step: forward + loss done")
            else:
                D_fake_det, D_real = netD(x_pred_t.detach()), netD(x_t)
                # both terms are plain summands of the scalar handed to backward(): unit upstream gradient
                loss_D = lsgan_loss(D_fake_det, 0.0, unit_grad=True) + lsgan_loss(D_real, 1.0, unit_grad=True)
            d_async = os.environ.get("STEGAN_D_WGRAD_ASYNC", "0") == "1"     # experiment, see below
            if d_async:
                from . import ops as _ops
                _ops.wgrad_async(True)
            try:
                loss_D.backward()                  # (weight gradients aside, as in the generator's backward: 8.23 vs
            finally:                               #  7.84 ms/step -- three streams are already busy here)
                if d_async:
                    _ops.wgrad_async(False)
            ops.mark("D step: backward done")
            self.optD.finalize_grads()
            ops.mark("D step: gradients finalized")
            out["discriminator"] = loss_D.detach()

    def phase_g(self) -> None:
        """D's optimiser step (after its gradient exchange), the generator-side discriminator passes, the generator's
        backward pass up to its gradients."""
        st = self._st
        if st.get("side_bwd_pending"):
            main_stream, side = st["main_stream"], self._side_stream()
            side.wait_stream(main_stream)
            torch.cuda.set_stream(side)
            try:
                self._p_side_bwd()             # (autograd runs each node on the stream of its forward: this side stream)
            finally:
                torch.cuda.set_stream(main_stream)
            st["side"] = side                  # joined in _p_g2, in front of the generator's backward pass
            st["side_bwd_pending"] = False
        self._p_g1()
        self._p_g2()

    def _p_g1(self) -> None:
        """D's optimiser step and the generator-side discriminator passes with the updated D, up to the D-side losses."""
        t = self.cfg.train
        netD = self.netD
        st = self._st
        out, x_pred_t, x_t = st["out"], st["x_pred_t"], st["x_t"]
        if t.loss_adversarial:
            if self.comm is not None:
                self.comm.wait(getattr(self, "_d_handle", None))
                self.optD.step(grad_scale=1.0 / self.comm.world_size, finalized=True, scaler=self.scaler)
            else:
                self.optD.step(finalized=True, scaler=self.scaler)                       # D weights change HERE
            ops.mark("D optimiser done")

        # ---- generator-side discriminator passes with the updated D (train.py:206-211, 257-264)
        self._set_d_grad(False)          # D's weight grads from loss_G are discarded by the reference: skip them
        from . import ops as _ops
        fold = self.fold_fm and bool(t.loss_feat_match_error) and float(t.loss_feat_match_weight) != 0.0
        if fold:
            _ops.fm_fold_begin()         # the feature maps' producers add the feature-matching gradients themselves
        # the generated EMG has two gradient sources -- the discriminators here and the side branch's gradient, handed to
        # autograd as a root in _p_g2: they meet in ONE hand-written sum (ops.fan_out) instead of the engine's ATen add
        x_for_d, st["x_pred_root"] = ops.fan_out(x_pred_t, 2)
        st["x_for_d"] = x_for_d
        try:
            if fold:                          # feature-matching gradients are parked, not returned to autograd (ops.ActFuse)
                with ops.act_fuse_scope():
                    D_fake = netD(x_for_d)
            else:
                D_fake = netD(x_for_d)
            ops.mark("G step: D(fake) done")
            with torch.no_grad():
                D_real = netD(x_t)
            ops.mark("G step: D(real) done")
            main_terms = []                                              # (key | None, weighted 0-dim loss, weight)
            if t.loss_adversarial:
                main_terms.append((None, lsgan_loss(D_fake, 1.0, unit_grad=True), 1.0))
            if t.loss_feat_match_error:
                w_fm = float(t.loss_feat_match_weight)                   # folded into the loss kernels
                loss_feat_w = feature_matching_loss(D_fake, D_real, weight=_w_eff(w_fm), unit_grad=True, fold=fold)
                n_fm = sum(len(sc) - 1 for sc in D_fake)
                if fold and _ops._FM_COUNT[0] != n_fm:
                    # a gradient that went back to autograd would meet feature maps whose producers no longer apply their
                    # own activation derivative (ops.act_fuse_scope): refuse instead of training on a wrong gradient
                    raise RuntimeError(f"ste_gan_amd: {n_fm - _ops._FM_COUNT[0]} of {n_fm} feature-matching gradients were "
                                       "not parked for the backward kernels")
                main_terms.append(("feature_matching", loss_feat_w, w_fm))
            ops.mark("G step: losses done")
        except BaseException:
            _ops.fm_fold_abort()
            self._set_d_grad(True)
            raise
        st["main_terms"], st["fold"] = main_terms, fold

    def _p_g2(self) -> None:
        """Join of the side branch, ONE backward pass through the generator for both gradient sources, its gradients."""
        from . import ops as _ops
        st = self._st
        out, x_pred_t, side, main_stream = st["out"], st["x_pred_t"], st["side"], st["main_stream"]
        loss_side, g_side, main_terms, fold = st["loss_side"], st["g_side"], st["main_terms"], st["fold"]
        try:
            # one pass through the generator for both gradient sources; the scalar roots meet a persistent 1.0 instead of
            # autograd's ones_like fills
            roots, grads = [], []
            one = ops.one_like(x_pred_t)
            for _, v, w in main_terms:
                if v.requires_grad and w != 0.0:
                    roots.append(v); grads.append(one)
            x_root, x_for_d = st.pop("x_pred_root", x_pred_t), st.pop("x_for_d", None)
            if (self.early_d_backward and side is not None and g_side is not None and roots and x_for_d is not None
                    and x_for_d is not x_root and x_for_d.requires_grad):
                # Round 5: the discriminators' input gradient reads nothing of the side branch, so it runs BEFORE the join,
                # beside the tail of the encoder's backward pass (the main stream used to idle ~0.45 ms at the join, then run
                # these ~0.4 ms alone).  Same sums: the two gradients of x_pred meet in fan_out's kernel, in the same order.
                (g_d,) = torch.autograd.grad(roots, x_for_d, grad_outputs=grads)
                ops.mark("G step: D input gradient done")
                roots, grads = [x_for_d], [g_d]
            if side is not None:
                main_stream.wait_stream(side)                  # join: the side branch's losses and gradient are complete
            ops.mark("G step: side joined")
            if g_side is not None:
                roots.append(x_root); grads.append(g_side)
            if roots:
                _ops.wgrad_async(True)         # generator backward: weight gradients beside the input-gradient chain
                try:
                    torch.autograd.backward(roots, grads)
                finally:
                    _ops.wgrad_async(False)
            # loss_G and the unweighted feature-matching value in one launch (train.py:209-264: loss_G += w * term)
            ins = [v for _, v, _ in main_terms] + ([loss_side] if loss_side is not None else [])
            loss_G = None
            if ins:
                rows = [[(1.0 if w != 0.0 else 0.0) for _, _, w in main_terms] + ([1.0] if loss_side is not None else [])]
                for i, (key, _, w) in enumerate(main_terms):
                    if key is not None:
                        rows.append([(1.0 / _w_eff(w) if j == i else 0.0) for j in range(len(ins))])
                vals = _ops.scalar_combine(ins, rows)
                loss_G = vals[0]
                k = 1
                for key, _, _ in main_terms:
                    if key is not None:
                        out[key] = vals[k]; k += 1
            if fold:
                _ops.fm_fold_end()       # raises if a parked gradient was not collected
        finally:
            _ops.fm_fold_abort()
            self._set_d_grad(True)
        st["main_terms"] = None
        ops.mark("G step: backward done")
        self.optG.finalize_grads()
        ops.mark("G step: gradients finalized")
        out["generator"] = loss_G if loss_G is not None else torch.zeros((), device=x_pred_t.device)
        out["x_pred"] = x_pred_t.detach()

    def phase_opt(self) -> Dict[str, torch.Tensor]:
        """The generator's optimiser step (after its gradient exchange)."""
        if self.comm is not None:
            self.comm.wait(getattr(self, "_g_handle", None))
            self.optG.step(grad_scale=1.0 / self.comm.world_size, finalized=True, scaler=self.scaler)
        else:
            self.optG.step(finalized=True, scaler=self.scaler)
        ops.mark("G optimiser done")
        ops.zero_pool_end()
        out = self._st["out"]
        if self.scaler is not None:
            self.scaler.update()                                   # once per iteration, behind both steps (train.py:268)
            self.scaler.detach()
            out["loss_scale"] = self.scaler.state[0]               # (a view of the live state: the scale of the NEXT step)
        self._st = None                     # drop the step's tensors
        self.last = out
        return out

    def close(self) -> None:
        """Release what the step holds on the device side of the data-parallel exchange (ordered shutdown,
        ``parallel.shutdown``)."""
        self._st = None
        self.last = {}
        self._d_handle = self._g_handle = None
        if self.comm is not None:
            self.comm.close()

    @staticmethod
    def read_stats(out: Dict[str, torch.Tensor]) -> Dict[str, float]:
        """Single synchronising readback of the step's scalars (the reference pays ~10 .item() syncs)."""
        keys = [k for k in out if k not in ("x_pred", "phoneme_counters")]
        vals = torch.stack([out[k].float().reshape(()) for k in keys]).tolist()
        stats = dict(zip(keys, vals))
        if "phoneme_counters" in out:
            c = out["phoneme_counters"].tolist()
            stats.update(num_phones=c[0], num_silence_phones=c[1], num_correct_phones=c[2],
                         num_correct_phones_no_silence=c[3])
        return stats


def _w_eff(w: float) -> float:
    """Weight a loss kernel runs with: the configured one, or 1 for a term enabled with weight 0 (logged, not trained on)."""
    return w if w != 0.0 else 1.0


def _leak_failed_graphs(*graphs) -> None:
    """Keep the CUDAGraph objects of a capture that raised alive FOR EVER.

    Their destructor is not safe to run: at::cuda::CUDAGraph::~CUDAGraph() unregisters the graph from the generator states it
    touched and, for a capture that did not complete, HIPGeneratorImpl::unregister_graph throws c10::Error ("The graph should
    be registered to the state") -- an exception out of a destructor, i.e. std::terminate: the process dies with SIGABRT as
    soon as the last reference goes (round 5 caught this with the stack of the injected-failure test; it is what turned a
    capture that merely RAISED in round 4's single-graph mode into the "silent SIGABRT during the capture").  An extra
    reference that is never dropped costs a few hundred bytes per failed capture."""
    import ctypes
    for g in graphs:
        if g is not None:
            ctypes.pythonapi.Py_IncRef(ctypes.py_object(g))


class GraphedTrainStep:
    """The whole train step captured once into a hipGraph and replayed per batch.

    A step is ~800 small launches; captured, the host only refreshes the static input buffers (and the learning rate word
    when the scheduler changed it) and calls ``hipGraphLaunch``; AdamW's step counter and bias corrections advance inside the
    graph (``stegan_adamw_advance``), so replays can queue arbitrarily far ahead of the GPU.  Requirement: fixed batch shapes.

    Single process: ONE graph.  Under torch.distributed (one process per GPU): THREE graphs -- the step's phases
    (GANTrainStep.phase_d / phase_g / phase_opt), sharing one memory pool, the side branch cut in two so that no graph
    waits for it (see __init__) -- with the two RCCL gradient all-reduces launched eagerly between them: collectives stay outside the captures, and an eager step would be bound by the host (22.7 ms of
    launch overhead against 8.8 ms of GPU time).
    """

    CAPTURE_ERROR_MODE = "thread_local"      # pinned by tests/test_host_logic.py (see __init__)
    DDP_GRAPH_DEFAULT = "three"              # arrangement under torch.distributed unless STEGAN_DDP_GRAPH says otherwise

    def __init__(self, step: GANTrainStep, example_batch: Dict[str, torch.Tensor], warmup: int = 3):
        self.step = step
        self.static = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in example_batch.items()}
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                step(self.static)              # (every rank runs the same number of exchanges)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        # the replayed step zeroes its gradient arenas inside the optimiser kernels instead of with fills at its start
        if os.environ.get("STEGAN_ZERO_IN_STEP", "1") != "0":
            step.optG.set_zero_in_step(True)
            step.optD.set_zero_in_step(True)
        self._counts = (step.optG.step_count, step.optD.step_count)
        # Prepared weight copies inside the captured step.  The generator's are re-derived at its start (its weights changed
        # at the end of the previous step): invalidate them so that the capture records the launches.  The discriminator's
        # weights change in the MIDDLE of a step and their copies are re-derived right there, so at the start of the next
        # step they are current: the capture records NO preparation in front of the discriminator step (0.2 ms of launches
        # on the critical path) and __call__ refreshes them eagerly whenever the host changed weights between replays.
        if step.optG.pack is not None:
            step.optG.pack.invalidate()
        self._refresh_d()
        # capture_error_mode "thread_local": with a process group alive, the RCCL watchdog THREAD polls its work events with
        # hipEventQuery at any time; in the default "global" mode such a call from another thread invalidates a capture in
        # progress (hipErrorStreamCaptureInvalidated, then the watchdog aborts the process) -- seen once in three runs.
        mode = dict(capture_error_mode=self.CAPTURE_ERROR_MODE)
        if os.environ.get("STEGAN_MAIN_PRIORITY"):            # experiment hook: capture on a stream with a priority
            mode["stream"] = torch.cuda.Stream(priority=int(os.environ["STEGAN_MAIN_PRIORITY"]))
        self.ddp_mode = None
        if step.comm is None:
            self.graph = torch.cuda.CUDAGraph()
            try:
                with torch.cuda.graph(self.graph, **mode):
                    self.out = step(self.static)
            except BaseException:
                _leak_failed_graphs(self.graph)
                raise
            self.graphs = None
        else:
            self.graph, self.graphs = None, None
            # ---- torch.distributed.  "three" (default): the phases as three graphs around two eagerly launched exchanges, the
            # arrangement of rounds 2-3.  "single" (STEGAN_DDP_GRAPH=single, RCCL only): ONE graph for the whole step with the two
            # gradient all-reduces CAPTURED inside it, on the communication stream (RCCL collectives are capturable like
            # NCCL's: scripts/rccl_capture_probe.py).  No graph boundary drains the streams (0.45 ms of a 6.8 ms step with three
            # graphs; 6.19 against 6.66 ms with a 1-rank group), and the exchanges overlap whatever does not depend on them by
            # construction.  It is NOT the default: in one of about six runs the process died with a silent SIGABRT (no HIP,
            # RCCL or c10d message) during the capture, after the first collective had been captured -- a crash at N > 1 costs
            # a whole run, 0.5 ms does not.  Falls back to "three" when the capture raises.
            want = os.environ.get("STEGAN_DDP_GRAPH", self.DDP_GRAPH_DEFAULT)
            if want == "single" and dist.get_backend(step.comm.group) != "nccl":
                want = "three"                                   # (gloo's collectives cannot be captured)
            if want == "single":
                try:
                    step._d_handle = step._g_handle = None
                    step.join_side_after_phase_d = False
                    step.split_side = False
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, **mode):
                        self.out = step(self.static)
                    self.graph, self.ddp_mode = g, "single"
                except Exception as e:                           # noqa: BLE001 -- reported; the three-graph path follows
                    logging.warning(f"single-graph capture with RCCL collectives failed ({type(e).__name__}: {e}); "
                                    "falling back to three graphs around eager exchanges")
                    _leak_failed_graphs(g)                       # (its destructor would abort the process: see there)
                    step.abort()
                    torch.cuda.synchronize()
                    self.graph = None
                    # the failed capture ran the HOST side of a step (optimiser epochs, prepared-copy stamps) without executing
                    # a kernel: bring the bookkeeping back to "weights as they are, copies to be re-derived", as in front of
                    # the first capture
                    step.optG.step_count, step.optD.step_count = self._counts
                    if step.optG.pack is not None:
                        step.optG.pack.invalidate()
                    if step.optD.pack is not None:
                        step.optD.pack.invalidate()
                    self._refresh_d()
            if self.graph is None:
                self._capture_three(step, mode)
                self.ddp_mode = "three"
        # capture ran the Python side once without executing kernels: undo its optimiser step counts
        step.optG.step_count, step.optD.step_count = self._counts
        self.last = self.out

    def _capture_three(self, step: GANTrainStep, mode) -> None:
        step._d_handle = step._g_handle = None            # nothing to await inside the captures
        step.join_side_after_phase_d = True
        # THREE graphs: phase_d | phase_g | phase_opt around the two exchanges.  The side branch (multi-TD + frozen
        # encoder, forward and backward: 2.5 ms, the longest arm of the step) would make the first graph end 1.3 ms after
        # the discriminator's gradients are ready if it ran there as a whole (round 2), so it is cut in two: forward +
        # losses in the first graph beside the discriminator step, backward in the second beside the D update and the
        # generator-side discriminator passes (GANTrainStep.split_side).  Measured on one GPU with a 1-rank RCCL group
        # (scripts/ddp_graph_timing.py): see DESIGN.md.  (Tried and dropped: the branch as a FOURTH graph replayed on its own
        # stream across the exchange -- 8.5 ms/step: streams map onto 4 hardware queues, the replay stream shared one with
        # the discriminator graph's internal fork streams and the two ran one after the other.)
        step.split_side = os.environ.get("STEGAN_SPLIT_SIDE", "1") != "0"
        g1, g2, g3 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(g1, **mode):
                step.phase_d(self.static)
            with torch.cuda.graph(g2, pool=g1.pool(), **mode):
                step.phase_g()
            with torch.cuda.graph(g3, pool=g1.pool(), **mode):
                self.out = step.phase_opt()
        except BaseException:
            _leak_failed_graphs(g1, g2, g3)
            step.abort()        # (the phases are called directly here: GANTrainStep.__call__'s own cleanup does not run)
            raise
        self.graphs = (g1, g2, g3)

    def _weights_signature(self):
        return sum(p._version for p in self.step.netD.parameters()) + sum(b._version for b in self.step.netD.buffers())

    def _refresh_d(self):
        """Bring the discriminator's prepared weight copies up to date, eagerly, on the current stream."""
        netD = self.step.netD
        if hasattr(netD, "_prepare_weights"):
            netD._prepare_weights()
        else:
            for m in netD.modules():
                prep = getattr(m, "prepared", None)
                if prep is not None:
                    prep()
        self._d_sig = self._weights_signature()

    def _stage(self, opt: FusedAdamW):
        opt.step_count += 1          # host mirror of the device-side counter the captured advance kernel increments
        opt.sync_lr()

    def load_batch(self, batch_dict: Dict[str, torch.Tensor]) -> None:
        """Copy a batch into the graph's static input buffers (stream ordered, no step)."""
        for k, v in batch_dict.items():
            if torch.is_tensor(v) and v is not self.static[k]:
                self.static[k].copy_(v, non_blocking=True)

    def __call__(self, batch_dict: Optional[Dict[str, torch.Tensor]] = None) -> Dict[str, torch.Tensor]:
        """``batch_dict`` is copied into the graph's static input buffers; None (or ``self.static`` itself): the batch is
        already there -- a producer that writes into ``self.static`` (the batch-assembly kernel, a benchmark whose
        synthetic batch is resident: ``load_batch``) saves the five device-to-device copies in front of every replay."""
        if batch_dict is not None and batch_dict is not self.static:
            self.load_batch(batch_dict)
        self._stage(self.step.optD)
        self._stage(self.step.optG)
        if self._weights_signature() != self._d_sig or self.step.optD.pack is None or \
                self.step.optD.pack.prepared_version != (self.step.optD.pack.version, ops.COMPUTE_DTYPE):
            self._refresh_d()                  # load_state_dict / eager code touched the discriminator since the last replay
        if self.graphs is None:
            self.graph.replay()
            self._after_replay()
            return self.out
        step = self.step
        g1, g2, g3 = self.graphs
        g1.replay()
        step.exchange_d()                      # eager: all-reduce of D's gradient arena on the communication stream
        step.comm.wait(step._d_handle); step._d_handle = None
        g2.replay()
        step.exchange_g()
        step.comm.wait(step._g_handle); step._g_handle = None
        g3.replay()
        self._after_replay()
        return self.out

    def close(self) -> None:
        """Ordered teardown: drain the device, destroy the captured graphs (and with them the shared memory pool), drop the
        static batch and the step's communication stream.  Called by ``parallel.shutdown`` before the process group goes."""
        torch.cuda.synchronize()
        self.graph = None
        self.graphs = None
        self.out = self.last = None
        self.static = {}
        # eager users of the optimisers get torch.optim semantics back (.grad kept until zero_grad())
        self.step.optG.set_zero_in_step(False)
        self.step.optD.set_zero_in_step(False)
        self.step.close()
        torch.cuda.synchronize()

    def _after_replay(self):
        """The replayed optimiser kernels changed the weights behind the host's back: mark every prepared weight copy stale
        for EAGER users of the networks (validation, sampling).  The captured step re-derives its own copies at its start."""
        self.step.optG.flat.epoch += 1
        if self.step.optG.pack is not None:
            self.step.optG.pack.invalidate()
        # (the discriminator's copies were re-derived inside the step after its update: still current)

    read_stats = staticmethod(GANTrainStep.read_stats)


class _NullWriter:
    def add_scalar(self, *a, **k):
        pass

    add_figure = add_scalar


def _summary_writer(model_directory: Path):
    try:
        from torch.utils.tensorboard import SummaryWriter
        return SummaryWriter(str(model_directory.absolute()))
    except Exception:                     # tensorboard is not installed in every image
        logging.warning("tensorboard unavailable: scalars are logged to the text log only")
        return _NullWriter()


def make_optimizers(netG, netD):
    """Flat arenas + fused AdamW + multi-tensor conv handling (ConvPack) for both networks."""
    from .optim import ConvPack
    fg, fd = FlatParams(netG), FlatParams(netD)
    return FusedAdamW(fg, pack=ConvPack(netG, fg)), FusedAdamW(fd, pack=ConvPack(netD, fd))


def build_training_objects(cfg, device, emg_enc_ckpt):
    netG = init_emg_generator(cfg).to(device)
    netD = init_emg_discriminators(cfg).to(device)
    emg_encoder = load_emg_encoder(cfg, device, emg_enc_ckpt)
    emg_encoder.eval().to(device)
    multi_td_loss = MultiTimeDomainFeatureLoss(cfg.data.num_emg_channels).to(device)
    emg_encoder_loss = EMGEncoderLoss(emg_encoder).to(device)
    optG, optD = make_optimizers(netG, netD)
    return netG, netD, emg_encoder, multi_td_loss, emg_encoder_loss, optG, optD


def dist_info():
    """(rank, world_size) of this process; (0, 1) when ``torch.distributed`` is not initialised."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def init_distributed_from_env() -> Optional[torch.device]:
    """One process per GPU under ``torch.distributed.run``: picks ``cuda:LOCAL_RANK`` and joins the RCCL group BEFORE
    anything else touches the GPU.  Returns the device, or None for a plain single-process start."""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1 or (dist.is_available() and dist.is_initialized()):
        return None
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: the only mode the host driver supports
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist.init_process_group("nccl", device_id=device)
    return device


def train(cfg, model_directory: Path, checkpoint: Path, torch_device: str, debug: bool, emg_enc_ckpt: Path = None,
          loaders=None):
    """``cfg.train.mixed_precision`` selects the compute mode for the duration of the run (see the module docstring);
    the mode in force before the call is restored afterwards.  Everything else: :func:`_train`."""
    previous = ops.COMPUTE_DTYPE
    # cfg.train.mixed_precision is the reference's switch (fp16 autocast + GradScaler, train.py:151,181,204); which 2-byte
    # operand type it means here: cfg.train.mixed_precision_dtype = "bf16" (default: BASELINE configs 3 / 4, no loss scaling
    # needed) or "f16" (BASELINE config 5: fp16 operands + the device-side GradScaler)
    half = str(cfg.train.get("mixed_precision_dtype", "bf16"))
    if half not in ("bf16", "f16"):
        raise ValueError(f"cfg.train.mixed_precision_dtype must be 'bf16' or 'f16', got {half!r}")
    ops.set_compute_dtype(half if bool(cfg.train.get("mixed_precision", False)) else "f32")
    try:
        return _train(cfg, model_directory, checkpoint, torch_device, debug, emg_enc_ckpt, loaders)
    finally:
        # ordered release of what the run captured (graphs + their pool, static batches, the communication stream): the
        # caller may destroy the process group next (main(), parallel.shutdown)
        for st in reversed(_LIVE_STEPS):
            st.close()
        _LIVE_STEPS.clear()
        ops.set_compute_dtype(previous)


_LIVE_STEPS: list = []          # the (graphed) steps of the run in progress, closed by train() on the way out


def _train(cfg, model_directory: Path, checkpoint: Path, torch_device: str, debug: bool, emg_enc_ckpt: Path = None,
           loaders=None):
    """Full training run -- ste_gan/train.py:39-499.  ``loaders`` = (train, valid, test) DataLoaders whose batches
    are dicts keyed by ``DataType`` (data/collate.py:54-61); when omitted they come from ``loaders_via_config``.

    Data parallel (``torch.distributed`` initialised, see :func:`main`): every rank holds a replica -- rank 0's initial
    weights are broadcast --, walks its own shard of each epoch's batch plan and joins the two gradient all-reduces of
    the step; validation, TensorBoard scalars and every file under ``model_directory`` belong to rank 0, the other
    ranks wait at a barrier.  Caller-supplied ``loaders`` must already be sharded per rank."""
    device = torch.device(torch_device)
    if device.type != "cuda":
        raise RuntimeError("ste_gan_amd.train runs on MI355X (cuda device) only; there is no CPU fallback")
    rank, world = dist_info()
    is_main = rank == 0
    logging.info(f"Convolution operand type: {ops.COMPUTE_DTYPE} (cfg.train.mixed_precision)")
    logging.info("Initializing Models")
    netG, netD, emg_encoder, multi_td_loss, emg_encoder_loss, optG, optD = build_training_objects(cfg, device, emg_enc_ckpt)
    writer = _summary_writer(model_directory) if is_main else _NullWriter()
    if checkpoint is not None:
        logging.info(f"Loading checkpoint: {checkpoint}")
        netG, netD, optG, optD, start_epoch, steps = load_latest_checkpoint(checkpoint, device, netG, netD, optG, optD)
    else:
        start_epoch, steps = -1, 0
    if world > 1:                            # identical replicas whatever each process drew at initialisation
        for opt in (optG, optD):
            dist.broadcast(opt.flat.flat, src=0)
            if opt.pack is not None:
                opt.pack.invalidate()
        for buf in itertools.chain(netG.buffers(), netD.buffers()):      # spectral-norm u / v
            dist.broadcast(buf, src=0)
    scheduler_g = ExponentialLR(optG, gamma=.999, last_epoch=start_epoch if checkpoint is not None else -1)
    scheduler_d = ExponentialLR(optD, gamma=.999, last_epoch=start_epoch if checkpoint is not None else -1)

    np.random.seed(cfg.train.random_seed)
    torch.cuda.manual_seed(cfg.train.random_seed)
    torch.manual_seed(cfg.train.random_seed)
    random.seed(cfg.train.random_seed)
    if loaders is None:
        try:
            from .data.loader import loaders_via_config
        except ImportError as e:
            raise NotImplementedError("dataset loading (ste_gan/data) is outside the accelerated path this round: pass "
                                      "loaders=(train, valid, test) yielding DataType-keyed batch dicts") from e
        loaders = loaders_via_config(cfg, device=device, rank=rank, world_size=world)
    train_loader, valid_loader, test_loader = loaders
    if is_main and hasattr(train_loader, "save_session_and_speaking_mode_mapping_json") and \
            getattr(train_loader, "session_id_to_idx", None) is not None:
        train_loader.save_session_and_speaking_mode_mapping_json(model_directory)          # train.py:121-122

    step_fn = GANTrainStep(cfg, netG, netD, emg_encoder_loss, multi_td_loss, optG, optD)
    _LIVE_STEPS.append(step_fn)
    # hipGraph replay per batch SHAPE: an eager step is ~800 launches and bound by the host (22.7 ms against 8.8 ms of GPU
    # time at B = 64).  The first batch of a shape runs eagerly -- that IS its training step, and it warms every lazy
    # allocation -- then the step is captured (capture executes nothing) and later batches of that shape replay it.
    graphed: Dict[tuple, GraphedTrainStep] = {}
    use_graphs = torch.device(device).type == "cuda" and os.environ.get("STEGAN_TRAIN_GRAPH", "1") != "0"
    speech_feature_type = cfg.model.speech_feature_type
    crop = crop_to_hop(cfg.train.chunk_size)
    if crop != cfg.train.chunk_size:
        logging.warning(f"chunk_size {cfg.train.chunk_size} is not a multiple of {K.HOPSIZE}: real EMG is cropped to {crop}")
    log_start = time.time()
    best_su_loss = np.inf
    tags = ("discriminator", "multi_td", "speech_unit", "phoneme", "waveform", "feature_matching", "generator")
    for epoch in itertools.count(start_epoch):
        logging.info(f"Starting epoch {epoch + 1}")
        acc = dict(n=0, ok=0, sil=0, ok_ns=0)
        for iterno, batch_dict in enumerate(train_loader):
            batch = {k: (v.to(device, non_blocking=True) if torch.is_tensor(v) else v) for k, v in batch_dict.items()}
            if batch[DataType.REAL_EMG].shape[1] > crop >= K.HOPSIZE:
                batch[DataType.REAL_EMG] = batch[DataType.REAL_EMG][:, :crop].contiguous()
            key = tuple((str(k), tuple(v.shape)) for k, v in batch.items() if torch.is_tensor(v))
            stats = None
            if not use_graphs:
                out = step_fn(batch)
            elif key in graphed:
                out = graphed[key](batch)
            else:
                out = step_fn(batch)                                   # eager: this batch's step, and the warm-up
                stats = step_fn.read_stats(out)                        # (read before a capture runs the Python side again)
                if len(graphed) < 4:                                   # more distinct shapes than that: the rest stay eager
                    graphed[key] = GraphedTrainStep(step_fn, batch, warmup=0)
                    _LIVE_STEPS.append(graphed[key])
            if stats is None:
                stats = step_fn.read_stats(out)
            for tag in tags:
                if tag in stats:
                    writer.add_scalar(f"train_loss/{tag}", stats[tag], steps)
            if "loss_scale" in stats:                                  # fp16 mode: the GradScaler's scale (not in the reference's logs)
                writer.add_scalar("train/loss_scale", stats["loss_scale"], steps)
            if "num_phones" in stats:
                acc["n"] += stats["num_phones"]; acc["ok"] += stats["num_correct_phones"]
                acc["sil"] += stats["num_silence_phones"]; acc["ok_ns"] += stats["num_correct_phones_no_silence"]
                writer.add_scalar("train_loss/phoneme_acc_batch",
                                  phoneme_accuracy(stats["num_phones"], stats["num_correct_phones"]), steps)
                writer.add_scalar("train_loss/phoneme_acc_batch_no_sil",
                                  phoneme_accuracy_no_silence(stats["num_phones"], stats["num_correct_phones_no_silence"],
                                                              stats["num_silence_phones"]), steps)
            if steps % cfg.train.interval_log == 0:
                pa = phoneme_accuracy(acc["n"], acc["ok"])
                pa_ns = phoneme_accuracy_no_silence(acc["n"], acc["ok_ns"], acc["sil"])
                logging.info(f"Epoch {epoch} ({iterno}/{len(train_loader)}) | Steps {steps} | ms/batch "
                             f"{1e3 * (time.time() - log_start) / cfg.train.interval_log:5.2f} | Train Loss: "
                             f"{stats['generator']:.4f} | Ph. Acc. (Avg.) {pa:.2f} | Ph. Acc. Avg. (No Sil) {pa_ns:.2f}")
                writer.add_scalar("train_loss/phoneme_accuracy_avg", pa, steps)
                writer.add_scalar("train_loss/phoneme_accuracy_avg_no_sil", pa_ns, steps)
                log_start = time.time()

            if steps % cfg.train.interval_valid == 0:
                if is_main:                  # replicas are identical: rank 0 validates the whole partition
                    val = validate(netG, multi_td_loss, emg_encoder_loss, valid_loader, speech_feature_type, device)
                    for k, v in val.items():
                        writer.add_scalar(f"val_loss/{k}", v, steps)
                    logging.info(" | ".join(f"val {k}: {v:.5f}" for k, v in val.items()))
                    if val["speech_unit"] < best_su_loss:
                        best_su_loss = val["speech_unit"]
                        torch.save(netG.state_dict(), model_directory / "best_netG.pt")
                        torch.save(netD.state_dict(), model_directory / "best_netD.pt")
                _barrier(world)

            if steps > 0 and steps % cfg.train.interval_save == 0:                      # train.py:421
                if is_main:
                    save_checkpoint(model_directory, f"{steps:08d}", netG, netD, optG, optD, epoch, steps)
                _barrier(world)
            if steps >= cfg.train.max_steps:
                if is_main:
                    save_checkpoint(model_directory, "final", netG, netD, optG, optD, epoch, steps)
                    with open(model_directory / ".done", "w") as fp:                    # train.py:463-466
                        fp.write(f"done: {time.time()}")
                _barrier(world)
                return
            steps += 1
            if debug:
                break
        if cfg.train.loss_adversarial:                                                  # train.py:470-472
            scheduler_d.step()
        scheduler_g.step()
        if epoch % 5 == 0:
            if is_main:
                save_checkpoint(model_directory, "last", netG, netD, optG, optD, epoch, steps)
            _barrier(world)
        if debug:
            return


def _barrier(world: int) -> None:
    if world > 1:
        dist.barrier()


def save_checkpoint(model_directory: Path, tag: str, netG, netD, optG, optD, epoch: int, steps: int):
    """File naming of train.py:421-457,478-494: netG-*.pt, netD-*.pt, checkpoint-*.pt = {epoch, steps, optG, optD}."""
    torch.save(netG.state_dict(), model_directory / f"netG-{tag}.pt")
    torch.save(netD.state_dict(), model_directory / f"netD-{tag}.pt")
    torch.save({"epoch": epoch, "steps": steps, "optG": optG.state_dict(), "optD": optD.state_dict()},
               model_directory / f"checkpoint-{tag}.pt")


@torch.no_grad()
def validate(netG, multi_td_loss, emg_encoder_loss, valid_loader, speech_feature_type, device) -> Dict[str, float]:
    """Validation pass of train.py:292-378 (fp32 here; the reference autocasts to fp16 on CUDA)."""
    from . import ops
    netG.eval()
    td, su, ph, wav = [], [], [], []
    n = ok = sil = ok_ns = 0
    for feat_batch in valid_loader:
        x_t = feat_batch[DataType.REAL_EMG].to(device)
        s_t = feat_batch[speech_feature_type].to(device)
        su_t = s_t if speech_feature_type == DataType.SPEECH_UNITS else feat_batch[DataType.SPEECH_UNITS].to(device)
        x_pred = netG(s_t, feat_batch[DataType.SESSION_INDEX].to(device), feat_batch[DataType.SPEAKING_MODE_INDEX].to(device))
        T = min(x_pred.shape[1], x_t.shape[1])
        x_t, x_pred = x_t[:, :T].contiguous(), x_pred[:, :T].contiguous()
        eo = emg_encoder_loss(x_pred, su_t, feat_batch[DataType.PHONEMES].to(device))
        scal = torch.stack([ops.mse_loss(x_pred, x_t), multi_td_loss(x_t, x_pred), eo.speech_unit_loss, eo.phoneme_loss]).tolist()
        wav.append(scal[0]); td.append(scal[1]); su.append(scal[2]); ph.append(scal[3])
        n += eo.num_phones; ok += eo.num_correct_phones; sil += eo.num_silence_phones; ok_ns += eo.num_correct_phones_no_silence
    return {"speech_unit": float(mean_error(su)), "multi_td": float(mean_error(td)), "phoneme": float(mean_error(ph)),
            "phoneme_accuracy_avg": phoneme_accuracy(n, ok),
            "phoneme_accuracy_avg_no_sil": phoneme_accuracy_no_silence(n, ok_ns, sil), "waveform": float(mean_error(wav))}


def main(cfg, continue_run: bool, debug: bool, emg_enc_ckpt: Path, **kwargs):
    """Entry point -- ste_gan/train.py:503-546 (same keyword arguments: ``main(**vars(args))``)."""
    print(f"Data root: {cfg.data.dataset_root}\ncontinue_run: {continue_run}\nDebug (argparse): {debug}")
    debug = debug or bool(cfg.train.debug)
    ddp_device = init_distributed_from_env()              # torchrun: cuda:LOCAL_RANK + RCCL group, before any GPU call
    rank, world = dist_info()
    output_directory = Path(cfg.model_base_dir) / create_ste_gan_model_name(cfg, add_timestamp=False, debug=debug)
    checkpoint = output_directory if (output_directory.exists() and continue_run) else None
    if world > 1:                                          # every rank must agree on resume-or-not before rank 0 mkdirs
        flag = torch.tensor([1 if checkpoint is not None else 0], device=ddp_device)
        dist.broadcast(flag, src=0)
        checkpoint = output_directory if int(flag.item()) else None
    output_directory.mkdir(exist_ok=True, parents=True)
    print(f"Output directory: {output_directory}")
    done_file = output_directory / ".done"
    if done_file.exists():
        logging.warning(f"Exiting training script as '.done' file exists: {done_file.absolute()}")
        sys.exit()
    config_file = output_directory / "config.yaml"
    if rank == 0 and not config_file.exists():
        import yaml
        with open(config_file, "w") as fp:
            yaml.safe_dump(_plain(cfg), fp)
    logging.getLogger().setLevel(logging.INFO)
    if rank == 0:
        logging.getLogger().addHandler(logging.FileHandler(str((output_directory / "log.txt").absolute())))
    device = ddp_device if ddp_device is not None else torch.device("cuda" if torch.cuda.is_available() else "cpu")
    try:
        train(cfg, output_directory, checkpoint, device, debug, emg_enc_ckpt)
    finally:
        if ddp_device is not None:
            from .parallel import shutdown
            shutdown()                         # train() has released its graphs / comm stream: drain, THEN destroy the group


def _plain(d):
    return {k: _plain(v) for k, v in d.items()} if isinstance(d, dict) else d


def parse_args(argv=None):
    """CLI of ste_gan/train.py:549-574."""
    parser = argparse.ArgumentParser()
    parser.add_argument("--config", type=str, default="configs/ste_gan_base_gantts.yaml",
                        help="The main training configuration for this run.")
    parser.add_argument("--data", type=str, default="configs/data/gaddy_and_klein_corpus.yaml",
                        help="A path to a data configuration file.")
    parser.add_argument("--emg_enc_cfg", type=str, default="configs/emg_encoder/conv_transformer.yaml",
                        help="A path to an EMG encoder configuration file.")
    parser.add_argument("--emg_enc_ckpt", type=str,
                        default="exp/emg_encoder/EMGEncoderTransformer_voiced_only__seq_len__200__data_gaddy_complete/"
                                "best_val_loss_model.pt",
                        help="Checkpoint of the pre-trained EMG encoder matching --emg_enc_cfg.")
    parser.add_argument("--checkpoint", type=Path, help="Optional checkpoint to start training from")
    parser.add_argument("--continue_run", action="store_true", help="Whether to continue training")
    parser.add_argument("--debug", action="store_true", help="Run the training script in debug mode.")
    parser = add_eval_hyperparams_to_parser(parser)
    return parser.parse_args(argv)


if __name__ == "__main__":
    _args = parse_args()
    _args.cfg = load_config(_args)
    main(**vars(_args))
