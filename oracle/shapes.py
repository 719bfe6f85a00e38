"""Reference state-dict key/shape layouts, restated -- TEST INFRASTRUCTURE (see stegan_oracle.py).

tests/golden/make_golden.py asserts these equal the ``state_dict()`` of the reference's own
modules (key order included); the product modules are then tested against them.
Layout source: SURVEY.md section 5.4; reference constructors cited per function.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, Tuple

Shapes = "OrderedDict[str, Tuple[int, ...]]"


def _wn(d, pre, cout, cin, *k):
    d[pre + ".bias"] = (cout,)
    d[pre + ".weight_g"] = (cout, 1) + (1,) * len(k)
    d[pre + ".weight_v"] = (cout, cin) + tuple(k)


def _sn(d, pre, cout, cin_g, k):
    d[pre + ".bias"] = (cout,)
    d[pre + ".weight_orig"] = (cout, cin_g, k)
    d[pre + ".weight_u"] = (cout,)
    d[pre + ".weight_v"] = (cin_g * k,)


def generator_shapes(speech_feature_type: str = "SPEECH_UNITS", channels: int = 768,
                     num_sessions: int = 17, num_emg_channels: int = 8, embedding_dim: int = 64):
    """EMGGeneratorGanTTS.__init__ -- ste_gan/models/generator.py:83-137, layers/conv.py:31-79."""
    d = OrderedDict()
    in_dim = (256 if speech_feature_type == "SPEECH_UNITS" else 25) + embedding_dim
    d["session_embeddings.weight"] = (num_sessions, embedding_dim)
    _wn(d, "gblocks.0", channels, in_dim, 1)
    c = channels
    last = 2 if speech_feature_type == "SPEECH_UNITS" else 1
    plan = [(c, c, 1), (c, c, 1), (c, c // 2, 2), (c // 2, c // 2, 2), (c // 2, c // 2, 2),
            (c // 2, c // 4, last), (c // 4, c // 4, 1), (c // 4, c // 4, 1)]
    for i, (ci, co, up) in enumerate(plan):
        o = 1 if up > 1 else 0
        pre = f"gblocks.{i + 1}"
        _wn(d, f"{pre}.conv1.{1 + o}", co, ci, 3)
        _wn(d, f"{pre}.conv1.{3 + o}", co, co, 3)
        _wn(d, f"{pre}.res1.{o}", co, ci, 1)
        _wn(d, f"{pre}.conv2.1", co, co, 3)
        _wn(d, f"{pre}.conv2.3", co, co, 3)
    _wn(d, "last_conv.1", num_emg_channels, c // 4, 3)
    return d


def discriminator_shapes(small: bool = True, num_emg_channels: int = 8):
    """DiscriminatorSmall / Discriminator -- ste_gan/models/discriminator.py:19-191."""
    d = OrderedDict()
    C = num_emg_channels
    for i in range(5):
        pre = f"multi_pooled_disc.{i}"
        chans = [C, 32, 256, 512] if small else [C, 32, 128, 512, 1024, 1024]
        k = 3 if small else 5
        for j in range(len(chans) - 1):
            _wn(d, f"{pre}.layers.{j}", chans[j + 1], chans[j], k, 1)
        _wn(d, f"{pre}.output", 1, chans[-1], 3, 1)
    if small:
        body = [(C, 128, 15, 1), (128, 256, 37, 4), (256, 512, 37, 16), (512, 1024, 5, 1)]
    else:
        body = [(C, 128, 15, 1), (128, 128, 41, 4), (128, 256, 41, 16), (256, 512, 41, 16),
                (512, 1024, 41, 16), (1024, 1024, 41, 16), (1024, 1024, 5, 1)]
    for i in range(3):
        pre = f"multi_scale_disc.{i}"
        for j, (ci, co, k, g) in enumerate(body):
            if i == 0:
                _sn(d, f"{pre}.layers.{j}", co, ci // g, k)
            else:
                _wn(d, f"{pre}.layers.{j}", co, ci // g, k)
        _wn(d, f"{pre}.output", 1, body[-1][1], 3)
    return d


def encoder_shapes(model_size: int = 768, num_extra_res_blocks: int = 3, num_transformer_layers: int = 6,
                   num_ins: int = 8, num_outs: int = 256, num_aux_outs: int = 48, nhead: int = 8,
                   dim_feedforward: int = 3072, max_rel: int = 100):
    """EMGEncoderTransformer -- models/emg_encoder.py:39-68; layers/conv.py:108-120; layers/transformer.py:29-42,63-86,134-166."""
    d = OrderedDict()
    M = model_size

    def bn(pre):
        d[pre + ".weight"] = (M,)
        d[pre + ".bias"] = (M,)
        d[pre + ".running_mean"] = (M,)
        d[pre + ".running_var"] = (M,)
        d[pre + ".num_batches_tracked"] = ()

    for i in range(1 + num_extra_res_blocks):
        cin = num_ins if i == 0 else M
        pre = f"conv_blocks.{i}"
        d[pre + ".conv1.weight"] = (M, cin, 3)
        d[pre + ".conv1.bias"] = (M,)
        bn(pre + ".bn1")
        d[pre + ".conv2.weight"] = (M, M, 3)
        d[pre + ".conv2.bias"] = (M,)
        bn(pre + ".bn2")
        d[pre + ".residual_path.weight"] = (M, cin, 1)
        d[pre + ".residual_path.bias"] = (M,)
        bn(pre + ".res_norm")
    d["w_raw_in.weight"] = (M, M)
    d["w_raw_in.bias"] = (M,)
    dq = M // nhead
    for i in range(num_transformer_layers):
        pre = f"transformer.layers.{i}"
        for n in "qkv":
            d[f"{pre}.self_attn.w_{n}"] = (nhead, M, dq)
        d[f"{pre}.self_attn.w_o"] = (nhead, dq, M)
        d[f"{pre}.self_attn.relative_positional.embeddings"] = (nhead, 2 * max_rel - 1, dq, 1)
        d[f"{pre}.linear1.weight"] = (dim_feedforward, M)
        d[f"{pre}.linear1.bias"] = (dim_feedforward,)
        d[f"{pre}.linear2.weight"] = (M, dim_feedforward)
        d[f"{pre}.linear2.bias"] = (M,)
        d[f"{pre}.norm1.weight"] = (M,)
        d[f"{pre}.norm1.bias"] = (M,)
        d[f"{pre}.norm2.weight"] = (M,)
        d[f"{pre}.norm2.bias"] = (M,)
    d["w_out.weight"] = (num_outs, M)
    d["w_out.bias"] = (num_outs,)
    d["w_aux.weight"] = (num_aux_outs, M)
    d["w_aux.bias"] = (num_aux_outs,)
    return d
