"""Deterministic counter-based parameter fill -- TEST INFRASTRUCTURE (see stegan_oracle.py header).

The reference's real checkpoints are 94 MB (G) / 47 MB (D) / 234 MB (encoder) and are not
available offline, so golden fixtures use weights that are a pure function of
(tensor name, shape): a splitmix64 hash of ``crc32(name) + element index`` mapped to
U(-1,1) and scaled by a rule chosen from the key suffix.  The same function is used when
generating goldens from the reference modules (tests/golden/make_golden.py, via
``load_state_dict``) and when testing the oracle / the HIP path, so only outputs are stored.
"""
from __future__ import annotations

import zlib
from typing import Dict, Iterable, Tuple

import numpy as np
import torch


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15))
        z = x
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def uniform(name: str, shape: Tuple[int, ...], salt: int = 0) -> np.ndarray:
    """U(-1,1) float64 array, a pure function of (name, salt, index)."""
    n = int(np.prod(shape)) if len(shape) else 1
    seed = np.uint64(zlib.crc32(name.encode()) + (salt << 32))
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) * np.uint64(0xD1342543DE82EF95) + seed
    bits = _splitmix64(idx) >> np.uint64(11)                      # 53 random bits
    u = bits.astype(np.float64) * (1.0 / (1 << 53))
    return (2.0 * u - 1.0).reshape(shape)


def fill_tensor(key: str, shape: Tuple[int, ...], all_shapes: Dict[str, Tuple[int, ...]]) -> np.ndarray:
    """Scale rule per reference state-dict key suffix (shapes: SURVEY.md section 5.4)."""
    u = uniform(key, shape)
    leaf = key.rsplit(".", 1)[-1]
    base = key[: -len(leaf)]
    if leaf == "num_batches_tracked":
        return np.zeros(shape, np.int64)
    if leaf == "weight_g":                                        # weight-norm gain, positive
        return (0.6 * (1.0 + 0.25 * u)).astype(np.float32)
    if leaf in ("weight_u",) or (leaf == "weight_v" and base + "weight_orig" in all_shapes):
        v = u / np.linalg.norm(u)                                 # spectral-norm unit vectors
        return v.astype(np.float32)
    if leaf == "running_var":
        return (1.0 + 0.3 * np.abs(u)).astype(np.float32)
    if leaf == "running_mean":
        return (0.1 * u).astype(np.float32)
    parent = base.split(".")[-2] if "." in base else ""
    if "norm" in parent or parent.startswith("bn"):
        # LayerNorm / BatchNorm affine
        return ((1.0 + 0.1 * u) if leaf == "weight" else 0.1 * u).astype(np.float32)
    if leaf == "embeddings":                                      # rel-pos table (H, 2R-1, D, 1)
        return (u * (3.0 / shape[-2]) ** 0.5).astype(np.float32)
    if key == "session_embeddings.weight" or key == "speaking_mode_embeddings.weight":
        return (1.7 * u).astype(np.float32)
    if leaf in ("w_q", "w_k", "w_v", "w_o"):                      # (H, in, out) xavier-ish
        fan = shape[1] * shape[2] + shape[0] * shape[2]
        return (u * (6.0 / fan) ** 0.5 * 2.0).astype(np.float32)
    if leaf == "bias":
        w = all_shapes.get(base + "weight_v") or all_shapes.get(base + "weight_orig") or all_shapes.get(base + "weight")
        fan_in = int(np.prod(w[1:])) if w is not None else shape[0]
        return (u / fan_in ** 0.5).astype(np.float32)
    # conv / linear weights (weight, weight_v under weight-norm, weight_orig under spectral norm)
    fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
    return (u * (3.0 / fan_in) ** 0.5).astype(np.float32)


def fill_state_dict(shapes: Dict[str, Tuple[int, ...]]) -> Dict[str, torch.Tensor]:
    shapes = {k: tuple(v) for k, v in shapes.items()}
    return {k: torch.from_numpy(np.ascontiguousarray(fill_tensor(k, s, shapes))) for k, s in shapes.items()}


def shapes_of(sd: Dict[str, torch.Tensor]) -> Dict[str, Tuple[int, ...]]:
    return {k: tuple(v.shape) for k, v in sd.items()}


def det_input(name: str, shape: Iterable[int], kind: str = "normal") -> torch.Tensor:
    """Deterministic synthetic inputs: 'normal' ~ approx N(0,1) (sum of 3 uniforms), 'tanh' = tanh(normal)."""
    shape = tuple(shape)
    z = (uniform(name, shape, 1) + uniform(name, shape, 2) + uniform(name, shape, 3))
    if kind == "tanh":
        z = np.tanh(z)
    return torch.from_numpy(z.astype(np.float32))


def det_ints(name: str, shape: Iterable[int], high: int) -> torch.Tensor:
    u = (uniform(name, tuple(shape), 7) + 1.0) * 0.5
    return torch.from_numpy(np.minimum((u * high).astype(np.int64), high - 1))
