"""CPU tests of the host logic behind data-parallel training (no GPU, no collectives needed).

* ``data.loader.plan_epoch``: the ranks' shards of one epoch together are exactly the single-process plan of a global
  batch of world * batch rows; every rank takes the same number of steps; world 1 keeps the reference's RNG order.
* ``optim.ExponentialLR`` follows ``torch.optim.lr_scheduler.ExponentialLR`` step for step, also across a resume.
* the session / speaking-mode id maps are written in the reference's JSON form (emg_dataset.py:202-217).
"""
import json
import types

import pytest
import torch

from ste_gan_amd.data.loader import DeviceLoader, plan_epoch
from ste_gan_amd.optim import ExponentialLR


@pytest.mark.parametrize("n,B,world", [(37, 4, 2), (64, 4, 4), (10, 4, 3), (9, 2, 2), (5, 4, 2)])
def test_plan_epoch_shards_partition_the_global_plan(n, B, world):
    frames = [30 + (7 * i) % 23 for i in range(n)]
    torch.manual_seed(123)
    full_order, full_starts = plan_epoch(frames, "train", B * world, 12)
    shards = []
    for r in range(world):
        torch.manual_seed(123)                                  # every rank seeds identically (train())
        shards.append(plan_epoch(frames, "train", B, 12, r, world))
    steps = {-(-len(o) // B) for o, _ in shards}
    assert len(steps) == 1                                      # same number of steps on every rank
    assert all(len(o) == len(shards[0][0]) for o, _ in shards)
    gb = B * world
    seen = []
    for i in range(0, n, gb):                                   # global batch j = concatenation of the ranks' rows of step j
        rows = min(gb, n - i)
        per = B if rows == gb else rows // world
        j = i // gb
        for r in range(world):
            o, s = shards[r]
            got_o, got_s = o[j * B:j * B + per], s[j * B:j * B + per]
            assert got_o == full_order[i + r * per:i + (r + 1) * per]
            assert got_s == full_starts[i + r * per:i + (r + 1) * per]
            seen += got_o
    assert len(set(seen)) == len(seen)                          # no utterance served twice in an epoch
    for o, s in shards:                                         # windows fit their utterances
        assert all(0 <= st <= max(0, frames[u] - 12) for u, st in zip(o, s))


def test_plan_epoch_world1_is_the_reference_rng_order():
    frames = [20, 40, 13, 33, 50]
    torch.manual_seed(7)
    order, starts = plan_epoch(frames, "train", 2, 12)
    torch.manual_seed(7)
    want_order = torch.randperm(5).tolist()                     # RandomSampler, then one randint per utterance per batch
    want_starts = [int(torch.randint(0, 1 + max(0, frames[u] - 12), (1,)).item()) for u in want_order]
    assert (order, starts) == (want_order, want_starts)
    assert plan_epoch(frames, "valid", 2, 12) == ([0, 1, 2, 3, 4], [0] * 5)


class _Opt:
    """What ExponentialLR touches of FusedAdamW."""
    def __init__(self, lr, initial_lr=None):
        self.param_groups = [dict(lr=lr)]
        if initial_lr is not None:
            self.param_groups[0]["initial_lr"] = initial_lr


def test_exponential_lr_matches_torch_fresh_and_resumed():
    p = torch.nn.Parameter(torch.zeros(1))
    topt = torch.optim.AdamW([p], lr=2e-4)
    tsch = torch.optim.lr_scheduler.ExponentialLR(topt, gamma=0.999, last_epoch=-1)
    mine = _Opt(2e-4)
    msch = ExponentialLR(mine, gamma=0.999, last_epoch=-1)
    for _ in range(5):
        assert mine.param_groups[0]["lr"] == pytest.approx(topt.param_groups[0]["lr"], rel=1e-15)
        topt.step(); tsch.step(); msch.step()
    # resume at epoch 4 from the saved optimiser state, as train.py:86-104 does.  The reference pins torch 2.0.1
    # (requirements.txt:97), whose ExponentialLR.get_lr is `group["lr"] if last_epoch == 0 else group["lr"] * gamma`: the
    # constructor's own step() on resume multiplies the LOADED lr by gamma once (newer torch releases skip that
    # multiplication through an `_is_initial` flag; the pinned behaviour is the one mirrored).
    g = topt.state_dict()["param_groups"][0]
    loaded = g["lr"]
    assert loaded == pytest.approx(2e-4 * 0.999 ** 5, rel=1e-12)
    mine2 = _Opt(g["lr"], g["initial_lr"])
    msch2 = ExponentialLR(mine2, gamma=0.999, last_epoch=4)
    assert msch2.last_epoch == 5
    want = loaded
    for _ in range(3):
        want = want * 0.999                                     # torch 2.0.1: chained from the group's current lr
        assert mine2.param_groups[0]["lr"] == pytest.approx(want, rel=1e-15)
        msch2.step()
    with pytest.raises(KeyError):
        ExponentialLR(_Opt(1e-3), gamma=0.9, last_epoch=3)


def test_mapping_json_matches_reference_format(tmp_path):
    ld = types.SimpleNamespace(session_id_to_idx={"4-24": 0, "5-4": 1}, speaking_mode_id_to_idx={"normal": 0, "silent": 1})
    DeviceLoader.save_session_and_speaking_mode_mapping_json(ld, tmp_path)
    assert json.load(open(tmp_path / "session_idx_to_id.json")) == {"0": "4-24", "1": "5-4"}
    assert json.load(open(tmp_path / "speaking_mode_idx_to_id.json")) == {"0": "normal", "1": "silent"}
    ld.session_id_to_idx = None
    with pytest.raises(RuntimeError):
        DeviceLoader.save_session_and_speaking_mode_mapping_json(ld, tmp_path)
