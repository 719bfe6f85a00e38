"""CPU, world_size 2, gloo: the data-parallel path of the train step.

Each rank runs the oracle's train step on ITS shard of the batch (ste_gan_amd.parallel.shard_batch) and exchanges
gradients through ste_gan_amd.parallel.FlatGradAllReduce -- the class the GPU trainer uses, here on its gloo / CPU
branch -- exactly where GANTrainStep does (after each backward, before each optimiser step).  Because every loss is a
batch mean, the result must equal the single-process step on the whole batch."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def _state():
    from oracle import shapes as oshapes
    from oracle.detfill import fill_state_dict
    return (fill_state_dict(oshapes.generator_shapes("SPEECH_UNITS", 16)), fill_state_dict(oshapes.discriminator_shapes(True)),
            fill_state_dict(oshapes.encoder_shapes(64, 3, 1, dim_feedforward=64)))


def _batch(B=4, T=5):
    from oracle.detfill import det_input, det_ints
    su = det_input("ddp.su", (B, T, 256))
    return {"su": su, "feats": su, "x_real": det_input("ddp.x", (B, 16 * T, 8), "tanh"),
            "sess": det_ints("ddp.sess", (B,), 17), "phonemes": det_ints("ddp.ph", (B, T), 48)}


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import stegan_oracle as O
    from ste_gan_amd.parallel import FlatGradAllReduce, shard_batch
    comm = FlatGradAllReduce()
    assert comm.world_size == world

    def sync(grads):
        keys = list(grads)
        flat = torch.cat([grads[k].reshape(-1) for k in keys])
        comm.wait(comm.launch(flat))                       # SUM over ranks (gloo branch: blocking)
        flat /= world
        out, o = {}, 0
        for k in keys:
            n = grads[k].numel()
            out[k] = flat[o:o + n].view_as(grads[k])
            o += n
        return out

    g_sd, d_sd, e_sd = _state()
    d_train = {k: v for k, v in d_sd.items() if not O.is_sn_buffer(d_sd, k)}
    local = shard_batch(_batch(), rank, world)
    assert local["su"].shape[0] == 2
    res = O.train_step(g_sd, d_sd, e_sd, O.adamw_init(g_sd), O.adamw_init(d_train), local, grad_sync=sync)
    torch.save({"g": g_sd, "d": d_sd, "loss_d": res["loss_d"]}, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_step_equals_full_batch_step(tmp_path):
    from oracle import stegan_oracle as O
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (torch.load(tmp_path / f"rank{r}.pt") for r in (0, 1))
    g_sd, d_sd, e_sd = _state()
    d_train = {k: v for k, v in d_sd.items() if not O.is_sn_buffer(d_sd, k)}
    O.train_step(g_sd, d_sd, e_sd, O.adamw_init(g_sd), O.adamw_init(d_train), _batch())
    for k in g_sd:
        assert torch.equal(r0["g"][k], r1["g"][k]), f"replicas diverged on {k}"
        torch.testing.assert_close(r0["g"][k], g_sd[k], rtol=2e-4, atol=2e-6, msg=lambda m: f"{k}: {m}")
    for k in d_train:
        assert torch.equal(r0["d"][k], r1["d"][k]), f"replicas diverged on {k}"
        torch.testing.assert_close(r0["d"][k], d_sd[k], rtol=2e-4, atol=2e-6, msg=lambda m: f"{k}: {m}")


def test_shard_batch_rejects_ragged_split():
    from ste_gan_amd.parallel import shard_batch
    with pytest.raises(ValueError):
        shard_batch({"x": torch.zeros(5, 3)}, 0, 2)
