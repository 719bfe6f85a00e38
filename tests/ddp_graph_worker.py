"""The graph replay under torch.distributed -- three graphs around eager exchanges (STEGAN_DDP_GRAPH=three, the default; side
branch split over the first two, STEGAN_SPLIT_SIDE=0: whole in the first) or ONE graph with the two RCCL all-reduces captured
inside it (=single, opt-in) -- under a 1-rank RCCL group against the golden fixture, ending in the ordered teardown
(``parallel.shutdown``); run in-process by tests/test_gpu_train_step.py and usable as a script (then a fatal signal prints the
Python AND the C++ stacks of every thread: a death in here must describe itself)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "golden"))

import cases  # noqa: E402
from test_gpu_train_step import build, close, norms, to_batch  # noqa: E402


def main(expect=None):
    """``expect``: the arrangement the capture must end in (default: what STEGAN_DDP_GRAPH asks for, else the library's own
    default -- a silent fallback would hide a capture failure)."""
    from ste_gan_amd.train import GraphedTrainStep
    name = "step_c48_b3"
    g = np.load(os.path.join(HERE, "golden", name + ".npz"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29300 + os.getpid() % 300), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    cfg, netG, netD, optG, optD, step = build(name)
    assert step.comm is not None
    sdG = {k: v.clone() for k, v in netG.state_dict().items()}
    sdD = {k: v.clone() for k, v in netD.state_dict().items()}
    gs = GraphedTrainStep(step, to_batch(cases.step_batch(name, 0)), warmup=2)
    want = expect or os.environ.get("STEGAN_DDP_GRAPH", GraphedTrainStep.DDP_GRAPH_DEFAULT)
    assert gs.ddp_mode == want, f"captured as {gs.ddp_mode!r}, wanted {want!r} (a silent fallback would hide a capture failure)"
    assert (gs.graphs is not None and len(gs.graphs) == 3) if want == "three" else (gs.graph is not None and gs.graphs is None)
    netG.load_state_dict(sdG); netD.load_state_dict(sdD)
    for opt in (optG, optD):
        opt.reset_state()
    for s in range(2):
        out = gs(to_batch(cases.step_batch(name, s)))
        st = gs.read_stats(out)
        ref = g[f"s{s}.losses"]
        close([st["discriminator"], st["generator"], st["multi_td"], st["speech_unit"], st["phoneme"],
               st["feature_matching"]], ref[[0, 1, 3, 4, 5, 6]], 1e-4, f"s{s}.losses")
        close(norms(netG.state_dict().values()), g[f"s{s}.g_param_norm"], 2e-5, f"s{s}.g_param_norm")
        close(norms(netD.state_dict().values()), g[f"s{s}.d_param_norm"], 2e-5, f"s{s}.d_param_norm")
    torch.cuda.synchronize()
    from ste_gan_amd.parallel import shutdown
    shutdown(gs, step)                 # ordered teardown: graphs + comm stream released, then destroy_process_group
    print("RESULT OK", flush=True)


if __name__ == "__main__":
    import faulthandler
    faulthandler.enable(all_threads=True)                        # Python stacks of all threads on SIGABRT / SIGSEGV / ...
    os.environ.setdefault("TORCH_SHOW_CPP_STACKTRACES", "1")
    try:
        torch._C._set_print_stack_traces_on_fatal_signal(True)   # ... and the C++ stacks (c10's fatal-signal handler)
    except AttributeError:
        pass
    main()
