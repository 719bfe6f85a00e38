"""Helpers shared by the timing scripts: bf16 activation image of a tensor, graph-captured back-to-back timing."""
import torch


def make_image(t):
    B, C, T = t.shape
    CB = (C + 7) // 8
    full = torch.zeros(B, CB * 8, T, device=t.device, dtype=t.dtype)
    full[:, :C] = t
    return full.view(B, CB, 8, T).permute(0, 1, 3, 2).contiguous().to(torch.bfloat16)


def timeit(fn, n=20):
    """us per call: n calls captured into one hipGraph (no host overhead, no event brackets), replayed three times."""
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for _ in range(n):
                fn()
    torch.cuda.current_stream().wait_stream(side)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (3 * n) * 1e3
