"""Cost of a dependency edge inside a captured hipGraph: a chain of N tiny kernels on ONE stream vs the same chain hopping
between two streams (every launch waits for the previous one on the other stream), and a fork-join pattern (main launches
a side kernel every step but only joins at the end).  Microseconds per kernel."""
import torch

N = 200
x = torch.zeros(256, device="cuda")
a, b = torch.cuda.Stream(), torch.cuda.Stream()


def time_graph(build):
    cap = torch.cuda.Stream()
    cap.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(cap):
        with torch.cuda.graph(g, stream=cap):
            build(cap)
    torch.cuda.current_stream().wait_stream(cap)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 5 / N * 1e3


def same(cap):
    for _ in range(N):
        x.add_(1.0)


def hop(cap):
    cur, other = cap, a
    for _ in range(N):
        other.wait_stream(cur)
        with torch.cuda.stream(other):
            x.add_(1.0)
        cur, other = other, cur
    cap.wait_stream(a)


def fork(cap):
    y = torch.zeros(256, device="cuda")
    for _ in range(N // 2):
        x.add_(1.0)
        a.wait_stream(cap)
        with torch.cuda.stream(a):
            y.add_(1.0)
    cap.wait_stream(a)


print(f"same stream {time_graph(same):.2f} us/kernel | alternating streams {time_graph(hop):.2f} | "
      f"main + forked side launches, one join {time_graph(fork):.2f}")
