#!/usr/bin/env python3
"""cost of a cross-stream dependency in front of a graph replay: stream-level wait_event before every replay vs an external event-wait
node captured at the head of the graph (torch.cuda.Event(external=True)).  Pattern of the lookahead training loop: a side stream
produces (short kernel + event record), the main stream replays a graph of a few kernels that consumes."""
import time
import torch

dev = torch.device('cuda')
x = torch.randn(8 << 20, device=dev)
y = torch.empty_like(x)
z = torch.zeros(1 << 16, device=dev)
side = torch.cuda.Stream()
main = torch.cuda.current_stream()


def body():
    for _ in range(6):
        torch.mul(x, 1.0001, out=y)


def capture(ev=None):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(main)
    with torch.cuda.stream(s):
        body()
    main.wait_stream(s)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)
        body()
    return g


def run(label, g, ev, wait_outside, n=300):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        with torch.cuda.stream(side):
            z.add_(1.0)
            ev.record(side)
        if wait_outside:
            main.wait_event(ev)
        g.replay()
    torch.cuda.synchronize()
    print(f'{label:44s}: {1e6 * (time.perf_counter() - t0) / n:8.2f} us per iteration')


ev_plain = torch.cuda.Event()
g_plain = capture()
run('no dependency (baseline)', g_plain, ev_plain, False)
run('stream-level wait_event before the replay', g_plain, ev_plain, True)
try:
    ev_ext = torch.cuda.Event(external=True)
    with torch.cuda.stream(side):
        ev_ext.record(side)
    torch.cuda.synchronize()
    g_ext = capture(ev_ext)
    run('external event-wait node inside the graph', g_ext, ev_ext, False)
except Exception as e:  # noqa: BLE001
    print('external event in capture failed:', repr(e)[:300])
