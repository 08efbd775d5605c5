"""What an event record / a cross-stream wait costs the stream the event is recorded on (GPU-bound: ~20 us kernels, the host runs
ahead).  Patterns: none; record (disable-timing event behind every kernel); fork (a second, low-priority stream waits on every
event and runs a small kernel -- the library's weight-gradient fork); fork10 (every tenth); join (the main stream waits on an
event of the side stream behind every kernel)."""
import sys
import torch
dev = torch.device("cuda:0")
x = torch.zeros(1 << 24, device=dev)
y = torch.zeros(1 << 16, device=dev)
side = torch.cuda.Stream(priority=0)
N = 300


def run(mode):
    evs = [torch.cuda.Event(enable_timing=False) for _ in range(N)]
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(N):
        x.add_(1.0)
        if mode == "record":
            evs[i].record()
        elif mode in ("fork", "fork10"):
            if mode == "fork" or i % 10 == 0:
                evs[i].record()
                side.wait_event(evs[i])
                with torch.cuda.stream(side):
                    y.add_(1.0)
        elif mode == "join":
            with torch.cuda.stream(side):
                y.add_(1.0)
                evs[i].record()
            torch.cuda.current_stream().wait_event(evs[i])
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / N


base = None
for mode in ("none", "none", "record", "fork", "fork10", "join", "none"):
    g = run(mode)
    print("%-8s  %.2f us per kernel on the GPU timeline" % (mode, g))
