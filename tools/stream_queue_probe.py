#!/usr/bin/env python
"""Which torch pool streams really run beside the current (null) stream?  HIP multiplexes its streams onto a few hardware queues;
two streams that share one are serialised.  For each of the first N pool streams: a single-thread spinning kernel
(torch.cuda._sleep) on the null stream and one on the pool stream, wall time against one spin (1.0 = side by side, 2.0 = serial)."""
import sys, time
import torch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
TICKS = 2_000_000
torch.cuda.init(); torch.zeros(1, device="cuda")
cur = torch.cuda.current_stream()


def wall(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


torch.cuda._sleep(TICKS)
one = min(wall(lambda: torch.cuda._sleep(TICKS)) for _ in range(3))
print(f"one spin: {one:.3f} ms")
out = []
for k in range(n):
    s = torch.cuda.Stream()

    def both():
        with torch.cuda.stream(s):
            torch.cuda._sleep(TICKS)
        torch.cuda._sleep(TICKS)

    both()
    t = min(wall(both) for _ in range(3))
    out.append(f"{k}:{t / one:.2f}")
print("pool stream : wall / one spin   " + "  ".join(out))
