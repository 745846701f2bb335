"""
Live per-kernel timing with HIP events on the launch stream (``torch.cuda.Event`` records on torch's
current stream, which is the stream every kernel of this package is launched on).  Used by bench.py
to fill the ``roofline`` object: algorithmic FLOPs (or bytes) per launch / measured launch duration.
"""
from __future__ import annotations

import collections

import torch


class KernelTimer:
    """kinds=None times every instrumented launch; a set of kind names (e.g. {"conv3x3"}) times only those, which keeps
    the event overhead inside a timed region negligible."""

    def __init__(self, kinds=None):
        self.records = []  # (kind, flops, bytes, start_event, end_event)
        self.kinds = set(kinds) if kinds else None

    def wants(self, kind):
        return self.kinds is None or kind in self.kinds

    def begin(self):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def end(self, kind, flops, nbytes, start):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        self.records.append((kind, float(flops), float(nbytes), start, e))

    def summary(self):
        """kind -> dict(launches, ms, flops, bytes, avg_us, tflops, gbps); call after a device synchronize."""
        agg = collections.OrderedDict()
        for kind, fl, by, s, e in self.records:
            a = agg.setdefault(kind, dict(launches=0, ms=0.0, flops=0.0, bytes=0.0))
            a["launches"] += 1
            a["ms"] += s.elapsed_time(e)
            a["flops"] += fl
            a["bytes"] += by
        for a in agg.values():
            sec = max(a["ms"], 1e-9) * 1e-3
            a["avg_us"] = a["ms"] * 1e3 / a["launches"]
            a["tflops"] = a["flops"] / sec / 1e12
            a["gbps"] = a["bytes"] / sec / 1e9
        return agg


_active = None


def set_timer(t):
    global _active
    _active = t


def active():
    return _active
