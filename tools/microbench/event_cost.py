#!/usr/bin/env python3
"""Host cost of the cross-stream ordering primitives (torch events vs rs_context_wait_for), null stream vs created streams."""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

rs = importlib.import_module("racing-slam_amd").rsgpu
a, b = rs.Context(0), rs.Context(0)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
null = torch.cuda.current_stream()
ev = torch.cuda.Event()


def t(fn, n=2000):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    dt = time.perf_counter() - t0
    torch.cuda.synchronize()
    return 1e6 * dt / n


print("torch ev.record(null stream)      %.2f us" % t(lambda: ev.record(null)))
print("torch ev.record(created stream)   %.2f us" % t(lambda: ev.record(s1)))
print("torch s2.wait_event(ev)           %.2f us" % t(lambda: s2.wait_event(ev)))
print("torch null.wait_event(ev)         %.2f us" % t(lambda: null.wait_event(ev)))
b.use_stream(s1)
print("rs a(null).wait_for(b(s1))        %.2f us" % t(lambda: a.wait_for(b)))
print("rs b(s1).wait_for(a(null))        %.2f us" % t(lambda: b.wait_for(a)))
a.use_stream(s2)
print("rs a(s2).wait_for(b(s1))          %.2f us" % t(lambda: a.wait_for(b)))
print("empty ctypes call (abi_version)   %.2f us" % t(lambda: a.lib.rs_abi_version()))
