#!/usr/bin/env python3
"""Host issue time and completion time of the pass's front-end chains (bench.py build_pass), one stream against four."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

args = argparse.Namespace(gpus=1)
e = bench.setup(args)
torch = e.torch
for streams, graph in ((1, False), (4, False), (4, True)):
    one_pass, cpu_pass, meta = bench.build_pass(e, streams, graph)
    fe = meta["front_end"]
    if graph:
        g = meta["last"]["front_end_graph"]
        fe = lambda serial: g.replay()      # noqa: E731
    for serial in ((True,) if streams == 1 else (False,) if graph else (True, False)):
        for _ in range(20):
            fe(serial)
        torch.cuda.synchronize()
        issue, done = [], []
        for _ in range(200):
            t0 = time.perf_counter()
            fe(serial)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            issue.append(t1 - t0)
            done.append(t2 - t0)
        issue.sort(); done.sort()
        print(f"streams={streams} graph={graph} serial={serial}: host issue {1e6 * issue[100]:.1f} us, complete {1e6 * done[100]:.1f} us (median of 200)")
