#!/usr/bin/env python3
"""A/B/A/B of the benchmark pass: every launch on one stream against the front-end chains on four streams (and as a graph)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

e = bench.setup(argparse.Namespace(gpus=1))
ov, _, meta = bench.build_pass(e, 4, graph=False)
serial = meta["one_pass_serial"]
ovg, _, meta_g = bench.build_pass(e, 4, graph=True)
for rep in range(3):
    for tag, fn in (("one stream", serial), ("4 streams", ov), ("4 streams, graph", ovg)):
        dt = bench.timed(e, fn, 200, 20)
        print(f"{tag:18s} {1e3 * dt / 200:.4f} ms/pass", flush=True)
