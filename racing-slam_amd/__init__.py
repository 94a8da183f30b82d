"""racing-slam_amd — MI355X-native implementation of Racing-SLAM's per-frame hot path.

Layout:
  csrc/      hand-written HIP kernels (gfx950) + the C-ABI of include/rsgpu.h -> librsgpu.so
  host/      C++ mirror of the reference's MapMatcher / Triangulation / Optimization /
             LocalWindow interfaces over the C-ABI
  rsgpu.py   ctypes binding used by tests/, bench.py and __graft_entry__.py
  synth.py   synthetic inputs shaped like BASELINE.json's configs
  build.py   hipcc build of librsgpu.so

The directory name contains a hyphen: import it with
    importlib.import_module("racing-slam_amd")
"""
from . import build as _build  # noqa: F401
from . import rsgpu, synth  # noqa: F401


def build(force=False, verbose=False):
    return _build.build(force=force, verbose=verbose)
