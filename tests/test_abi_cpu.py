"""CPU tests of the boundary: the C-ABI library loads, exports every symbol that
include/rsgpu.h declares, and refuses to compute without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "rsgpu.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rs_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(rs):
    lib = rs.load()
    syms = declared_symbols()
    assert len(syms) >= 20
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing
    assert sorted(rs.EXPORTS) == syms
    assert lib.rs_abi_version() == 3


def test_no_cpu_fallback(rs):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = C.c_void_p()
    rc = rs.load().rs_context_create(0, C.byref(h))
    assert rc == 6 and not h.value          # RS_ERR_NO_DEVICE
    with pytest.raises(rs.RsError):
        rs.Context(0)


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "racing-slam_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "pyoracle" not in src and "liboracle" not in src and "rs_oracle.h" not in src, f


def test_default_options_match_oracle(rs, oracle):
    a, b = rs.default_options(), oracle.default_options()
    for name, _ in a._fields_:
        assert getattr(a, name) == getattr(b, name), name


def test_bench_metric_is_baseline_json_s_string():
    """The line bench.py prints must carry BASELINE.json's metric character for character (the driver matches on it)."""
    import json
    import bench
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "BASELINE.json"), encoding="utf-8") as f:
        assert bench.METRIC == json.load(f)["metric"]
    assert bench.visible_gpu_count() >= 0        # sysfs / environment only: no HIP call
