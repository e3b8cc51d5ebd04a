"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/rlhip.h declares; calls fail loudly (no CPU fallback) when no GPU is present."""
import ctypes as C
import os
import re

import pytest

from ranklib_amd import _native as N

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ensure_built():
    if not os.path.exists(N.LIB_PATH):
        import __graft_entry__ as g
        g.build()


def test_library_exports_every_declared_symbol():
    _ensure_built()
    hdr = open(os.path.join(ROOT, "include", "rlhip.h")).read()
    declared = sorted(set(re.findall(r"\b(rl_[a-z0-9_]+)\s*\(", hdr)))
    assert declared, "no declarations found"
    L = C.CDLL(N.LIB_PATH)
    for name in declared:
        assert hasattr(L, name), "librlhip.so does not export %s" % name
    assert sorted(N.ABI_SYMBOLS) == declared
    assert L.rl_abi_version() == 3


def test_no_cpu_fallback_without_device(gpu_available):
    _ensure_built()
    if gpu_available:
        pytest.skip("GPU present")
    assert N.device_count() == 0
    with pytest.raises(N.RankLibError) as e:
        N.Trainer(n_trees=1)
    assert "no CPU fallback" in str(e.value)
    with pytest.raises(N.RankLibError):
        N.Model("<ensemble>\n</ensemble>\n")


def test_parameter_validation_messages():
    _ensure_built()
    L = N.lib()
    p = N.RlParams()
    L.rl_params_default(C.byref(p))
    # defaults of learning/tree/LambdaMART.java:37-42
    assert (p.n_trees, p.n_leaves, p.n_threshold, p.min_leaf_support, p.early_stop_rounds, p.metric_k) == \
        (1000, 10, 256, 1, 100, 10)
    assert abs(p.learning_rate - 0.1) < 1e-8
    h = C.c_void_p()
    p.n_trees = 0
    assert L.rl_create(C.byref(p), C.byref(h)) == -1
    assert b"n_trees" in L.rl_last_error()
