"""CPU-side checks of the C-ABI library: it loads without a GPU and exports every symbol include/*.h declares."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols(header="vqs.h"):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vqs_[a-z_0-9]+)\s*\(", src)))


def test_library_loads_and_exports_header_symbols():
    from t2v_metrics_amd import engine
    lib = engine.load_library()
    boundary, hooks = _declared_symbols(), _declared_symbols("vqs_debug.h")
    assert len(boundary) >= 15
    # the drop-in boundary carries no test hook or lab switch; those live in include/vqs_debug.h
    assert not [n for n in boundary if n.startswith(("vqs_debug_", "vqs_lab_"))], boundary
    assert hooks and all(n.startswith("vqs_debug_") for n in hooks), hooks
    declared = sorted(set(boundary) | set(hooks))
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/vqs.h / vqs_debug.h but not exported"
    assert set(declared) == set(engine.exported_symbols()), "engine.py signatures out of sync with include/vqs.h + vqs_debug.h"


def test_qwen_header_symbols_are_exported_and_bound():
    """include/vqs_qwen.h: every declared entry point is exported and has a ctypes signature in qwen/engine.py; sizes are
    pure arithmetic (no device needed)."""
    import ctypes
    from t2v_metrics_amd import engine
    from t2v_metrics_amd.qwen import engine as qengine
    lib = engine.load_library()
    declared = _declared_symbols("vqs_qwen.h")
    assert len(declared) >= 9
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/vqs_qwen.h but not exported"
    assert set(declared) == set(qengine._SIGS), "qwen/engine.py signatures out of sync with include/vqs_qwen.h"
    for name, (res, args) in qengine._SIGS.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    c = qengine.VqsQwenConfig(32, 1280, 16, 3420, 1176, 4, 3584, (1 << 7) | (1 << 15) | (1 << 23) | (1 << 31), 1e-6,
                              152064, 3584, 28, 28, 4, 18944, 1e-6)
    h = ctypes.c_void_p()
    assert lib.vqs_qwen_create(ctypes.byref(c), ctypes.byref(h)) == 0
    assert lib.vqs_qwen_packed_bytes(h) > 2 * 28 * 2 * 18944 * 3584          # at least the interleaved gate|up copies
    assert lib.vqs_qwen_vision_workspace_bytes(h, 3072, 3072) > 3072 * 1280 * 4
    assert lib.vqs_qwen_vision_workspace_bytes(h, 3071, 3072) == 0
    assert lib.vqs_qwen_score_workspace_bytes(h, 32, 808) > 32 * 808 * 3584 * 4
    lib.vqs_qwen_destroy(h)
    c.v_hidden = 1281
    assert lib.vqs_qwen_create(ctypes.byref(c), ctypes.byref(h)) != 0


def test_create_rejects_bad_config_and_reports():
    import ctypes
    from t2v_metrics_amd import engine
    from t2v_metrics_amd.config import get_config
    lib = engine.load_library()
    c = engine.make_vqs_config(get_config("tiny"))
    c.d_kv = 32
    h = ctypes.c_void_p()
    assert lib.vqs_create(ctypes.byref(c), ctypes.byref(h)) != 0
    assert b"d_kv" in lib.vqs_last_error(h)
    lib.vqs_destroy(h)
    c = engine.make_vqs_config(get_config("clip-flant5-xxl"))
    assert lib.vqs_create(ctypes.byref(c), ctypes.byref(h)) == 0
    # sizes are pure arithmetic: no device needed
    assert lib.vqs_packed_bytes(h) > 2 * 24 * (3 * 4096 * 4096 + 2 * 10240 * 4096)
    assert lib.vqs_score_workspace_bytes(h, 256, 33, 2) > 256 * 608 * 4096 * 4
    assert lib.vqs_score_workspace_bytes(h, 0, 33, 2) == 0
    lib.vqs_destroy(h)


def test_missing_library_fails_loudly(tmp_path):
    from t2v_metrics_amd import engine
    with pytest.raises(engine.VqsError, match="no CPU fallback"):
        engine.load_library(str(tmp_path / "nope.so"))


def test_engine_refuses_cpu_device():
    from t2v_metrics_amd import engine
    from t2v_metrics_amd.config import get_config
    with pytest.raises(engine.VqsError, match="no CPU path"):
        engine.VqsEngine(get_config("tiny"), {}, device="cpu")
