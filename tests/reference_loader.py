"""Import single files of the reference tree as they lie under /root/reference (this container only; the GPU box has no such tree).

The reference's package ``__init__`` files import every model family (API clients, decord, flash-attn ...); here the packages are
registered as bare module objects with the right ``__path__`` so that ``from ...constants import X`` inside a reference file
resolves, and only the files a test names are executed.  Optional third-party modules a file imports at its top but the tests
never call (cv2, decord, qwen_vl_utils) are replaced, when absent, by modules whose every attribute is a placeholder that raises if called."""
import importlib
import os
import sys
import types

import pytest

REF_ROOT = "/root/reference/t2v_metrics"


class _AbsentModule(types.ModuleType):
    """Stands in for an optional dependency that is not installed: ``from absent import anything`` works, calling it does not."""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)

        def absent(*a, **k):
            raise RuntimeError(f"{self.__name__}.{name} is not installed here; a test that reaches it must provide its own")
        return absent

PKG = "_t2v_reference_as_it_lies"


def reference_module(dotted: str, optional=()):
    """``reference_module("models.vqascore_models.mm_utils")`` -> that module of the reference; skips the test when the tree or one of
    the file's hard dependencies is not on this machine.  `optional`: top-level module names to stub when they do not import."""
    if not os.path.isfile(os.path.join(REF_ROOT, "score.py")):
        pytest.skip("the reference tree is not on this machine")
    for name, sub in ((PKG, ""), (PKG + ".models", "models"), (PKG + ".models.vqascore_models", "models/vqascore_models")):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = [os.path.join(REF_ROOT, sub)]      # a package whose __init__ is never executed
            m.__package__ = name
            sys.modules[name] = m
    for name in optional:
        if name not in sys.modules:
            try:
                importlib.import_module(name)
            except ImportError:
                sys.modules[name] = _AbsentModule(name)
    try:
        return importlib.import_module(PKG + "." + dotted)
    except ImportError as e:                                # a dependency of the reference missing here
        pytest.skip(f"the reference's {dotted} does not import here: {e}")
