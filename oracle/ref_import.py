"""Import the upstream GeoCalib LM path read-only from /root/reference (TEST INFRASTRUCTURE ONLY).

The upstream package imports cv2 / kornia / torchvision at package-import time
(geocalib/__init__.py:3 -> extractor.py:12 -> utils.py:8-12); none of them is touched by the
LM path, so empty placeholder modules are registered for them.  Nothing is copied, nothing under
/root/reference is written.  This module only works in the build container: the GPU box has no
/root/reference, so nothing that runs there may import it (golden vectors are generated here by
tests/golden/make_golden.py and committed).
"""
import importlib
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("GEOCALIB_REFERENCE", "/root/reference")


def available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "geocalib"))


class _Anything(types.ModuleType):
    """Placeholder module: any attribute access yields another placeholder."""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        sub = _Anything(f"{self.__name__}.{name}")
        setattr(self, name, sub)
        return sub

    def __call__(self, *a, **k):  # used as decorator / class factory at import time
        return self


def load():
    """Return a namespace with the reference modules on the LM path."""
    if not available():
        raise RuntimeError(f"reference not found at {REFERENCE_ROOT}")
    for name in ("cv2", "kornia", "torchvision", "torchvision.transforms", "kornia.geometry",
                 "kornia.geometry.transform"):
        if name not in sys.modules:
            try:
                importlib.import_module(name)
            except Exception:
                sys.modules[name] = _Anything(name)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    ns = types.SimpleNamespace()
    ns.lm_optimizer = importlib.import_module("geocalib.lm_optimizer")
    ns.perspective_fields = importlib.import_module("geocalib.perspective_fields")
    ns.camera = importlib.import_module("geocalib.camera")
    ns.gravity = importlib.import_module("geocalib.gravity")
    ns.misc = importlib.import_module("geocalib.misc")
    ns.utils = importlib.import_module("geocalib.utils")
    return ns
