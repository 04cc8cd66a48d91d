"""Loader of the staged reference (`oracle/_ref/`, see build_ref.py)  --  TEST / BENCH INFRASTRUCTURE ONLY.

`load()` imports the reference's own `models.mip`, `models.mip_nerf.MipNerf` and `datasets.datasets.Rays` from
oracle/_ref with the import recipe of SURVEY.md section 8(c): stub `cv2` (used only at datasets.py:196), put the staged tree
FIRST on sys.path so that its `datasets/` package wins over the pip-installed HuggingFace `datasets`.  Returns None when
nothing is staged (the caller then falls back to the numpy port and says so)."""
import importlib
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
REF_DIR = os.path.join(HERE, "_ref")


def available() -> bool:
    return os.path.exists(os.path.join(REF_DIR, "models", "mip_nerf.py"))


def load():
    if not available():
        return None
    sys.dont_write_bytecode = True
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    for name in ("datasets", "datasets.datasets", "models", "models.mip", "models.mip_nerf", "utils", "utils.lr_schedule"):
        m = sys.modules.get(name)
        if m is not None and not str(getattr(m, "__file__", "")).startswith(REF_DIR):
            del sys.modules[name]       # e.g. HuggingFace `datasets` imported earlier by something else
    sys.path.insert(0, REF_DIR)
    try:
        mip = importlib.import_module("models.mip")
        mip_nerf = importlib.import_module("models.mip_nerf")
        ds = importlib.import_module("datasets.datasets")
        lr = importlib.import_module("utils.lr_schedule")
    finally:
        sys.path.remove(REF_DIR)
    return types.SimpleNamespace(mip=mip, MipNerf=mip_nerf.MipNerf, Rays=ds.Rays, MipLRDecay=lr.MipLRDecay)
