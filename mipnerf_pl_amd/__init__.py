"""MI355X-native Mip-NeRF volume-rendering hot path (drop-in for models/mip.py + models/mip_nerf.py
of hjxwhy/mipnerf_pl).  Compute lives in csrc/ (hand-written HIP for gfx950 behind a C ABI);
this package is the thin Python host that mirrors the reference's class / function contracts."""
from .rays import Rays, Rays_keys, namedtuple_map  # noqa: F401

__all__ = ["Rays", "Rays_keys", "namedtuple_map", "MipNerf", "MLP"]


def __getattr__(name):
    if name in ("MipNerf", "MLP"):
        from . import model
        return getattr(model, name)
    raise AttributeError(name)
