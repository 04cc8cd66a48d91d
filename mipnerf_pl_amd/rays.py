"""`Rays` record of the reference (datasets/datasets.py:13-16): seven [B,k] float32 tensors."""
import collections

Rays = collections.namedtuple(
    "Rays", ("origins", "directions", "viewdirs", "radii", "lossmult", "near", "far"))
Rays_keys = Rays._fields


def namedtuple_map(fn, tup):
    """Apply `fn` to each element of `tup` and cast to `tup`'s namedtuple (datasets.py:20-22)."""
    return type(tup)(*map(fn, tup))
