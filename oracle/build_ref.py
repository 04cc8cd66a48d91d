#!/usr/bin/env python3
"""Recipe for `oracle/_ref/`: the reference's OWN hot-path modules, staged so they can travel to the GPU box.

TEST / BENCH INFRASTRUCTURE ONLY -- never imported by `mipnerf_pl_amd/`.

The reference (hjxwhy/mipnerf_pl) is Python, so there is nothing to compile: "building" `oracle/_ref` means copying
the few files its forward path needs from where they lie under /root/reference into the git-ignored directory
`oracle/_ref/` (listed in .gitignore, NOT in .gpurunignore: it ships with a gpurun snapshot like a built .so does and
never enters the history).  `/root/reference` does not exist on the GPU box; this staged copy is what lets
`bench.py`'s `cpu_baseline` leg time the reference's own `MipNerf.forward` (kind "reference") on the GPU host's cores
instead of the numpy port.

    python oracle/build_ref.py            # no-op with a message when /root/reference is absent

Files staged (verbatim, byte-identical; a manifest with their sha256 is written next to them):
    models/__init__.py, models/mip.py, models/mip_nerf.py   -- the hot path
    datasets/__init__.py, datasets/datasets.py              -- only for the `Rays` namedtuple mip.py:4 imports
    utils/lr_schedule.py                                    -- MipLRDecay (training-trajectory goldens)
"""
import hashlib
import json
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("MIPNERF_REFERENCE", "/root/reference")
DST = os.path.join(HERE, "_ref")
FILES = ["models/__init__.py", "models/mip.py", "models/mip_nerf.py", "datasets/__init__.py", "datasets/datasets.py",
         "utils/lr_schedule.py"]


def build(verbose: bool = True) -> bool:
    if not os.path.isdir(REF):
        if verbose:
            print(f"[build_ref] {REF} is absent: keeping whatever is staged under {DST}")
        return os.path.exists(os.path.join(DST, "MANIFEST.json"))
    manifest = {}
    for rel in FILES:
        src, dst = os.path.join(REF, rel), os.path.join(DST, rel)
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        shutil.copyfile(src, dst)
        manifest[rel] = hashlib.sha256(open(dst, "rb").read()).hexdigest()
    with open(os.path.join(DST, "MANIFEST.json"), "w") as f:
        json.dump({"source": "hjxwhy/mipnerf_pl (mounted at /root/reference), staged verbatim by oracle/build_ref.py",
                   "sha256": manifest}, f, indent=1)
    if verbose:
        print(f"[build_ref] staged {len(FILES)} reference files under {DST}")
    return True


if __name__ == "__main__":
    sys.exit(0 if build() else 1)
