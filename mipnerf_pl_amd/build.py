"""Build libmipnerf_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU).

    python -m mipnerf_pl_amd.build [--force]

Steps: (1) regenerate mlp_bf16_gen.hip / mlp_plan_gen.hpp from mlp_plan.py, (2) compile each
.hip translation unit to an object (the ray-math units with -ffp-contract=off, see
raymath.hpp), (3) link the shared library next to the sources.  No torch headers are used:
the library's only dependency is the HIP runtime.
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, os.environ.get("MIPNERF_LIB_NAME", "libmipnerf_hip.so"))
ARCH = "gfx950"

UNITS = [
    # (source, extra flags)
    ("kernels_ray.hip", ["-ffp-contract=off"]),
    ("kernels_train.hip", ["-ffp-contract=off"]),
    ("kernels_pack.hip", []),
    ("kernels_mlp_f32.hip", ["-ffp-contract=off"]),
    ("mlp_bf16_gen.hip", []),
    ("selftest.hip", ["-ffp-contract=off"]),
    ("capi.hip", []),
]
COMMON = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-fno-gpu-rdc", "-Wall",
          "-Wno-unused-function", "-Wno-unused-variable", "-Wno-unused-value", "-Wno-unused-result"]


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm to build the gfx950 kernels)")


def _digest(paths, flags) -> str:
    h = hashlib.sha256()
    for p in sorted(paths):
        with open(p, "rb") as f:
            h.update(os.path.basename(p).encode())
            h.update(f.read())
    h.update(" ".join(flags).encode())
    return h.hexdigest()


def generate() -> None:
    subprocess.check_call([sys.executable, os.path.join(CSRC, "gen_mlp_bf16.py"), CSRC])


def build(force: bool = False, verbose: bool = True) -> str:
    generate()
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp"))]
    deps.append(os.path.join(os.path.dirname(HERE), "include", "mipnerf_hip.h"))
    stamp = os.path.join(CSRC, ".build_stamp_" + os.path.basename(LIB))
    dig = _digest(deps, COMMON + sum((f for _, f in UNITS), []) + [f"{k}={v}" for k, v in sorted(os.environ.items()) if k.startswith("MLP_")])
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read() == dig:
        if verbose:
            print(f"[build] {LIB} is up to date")
        return LIB
    cc = hipcc()
    objs = []
    procs = []
    for src, extra in UNITS:
        obj = os.path.join(CSRC, src.replace(".hip", ".o"))
        cmd = [cc] + COMMON + extra + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print("[build]", " ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, cwd=CSRC, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if out and verbose:
            sys.stdout.write(out.decode(errors="replace"))
        if p.returncode != 0:
            failed = True
            print(f"[build] FAILED: {src}")
    if failed:
        raise RuntimeError("hipcc failed")
    # Link by hand (not through hipcc) against an EMPTY stub named libamdhip64.so with no SONAME, so the
    # library's DT_NEEDED entry is exactly "libamdhip64.so".  PyTorch-ROCm wheels bundle their own HIP
    # runtime under that soname-less name; a DT_NEEDED of "libamdhip64.so.7" (what hipcc's link step
    # records from /opt/rocm) would pull a SECOND HIP runtime into the process, whose streams, events and
    # synchronisation are invisible to torch.  With the un-versioned name the loader reuses whichever
    # runtime is already mapped (torch's), and falls back to /opt/rocm/lib (RUNPATH) in a torch-free host.
    stub_dir = os.path.join(CSRC, "_stub")
    os.makedirs(stub_dir, exist_ok=True)
    stub_c = os.path.join(stub_dir, "stub.c")
    with open(stub_c, "w") as f:
        f.write("/* empty: only provides the DT_NEEDED name libamdhip64.so */\n")
    subprocess.check_call(["gcc", "-shared", "-fPIC", "-o", os.path.join(stub_dir, "libamdhip64.so"), stub_c])
    rocm_lib = os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib")
    cmd = ["g++", "-shared", "-fPIC", "-o", LIB] + objs + [
        "-Wl,--no-as-needed", "-L" + stub_dir, "-lamdhip64", "-Wl,--as-needed", "-Wl,--allow-shlib-undefined",
        "-Wl,-rpath," + rocm_lib, "-Wl,--enable-new-dtags", "-lstdc++", "-lm"]
    if verbose:
        print("[build]", " ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    with open(stamp, "w") as f:
        f.write(dig)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
