"""Build libmipnerf_hip.so (the drop-in library) and libmipnerf_diag.so (measurement tooling) for gfx950 in-tree (hipcc cross-compiles without a GPU).

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

# generated MFMA kernels: ReLU must be a single v_max_f32 that the COMPILER emits (no inline asm next to MFMAs, see
# gen_mlp_bf16.py relu1); without IEEE mode llvm.maxnum needs no canonicalising pre-instruction
NO_IEEE = ["-fno-honor-nans", "-mno-amdgpu-ieee"]

UNITS = [
    # (source, extra flags)
    ("kernels_ray.hip", ["-ffp-contract=off"]),
    ("kernels_train.hip", ["-ffp-contract=off"]),
    ("kernels_360.hip", ["-ffp-contract=off"] + (["-DMIP_IPE360_ROW_PAD=" + os.environ["MLP_IPE360_ROW_PAD"]] if os.environ.get("MLP_IPE360_ROW_PAD") else [])),
    ("kernels_resample_grad.hip", ["-ffp-contract=off"]),
    ("kernels_pack.hip", []),
    ("kernels_mlp_f32.hip", ["-ffp-contract=off"]),
    ("kernels_gemm_f32.hip", ["-ffp-contract=off"]),
    ("mlp_bf16_gen.hip", NO_IEEE + ["-ffp-contract=off"]),    # the fused IPE must round like kernels_ray.hip
    ("mlp_bf16_trainfwd_gen.hip", NO_IEEE + ["-ffp-contract=off"]),
    ("mlp_bf16_dgrad_gen.hip", NO_IEEE),
    ("kernels_wgrad.hip", ["-DMIP_WGRAD_NT=" + os.environ.get("MLP_WGRAD_NT", "1"), "-DMIP_WGRAD_STAGES=" + os.environ.get("MLP_WGRAD_STAGES", "4"),
                           "-DMIP_WGRAD_TR=" + os.environ.get("MLP_WGRAD_TR", "0"),
                           "-DMIP_WGRAD_RECOMPUTE_PROBE=" + os.environ.get("MLP_WGRAD_RECOMPUTE_PROBE", "0"),
                           "-DMIP_WGRAD_RECOMPUTE_SCHED=" + os.environ.get("MLP_WGRAD_RECOMPUTE_SCHED", "0")]),
    ("kernels_eval.hip", ["-ffp-contract=off"]),
    ("selftest.hip", ["-ffp-contract=off"]),
    ("capi.hip", []),
]
# measurement tooling (in-process MFMA ceilings, CU -> CU hand-off probe): its own library, include/mipnerf_diag.h -- the drop-in
# library above is the hot path only
DIAG_LIB = os.path.join(CSRC, "libmipnerf_diag.so")
DIAG_UNITS = [("kernels_diag.hip", []), ("diag_capi.hip", [])]
COMMON = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-fno-gpu-rdc", "-Wall",
          "-Wno-unused-function", "-Wno-unused-variable", "-Wno-unused-value", "-Wno-unused-result",
          # range reduction of the bf16 pipeline's fast sine (raymath.hpp: sin_fast): 1 = two-float fp32, 0 = fp64
          "-DMIP_SIN_FAST_TWOFLOAT=" + os.environ.get("MLP_SIN_TWOFLOAT", "0"),
          # wave scans / reductions of the ray-side kernels (raywave.hpp): 1 = DPP row_shr / row_bcast, 0 = __shfl (ds_bpermute)
          "-DMIP_WAVE_DPP=" + os.environ.get("MLP_WAVE_DPP", "1")]


# Timing-experiment knobs whose build gives WRONG results (value = the harmless default).  A stale variable in the environment must not
# silently produce a product library with wrong gradients: such a build needs MIPNERF_EXPERIMENT_BUILD=1 AND its own MIPNERF_LIB_NAME, and
# the library it produces refuses mipnerf_create() unless the process sets MIPNERF_ALLOW_EXPERIMENT_LIB=1 (capi.hip).
WRONG_RESULT_KNOBS = {"MLP_ABLATE_BARRIER": "0", "MLP_ABLATE_WAIT": "0", "MLP_ABLATE_LDA": "0", "MLP_F32R_GEN_ABLATE": "0", "MLP_F32R_ABLATE": "",
                      "MLP_TRAIN_ABLATE_TMFMA": "0", "MLP_WGRAD_TR": "0", "MLP_TRAIN_SKIP_STORES": "0", "MLP_WGRAD_RECOMPUTE_PROBE": "0",
                      "MLP_PRE_ABLATE_STORES": "0", "MLP_TRUNK_ABLATE_PRELOADS": "0"}


def experiment_flags():
    """The wrong-result knobs that are switched on, as "NAME=value ..." ("" for a product build); raises when they are set without the opt-in."""
    on = [f"{k}={os.environ[k]}" for k, d in sorted(WRONG_RESULT_KNOBS.items()) if os.environ.get(k, d) != d]
    if on and (os.environ.get("MIPNERF_EXPERIMENT_BUILD") != "1" or os.path.basename(LIB) == "libmipnerf_hip.so"):
        raise RuntimeError("timing-experiment variables are set (" + " ".join(on) + "): they produce WRONG results.  Unset them, or opt in with "
                           "MIPNERF_EXPERIMENT_BUILD=1 and a MIPNERF_LIB_NAME other than libmipnerf_hip.so")
    return " ".join(on)


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
    """Run the two generators.  Every architecture of gen_mlp_bf16.VARIANTS gets its own inference kernel (mlp_bf16_gen_v<i>.hip),
    its training kernels + table blob when mlp_train_plan covers it, and a row in the generated dispatch headers; files of
    variants that no longer exist are removed first, so adding / removing a shape is an edit of VARIANTS and a rebuild."""
    import re
    for f in os.listdir(CSRC):
        if re.fullmatch(r"(mlp_bf16(_trainfwd|_trainfwd_pre|_dgrad)?_gen_v\d+\.(hip|o)|_gen_train_tables(_v\d+)?\.bin|mlp_f32r_gen_v\d+\.(hip|o)|_gen_f32r_tables_v\d+\.bin|"
                        r"pre_gemm_gen_v\d+\.(hip|o)|mlp_bf16_pre_gen_v\d+\.(hip|o)|mlp_bf16_fused_gen_v\d+\.(hip|o)|_gen_pre_tables_v\d+\.bin)", f):
            os.remove(os.path.join(CSRC, f))
    subprocess.check_call([sys.executable, os.path.join(CSRC, "gen_mlp_bf16.py"), CSRC])
    subprocess.check_call([sys.executable, os.path.join(CSRC, "gen_mlp_train.py"), CSRC])
    subprocess.check_call([sys.executable, os.path.join(CSRC, "gen_mlp_f32r.py"), CSRC])
    subprocess.check_call([sys.executable, os.path.join(CSRC, "gen_pre_gemm.py"), CSRC])


def variant_units():
    """(source, flags) of the generated per-variant kernels present after generate()."""
    import re
    out = []
    for f in sorted(os.listdir(CSRC)):
        if re.fullmatch(r"mlp_bf16_gen_v\d+\.hip", f) or re.fullmatch(r"mlp_bf16_trainfwd(_pre)?_gen_v\d+\.hip", f):
            out.append((f, NO_IEEE + ["-ffp-contract=off"]))
        elif re.fullmatch(r"(pre_gemm_gen|mlp_bf16_pre_gen|mlp_bf16_fused_gen)_v\d+\.hip", f):      # two-kernel bf16 form of wide encodings (gen_pre_gemm.py)
            out.append((f, NO_IEEE + ["-ffp-contract=off"]))
        elif re.fullmatch(r"mlp_bf16_dgrad_gen_v\d+\.hip", f):
            out.append((f, NO_IEEE))
        elif re.fullmatch(r"mlp_f32r_gen_v\d+\.hip", f):              # register-resident fp32 kernels (gen_mlp_f32r.py)
            out.append((f, NO_IEEE + ["-ffp-contract=off"] + (["-DMLP_F32R_ABLATE=" + os.environ["MLP_F32R_ABLATE"]] if os.environ.get("MLP_F32R_ABLATE") else []) +
                        ["-DMLP_F32R_NAT_NT=" + os.environ.get("MLP_F32R_NAT_NT", "1")]))
    return out


def tables_object() -> str:
    """Link the binary training tables (written by gen_mlp_train.py, one blob per trainable variant) into the library: a
    one-object file made with the assembler's .incbin, so mlp_train_plan.py stays the only definition of those tables."""
    src = os.path.join(CSRC, "_gen_train_tables.c")
    obj = os.path.join(CSRC, "train_tables.o")
    with open(src, "w") as f:
        f.write("/* generated by build.py */\n")
        import re
        for name in sorted(n for n in os.listdir(CSRC) if re.fullmatch(r"_gen_train_tables(_v\d+)?\.bin", n)):
            sfx = name[len("_gen_train_tables"):-len(".bin")]
            blob = os.path.join(CSRC, name)
            f.write('__asm__(".section .rodata\\n.global mip_train_tables%s\\n.balign 16\\n'
                    'mip_train_tables%s:\\n.incbin \\"%s\\"\\n.previous\\n");\n' % (sfx, sfx, blob))
        for name in sorted(n for n in os.listdir(CSRC) if re.fullmatch(r"_gen_f32r_tables_v\d+\.bin", n)):
            sfx = name[len("_gen_f32r_tables"):-len(".bin")]
            f.write('__asm__(".section .rodata\\n.global mip_f32r_tables%s\\n.balign 16\\n'
                    'mip_f32r_tables%s:\\n.incbin \\"%s\\"\\n.previous\\n");\n' % (sfx, sfx, os.path.join(CSRC, name)))
        for name in sorted(n for n in os.listdir(CSRC) if re.fullmatch(r"_gen_pre_tables_v\d+\.bin", n)):
            sfx = name[len("_gen_pre_tables"):-len(".bin")]
            f.write('__asm__(".section .rodata\\n.global mip_pre_tables%s\\n.balign 16\\n'
                    'mip_pre_tables%s:\\n.incbin \\"%s\\"\\n.previous\\n");\n' % (sfx, sfx, os.path.join(CSRC, name)))
    subprocess.check_call(["gcc", "-c", "-fPIC", src, "-o", obj])
    return obj


def build(force: bool = False, verbose: bool = True) -> str:
    experiment_flags()            # refuse BEFORE the generators overwrite the tracked sources with an ablated form
    generate()
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp", ".bin"))]
    deps.append(os.path.join(os.path.dirname(HERE), "include", "mipnerf_hip.h"))
    deps.append(os.path.join(os.path.dirname(HERE), "include", "mipnerf_diag.h"))
    stamp = os.path.join(CSRC, ".build_stamp_" + os.path.basename(LIB))
    exp = experiment_flags()
    units = UNITS[:-1] + variant_units() + [("capi.hip", ['-DMIPNERF_EXPERIMENT_BUILD="%s"' % exp] if exp else [])]          # capi.hip last
    dig = _digest(deps, COMMON + sum((f for _, f in units), []) + [f"{k}={v}" for k, v in sorted(os.environ.items()) if k.startswith("MLP_")])
    if not force and os.path.exists(LIB) and os.path.exists(DIAG_LIB) and os.path.exists(stamp) and open(stamp).read() == dig:
        if verbose:
            print(f"[build] {LIB} is up to date")
        return LIB
    cc = hipcc()
    objs = []
    procs = []
    for src, extra in units + DIAG_UNITS:
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
    diag_objs = objs[len(units):]
    objs = objs[:len(units)]
    objs.append(tables_object())
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
    # The dynamic symbol table is the C ABI and nothing else: a linker version script keeps `mipnerf_*` (the entry points
    # include/mipnerf_hip.h / mipnerf_diag.h declare) and makes every C++ launcher, kernel handle and table blob local.
    vers = os.path.join(stub_dir, "exports.map")
    with open(vers, "w") as f:
        f.write("{ global: mipnerf_*; local: *; };\n")
    cmd = ["g++", "-shared", "-fPIC", "-o", LIB] + objs + [
        "-Wl,--version-script=" + vers, "-Wl,--no-as-needed", "-L" + stub_dir, "-lamdhip64", "-Wl,--as-needed", "-Wl,--allow-shlib-undefined",
        "-Wl,-rpath," + rocm_lib, "-Wl,--enable-new-dtags", "-lstdc++", "-lm"]
    if verbose:
        print("[build]", " ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    cmd = ["g++", "-shared", "-fPIC", "-o", DIAG_LIB] + diag_objs + cmd[cmd.index("-Wl,--version-script=" + vers):]
    if verbose:
        print("[build]", " ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    with open(stamp, "w") as f:
        f.write(dig)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
