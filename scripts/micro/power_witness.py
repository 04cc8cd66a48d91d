#!/usr/bin/env python3
"""Independent witness for the power ceiling of DESIGN 4.1.  Tooling only.

The claim to test: on MLP-like NON-ZERO bf16 operands the MI355X matrix pipe is limited by the package power envelope (shader clock ~1.6-1.7 GHz
instead of 2.4), so 0.70 of the 2.5 PF datasheet peak is out of reach for ANY kernel with this operand mix -- not only for k_mlp_bf16 and the
builder's own micro-benchmarks (libmipnerf_diag.so).  The witness is the vendor GEMM (torch.matmul -> hipBLASLt) on the same operand
distributions, with the SMI's shader clock and socket power sampled at ~10 Hz beside every workload:

  * torch.matmul bf16 8192^3, NN and NT, operands all zero / MLP-like random (weights U(-0.1, 0.1), activations relu(N(0, 1)))
  * the MLP's own GEMM shape: [524288, 256] x [256, 256], chained x 8, zero / random
  * k_mlp_bf16 (mipnerf_time_mlp) zero / random, and the in-process ceilings of libmipnerf_diag.so (register-fed / LDS-fed / LDS + DMA-fed)

usage: power_witness.py [--seconds 2.5] [--out gpurun_out/power_witness.txt]
"""
import argparse
import ctypes as C
import os
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

PEAK = 2500.0


class Smi:
    """~10 Hz sampler of (shader clock MHz, socket power W): amdsmi python binding when it works, else the rocm-smi CLI (slower)."""

    def __init__(self):
        self.kind, self.h, self.cap = None, None, None
        try:
            import amdsmi
            amdsmi.amdsmi_init()
            self.amdsmi = amdsmi
            self.h = amdsmi.amdsmi_get_processor_handles()[0]
            self._read_amdsmi()
            self.kind = "amdsmi"
            try:
                self.cap = amdsmi.amdsmi_get_power_cap_info(self.h)
            except Exception as ex:      # noqa: BLE001
                self.cap = f"n/a ({ex})"
        except Exception as ex:          # noqa: BLE001
            self.err = f"{type(ex).__name__}: {ex}"
            try:
                self._read_cli()
                self.kind = "rocm-smi"
            except Exception as ex2:     # noqa: BLE001
                self.err += f"; rocm-smi: {type(ex2).__name__}: {ex2}"
        self.samples, self._stop, self._t = [], threading.Event(), None

    def _read_amdsmi(self):
        a = self.amdsmi
        clk = pw = None
        extra = {}
        try:
            m = a.amdsmi_get_gpu_metrics_info(self.h)
            g = [c for c in (m.get("current_gfxclks") or []) if isinstance(c, (int, float)) and 0 < c < 60000]
            if g:
                clk = sum(g) / len(g)
                extra["gfxclk_min"], extra["gfxclk_max"] = min(g), max(g)
            for k in ("current_socket_power", "average_socket_power"):
                v = m.get(k)
                if isinstance(v, (int, float)) and 0 < v < 5000:
                    pw = float(v)
                    break
            for k in ("temperature_hotspot", "throttle_status", "indep_throttle_status", "accumulated_prochot_residency_acc", "ppt_residency_acc", "socket_thm_residency_acc"):
                if k in m:
                    extra[k] = m[k]
        except Exception:                # noqa: BLE001
            pass
        if clk is None:
            c = a.amdsmi_get_clock_info(self.h, a.AmdSmiClkType.GFX)
            clk = float(c.get("clk", c.get("cur_clk")))
        if pw is None:
            p = a.amdsmi_get_power_info(self.h)
            for k in ("current_socket_power", "average_socket_power", "socket_power"):
                v = p.get(k)
                if isinstance(v, (int, float)) and 0 < v < 5000:
                    pw = float(v)
                    break
        return clk, pw, extra

    def _read_cli(self):
        import json
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=10).stdout
        d = list(json.loads(out).values())[0]
        clk = pw = None
        for k, v in d.items():
            if "sclk" in k.lower() and "(" in str(v):
                clk = float(str(v).split("(")[1].split("Mhz")[0])
            if "power" in k.lower() and "(w)" in k.lower():
                pw = float(v)
        return clk, pw, {}

    def read(self):
        return self._read_amdsmi() if self.kind == "amdsmi" else self._read_cli()

    def start(self):
        self.samples, self._stop = [], threading.Event()

        def loop():
            while not self._stop.is_set():
                t = time.perf_counter()
                try:
                    self.samples.append((t,) + self.read())
                except Exception:        # noqa: BLE001
                    pass
                self._stop.wait(max(0.0, 0.1 - (time.perf_counter() - t)))
        if self.kind:
            self._t = threading.Thread(target=loop, daemon=True)
            self._t.start()

    def stop(self, t_from):
        """mean / max over the samples taken after t_from (the heated-up part of the run)"""
        self._stop.set()
        if self._t:
            self._t.join()
        s = [x for x in self.samples if x[0] >= t_from and x[1] is not None]
        if not s:
            return dict(n=0)
        clk = [x[1] for x in s]
        pw = [x[2] for x in s if x[2] is not None]
        r = dict(n=len(s), sclk_mean=sum(clk) / len(clk), sclk_min=min(clk), sclk_max=max(clk))
        if pw:
            r.update(power_mean=sum(pw) / len(pw), power_max=max(pw))
        if s[-1][3]:
            r["last_extra"] = s[-1][3]
        return r


def operands(kind, m, k, n, dev, transpose_b):
    """A [m, k] 'activations', B 'weights' ([k, n], or [n, k] for the NT form)"""
    if kind == "zero":
        a = torch.zeros(m, k, device=dev, dtype=torch.bfloat16)
        b = torch.zeros((n, k) if transpose_b else (k, n), device=dev, dtype=torch.bfloat16)
    else:
        g = torch.Generator(device=dev).manual_seed(1)
        a = torch.relu(torch.randn(m, k, device=dev, generator=g)).to(torch.bfloat16)
        b = ((torch.rand((n, k) if transpose_b else (k, n), device=dev, generator=g) * 2 - 1) * 0.1).to(torch.bfloat16)
    return a, b


def run_timed(fn, seconds, flop_per_call, smi):
    """fn() enqueues ONE call; run for ~`seconds`, first 40 % heat-up; TF/s from events around the rest"""
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        fn()
    e1.record()
    torch.cuda.synchronize()
    per = e0.elapsed_time(e1) / 5 * 1e-3
    n_heat = max(1, int(0.4 * seconds / per))
    n_meas = max(1, int(0.6 * seconds / per))
    smi.start()
    for _ in range(n_heat):
        fn()
    torch.cuda.synchronize()
    t_from = time.perf_counter()
    e0.record()
    for _ in range(n_meas):
        fn()
    e1.record()
    torch.cuda.synchronize()
    s = smi.stop(t_from)
    dt = e0.elapsed_time(e1) * 1e-3
    return dict(tflops=flop_per_call * n_meas / dt / 1e12, ms=dt / n_meas * 1e3, calls=n_meas, **s)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=2.5)
    ap.add_argument("--out", default="gpurun_out/power_witness.txt")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.cuda.init()
    smi = Smi()
    rows = []
    lines = []

    def emit(s=""):
        print(s, flush=True)
        lines.append(s)

    emit(f"# power witness: {torch.cuda.get_device_name(0)}; torch {torch.__version__}; SMI source: {smi.kind or 'NONE (' + getattr(smi, 'err', '') + ')'}; "
         f"power cap: {smi.cap}; {a.seconds} s per workload (first 40 % heat-up, SMI samples and TF/s from the rest)")
    if smi.kind:
        emit(f"# idle: sclk / power = {smi.read()[:2]}")

    def add(name, r):
        rows.append((name, r))
        emit(f"{name:58s} {r['tflops']:8.1f} TF/s  {r['tflops'] / PEAK:6.3f} of 2.5 PF  {r['ms']:9.4f} ms/call  "
             f"sclk {r.get('sclk_mean', float('nan')):7.0f} MHz [{r.get('sclk_min', float('nan')):.0f}, {r.get('sclk_max', float('nan')):.0f}]  "
             f"power {r.get('power_mean', float('nan')):6.0f} W (max {r.get('power_max', float('nan')):.0f})  n={r.get('n', 0)}"
             + (f"  {r['last_extra']}" if r.get("last_extra") else ""))

    # ---- vendor GEMM, square ----
    for kind in ("zero", "random"):
        for nt in (False, True):
            A, B = operands(kind, 8192, 8192, 8192, dev, nt)
            out = torch.empty(8192, 8192, device=dev, dtype=torch.bfloat16)
            Bm = B.t() if nt else B
            add(f"torch.matmul bf16 8192^3 {'NT' if nt else 'NN'} {kind}", run_timed(lambda: torch.matmul(A, Bm, out=out), a.seconds, 2 * 8192 ** 3, smi))
            del A, B, out
    # ---- vendor GEMM, the MLP's shape chained x 8 ----
    M = 524288
    for kind in ("zero", "random"):
        for nt in (False, True):
            X, W = operands(kind, M, 256, 256, dev, nt)
            Wm = W.t() if nt else W
            y0, y1 = torch.empty_like(X), torch.empty_like(X)

            def chain():
                torch.matmul(X, Wm, out=y0)
                for i in range(7):
                    src, dst = (y0, y1) if i % 2 == 0 else (y1, y0)
                    torch.matmul(src, Wm, out=dst)
            if kind == "random":       # keep the chained activations O(1): U(-0.1, 0.1) has a gain of sqrt(256 / 300) per layer, relu(N(0,1)) rows start at ~0.8 rms
                W.mul_(1.08)
            add(f"torch.matmul bf16 [524288,256]x[256,256] x8 chain {'NT' if nt else 'NN'} {kind}", run_timed(chain, a.seconds, 8 * 2 * M * 256 * 256, smi))
            if kind == "random":
                emit(f"    (rms of the chain's last output: {float(y1.float().pow(2).mean().sqrt()):.3g}; zeros: {float((y1 == 0).float().mean()):.3f})")
            del X, W, y0, y1
    # ---- k_mlp_bf16 ----
    from mipnerf_pl_amd import MipNerf, _lib as L
    import synthetic_inputs as si
    for kind in ("zero", "random"):
        params = si.make_params(seed=0, density_gain=40.0)
        if kind == "zero":
            params = {k: v * 0 for k, v in params.items()}
        m = MipNerf(num_samples=128, precision="bf16")
        m.load_state_dict({"mlp." + k: torch.from_numpy(v.copy()) for k, v in params.items()})
        m = m.to(dev)
        enc = (torch.rand(M, 96, device=dev) * 2 - 1).to(torch.bfloat16)
        venc = torch.zeros(4096, 32, device=dev, dtype=torch.bfloat16)
        venc[:, :27] = (torch.rand(4096, 27, device=dev) * 2 - 1).to(torch.bfloat16)
        if kind == "zero":
            enc.zero_()
            venc.zero_()
        out = torch.empty(M, 4, device=dev)
        ctx = m.mlp.native(dev)
        ms = C.c_float()
        st = torch.cuda.current_stream().cuda_stream

        def mlp(iters=1):
            L.check(L.lib().mipnerf_time_mlp(ctx.handle, M, 128, enc.data_ptr(), venc.data_ptr(), m.precision, out.data_ptr(), iters, C.byref(ms), st), "time_mlp")
        mlp(20)
        per = ms.value * 1e-3
        smi.start()
        mlp(max(1, int(0.4 * a.seconds / per)))
        t_from = time.perf_counter()
        n = max(1, int(0.6 * a.seconds / per))
        mlp(n)
        s = smi.stop(t_from)
        add(f"k_mlp_bf16 (encodings given) 524288 samples {kind}", dict(tflops=1220608 * M / (ms.value * 1e-3) / 1e12, ms=ms.value, calls=n, **s))
    # ---- in-process ceilings (the builder's own micro-benchmark, for the same table) ----
    D = L.diag_lib()
    if D is not None:
        st = torch.cuda.current_stream().cuda_stream
        for lds, rnd, name in ((0, 0, "register-fed zero"), (0, 1, "register-fed random"), (1, 1, "LDS-fed random"), (2, 1, "LDS + DMA-fed random")):
            r = (C.c_double * 3)()
            smi.start()
            t0 = time.perf_counter()
            L.diag_check(D.mipnerf_mfma_ceiling(lds, 2, rnd, float(a.seconds), r, st), "ceiling")
            s = smi.stop(t0 + 0.5 * a.seconds)
            add(f"mipnerf_mfma_ceiling {name} (clock implied by issue rate {r[2]:.3f} GHz)", dict(tflops=r[0], ms=r[1], calls=0, **s))
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    with open(a.out, "w") as f:
        f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
