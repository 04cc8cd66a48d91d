"""Plan of the two-kernel bf16 MLP for encodings too wide for the wave-private LDS area of k_mlp_bf16 (mlp_plan.py): the
unbounded-scene model's 672 off-axis IPE features = 42 k-steps per sample, 43 KiB of bf16 B operands per wave.

    reference: models/mip_nerf.py:75-111 (MLP.forward) -- layer 0 `relu(W0 enc + b0)` and, in the skip layer, the
    `W5[:, 256:] enc` half of `W5 cat(x, enc) + b5` are the only places the encoding enters.

Kernel 1, `k_pre_gemm` (csrc/gen_pre_gemm.py): the two contractions over the encoding, K-STEP-MAJOR -- a wave owns 32 samples and
all 8 output tiles of one matrix at a time (128 accumulator registers), so every encoding k-step is loaded ONCE per pass straight
from global memory into registers (one 16-byte vector per lane = one MFMA B operand), eight MFMAs per k-step; the weights stream
through the same 2 x 32-KiB LDS ring as k_mlp_bf16's.  Pass 0: W0, bias b0, ReLU, bf16 -> the register set X of the trunk kernel,
stored as 16 lane-linear 1-KiB fragments per wave tile.  Pass 1: W5[:, 256:], bias b5, fp32 -> accumulator images of the skip
layer (8 tiles x 4 KiB per wave tile).
Kernel 2: the trunk = k_mlp_bf16 generated from `Plan.build(arch, pre_gemm=True)`: starts at layer 1 with X preloaded, the skip
layer accumulates W5[:, :256] x on top of the images.

Arithmetic: the same products, the same fp32 accumulation, the same bf16 roundings as a single kernel would do -- only the ORDER
of the skip layer's sum differs (encoding part first).  `emulate_pre_wave` below is the numpy restatement the CPU tests hold
against the oracle; `PrePlan.blob()` the tables capi.hip packs the weights with.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Tuple

import numpy as np

from .mlp_plan import Arch, KSTEP, NATURAL, Plan, TILE, bf16_round

GROUP = 32           # chunks per ring slot (8 waves x 4), as in gen_mlp_bf16.py
RING_SLOTS = 3       # k_pre_gemm has the LDS for a three-slot ring (two groups in flight)
RING_MULTIPLE = GROUP * RING_SLOTS   # the stream is a whole number of ring revolutions, so the ring phase is the same for every tile
MAGIC = 0x50524731   # 'PRG1'
# k_pre_gemm's work split.  0 [default]: one matrix after the other over a 256-sample tile (the encoding is read twice: the kernel sits on the
# HBM roof, 9.1 GB per level); stream order [matrix][k-step][tile].  1: the two matrices go to the two HALVES of a workgroup -- waves 0-3 contract
# a 128-sample tile with W0, waves 4-7 the same samples with W_skip[:, 256:], so a tile's encoding comes from HBM once (the second wave finds it
# in L1 / L2); stream order [k-step][matrix][tile].  Built, parity-green, and SLOWER: 8.56-8.57 ms per forward against 8.27-8.31 (three alternating
# pairs, profiles/r04y_pre_gemm_split_ab.txt) -- every wave uses half of each ring group, so there is a ring barrier per 16 instead of 32 of its MFMAs
# and twice the L2 -> LDS weight traffic per sample, which costs more than the second HBM read of the encoding.  Kept as a knob (both forms are
# generated and checked by tests/test_pre_gemm_cpu.py).
SPLIT = __import__("os").environ.get("MLP_PRE_SPLIT", "0") == "1"


def supported(a: Arch) -> bool:
    """one skip layer, 256-wide trunk (8 output tiles per pass), an encoding that is a whole number of k-steps"""
    skips = [i for i in range(a.net_depth) if (i - 1) % a.skip_index == 0 and i > 1]
    return (not a.bf16_kernels and a.net_width == 256 and a.net_width_condition <= 256 and a.xyz_dim % KSTEP == 0 and a.xyz_dim > 96
            and len(skips) == 1 and a.use_viewdirs and a.net_depth_condition == 1)


@dataclass
class PrePlan:
    arch: Arch
    trunk: Plan = None
    fused: Plan = None                                      # the one-kernel form of the same model (round 6)
    skip_layer: int = 0
    nk: int = 0                                             # encoding k-steps
    chunks: List[Tuple[int, int, int]] = field(default_factory=list)      # (pass = matrix, ks, tile); pass -1: zero padding
    n_real_chunks: int = 0
    split: bool = True

    @staticmethod
    def build(arch: Arch, split: bool = None) -> "PrePlan":
        if not supported(arch):
            raise NotImplementedError("the pre-GEMM form is generated for fp32-only variants with a wide encoding, a 256-wide trunk and one skip layer")
        p = PrePlan(arch, trunk=Plan.build(arch, pre_gemm=True), fused=Plan.build(arch, fused=True), split=SPLIT if split is None else bool(split))
        p.skip_layer = [i for i in range(arch.net_depth) if (i - 1) % arch.skip_index == 0 and i > 1][0]
        p.nk = arch.xyz_dim // KSTEP
        nt = arch.net_width // TILE
        if p.split:
            p.chunks = [(ps, ks, t) for ks in range(p.nk) for ps in range(2) for t in range(nt)]
        else:
            p.chunks = [(ps, ks, t) for ps in range(2) for ks in range(p.nk) for t in range(nt)]
        p.n_real_chunks = len(p.chunks)
        while len(p.chunks) % RING_MULTIPLE:
            p.chunks.append((-1, 0, 0))
        return p

    @property
    def ntiles(self):
        return self.arch.net_width // TILE

    def _wb(self, ps):
        """(weight tensor index, bias tensor index, first weight column, leading dimension) of pass ps"""
        names = [n for n, _ in self.arch.param_shapes()]
        a = self.arch
        if ps == 0:
            return names.index("layers.0.0.weight"), names.index("layers.0.0.bias"), 0, a.xyz_dim
        i = self.skip_layer
        return names.index(f"layers.{i}.0.weight"), names.index(f"layers.{i}.0.bias"), a.net_width, a.net_width + a.xyz_dim

    def pack_table(self) -> np.ndarray:
        """int32 [n_chunks, 64, 8]: flat parameter index of the value at (chunk, lane (hi, m), slot j) = W[32 t + m][col0 + 16 ks + 8 hi + j]"""
        offs, _ = self.trunk.param_offsets()
        tab = np.full((len(self.chunks), 64, 8), -1, np.int32)
        m = np.arange(TILE)
        for ci, (ps, ks, t) in enumerate(self.chunks):
            if ps < 0:
                continue
            wt, _, col0, ld = self._wb(ps)
            for hi in range(2):
                for j in range(8):
                    c = Plan.kmap(NATURAL, ks, hi, j)
                    tab[ci, hi * 32 + m, j] = offs[wt] + (t * TILE + m) * ld + col0 + c
        return tab

    def bias_table(self) -> np.ndarray:
        """int32 [2 * ntiles, 2, 16]: accumulator image of (pass, tile): the bias of the row held by (lane-half, register)"""
        offs, _ = self.trunk.param_offsets()
        tab = np.full((2 * self.ntiles, 2, 16), -1, np.int32)
        for ps in range(2):
            _, bt, _, _ = self._wb(ps)
            for t in range(self.ntiles):
                for hi in range(2):
                    for r in range(16):
                        tab[ps * self.ntiles + t, hi, r] = offs[bt] + t * TILE + Plan.drow(hi, r)
        return tab

    def blob(self) -> bytes:
        """header (16 int32) + gemm pack table + gemm bias table + trunk pack table + trunk bias table + (round 6) the one-kernel form's pack
        and bias tables (flat parameter indices, -1 = zero); h[10] / h[11] = chunks / tiles of the one-kernel form"""
        gp, gb = self.pack_table().ravel(), self.bias_table().ravel()
        tp, tb = self.trunk.pack_table().ravel(), self.trunk.bias_table().ravel()
        fp, fb = self.fused.pack_table().ravel(), self.fused.bias_table().ravel()
        _, total = self.trunk.param_offsets()
        h = np.zeros(16, np.int32)
        h[:12] = [MAGIC, len(self.chunks), self.n_real_chunks, gb.size, len(self.trunk.chunks), self.trunk.n_real_chunks, self.trunk.n_tiles,
                  total, self.nk, self.ntiles, len(self.fused.chunks), self.fused.n_tiles]
        return b"".join(x.astype(np.int32).tobytes() for x in (h, gp, gb, tp, tb, fp, fb))


def emulate_pre_gemm(p: PrePlan, flat_params: np.ndarray, enc: np.ndarray, round_bf16: bool = False):
    """One wavefront (32 samples) of k_pre_gemm: chunks in stream order, MFMA 32x32x16 semantics.  enc [32, xyz_dim].
    Returns (X [16, 64, 8] = the trunk's preloaded register set, images [8, 64, 16] = the skip layer's accumulators)."""
    rnd = bf16_round if round_bf16 else (lambda z: z.astype(np.float32))
    fp = np.concatenate([flat_params.astype(np.float32), np.zeros(1, np.float32)])
    stream = rnd(fp[p.pack_table()])
    bias = fp[p.bias_table()]
    lanes_hi = np.repeat(np.arange(2), 32)
    lanes_n = np.tile(np.arange(32), 2)
    b = np.zeros((p.nk, 64, 8), np.float32)
    for ks in range(p.nk):
        for j in range(8):
            b[ks, :, j] = enc[lanes_n, ks * 16 + lanes_hi * 8 + j]
    b = rnd(b)
    nt = p.ntiles
    acc = np.zeros((2, nt, 64, 16), np.float32)
    for ps in range(2):
        for t in range(nt):
            acc[ps, t] = bias[ps * nt + t][lanes_hi]
    for ci, (ps, ks, t) in enumerate(p.chunks):
        if ps < 0:
            continue
        A = stream[ci].reshape(2, 32, 8)
        Bm = b[ks].reshape(2, 32, 8)
        D = np.einsum("hmj,hnj->mn", A.astype(np.float64), Bm.astype(np.float64)).astype(np.float32)
        for hi in range(2):
            for r in range(16):
                acc[ps, t, hi * 32:(hi + 1) * 32, r] += D[Plan.drow(hi, r), :]
    x = np.zeros((2 * nt, 64, 8), np.float32)
    a0 = np.maximum(acc[0], 0)
    for t in range(nt):
        x[2 * t] = a0[t, :, 0:8]
        x[2 * t + 1] = a0[t, :, 8:16]
    return rnd(x), acc[1]


def emulate_pre_wave(p: PrePlan, flat_params: np.ndarray, enc: np.ndarray, view: np.ndarray, round_bf16: bool = False):
    """both kernels for one wavefront: raw (rgb [32, 3], density [32])"""
    from .mlp_plan import emulate_wave
    x, images = emulate_pre_gemm(p, flat_params, enc, round_bf16)
    return emulate_wave(p.trunk, flat_params, None, view, round_bf16, pre_x=x, pre_acc=images)
