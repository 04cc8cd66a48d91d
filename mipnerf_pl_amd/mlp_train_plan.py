"""Static plan of the bf16 TRAINING kernels of the Mip-NeRF MLP (forward-with-save, dgrad, wgrad).

Reference semantics: torch autograd through models/mip_nerf.py:75-111.  With z_j = W_j a_j + b_j and
a_{j+1} = relu(z_j):   delta_j = (W_{j+1}^T delta_{j+1}) * [a_{j+1} > 0],   dW_j = delta_j a_j^T,
db_j = sum_samples delta_j.

Three kernels share the register dataflow of the inference kernel (mlp_plan.py): a wavefront owns 32
samples, lane (hi, n) holds 8 features of sample n per "k-step" register.

1. forward-with-save (generated, mlp_bf16_trainfwd_gen.hip): the inference schedule plus, per output tile,
   * the ReLU bit mask (1 bit per activation) and
   * the TRANSPOSED activations ("T-block"): two MFMAs against constant selection matrices turn the two
     k-step registers of a 32-feature tile (lane = sample) into a 32x32 D tile whose lane (hi', n) holds
     feature column n for 16 samples -- exactly the operand form the weight-gradient MFMA needs
     (contraction over samples).  Column n of a T-block <-> feature `colfeat(kind, n)` of the block;
     fragment f, lane-half hi', slot j <-> sample `frag_sample(f, hi', j)` of the wave's 32.
2. dgrad (generated, mlp_bf16_dgrad_gen.hip): the same streaming structure with W^T chunks; consumes
   d_raw and the bit masks, keeps delta in registers across layers and writes delta's T-blocks.
3. wgrad (kernels_wgrad.hip, table driven): per job a workgroup accumulates
   dW[out-block a][in-block b] += GT[a] * HT[b]^T over its share of the wave tiles (8 waves = 8 A-blocks,
   <= 8 B-blocks each, + one MFMA against ones for the bias), writes fp32 partials; a reduce kernel sums
   the splits and scatters into the flat gradient (index table below).

The bottleneck layer has no activation, so its two T-blocks are never stored: with M = sum_s delta_view x8^T
(one job, fp32, kept in a scratch region behind the parameters) the chain rule gives
dW_view[:, :W] = M W_extra^T + db_view b_extra^T,  dW_extra = W_view[:, :W]^T M,  db_extra = W_view[:, :W]^T db_view
(`post_process`, a 33-MFLOP fp32 kernel after the reduction) -- 16 of 157 T-blocks less to write and 24 less to read.

This module is the single source of truth for: block ids, the dgrad weight stream (pack table), the wgrad
job list and the partial -> parameter index table.  build.py dumps them to a binary blob that is linked
into the library (.incbin); tests emulate the whole dataflow in numpy against the oracle's gradients.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

from .mlp_plan import DLAYOUT, KSTEP, NATURAL, TILE, Arch, Plan, bf16_round

GROUP = 32           # chunks per LDS ring slot (must match the generators)
SLOTS = 2
WG_WAVES = 8
MAX_JOB_BLOCKS = 8   # A-blocks (= waves) and B-blocks per wgrad job
BIAS_SLOT = 8        # accumulator slot of the bias MFMA
JOB_SLOTS = 9
JOB_FLOATS = WG_WAVES * JOB_SLOTS * 64 * 16      # fp32 partials per (job, split)


def colfeat(kind: int, n: int) -> int:
    """feature (0..31, within its 32-feature block) held by column n of a T-block"""
    if kind == NATURAL:
        return n
    u, hs, j = n >> 4, (n >> 3) & 1, n & 7
    return 8 * (2 * u + (j >> 2)) + 4 * hs + (j & 3)


def frag_sample(f: int, hi: int, j: int) -> int:
    """sample (0..31 of the wave tile) in slot j of lane-half hi of fragment f of a T-block"""
    return Plan.drow(hi, 8 * f + j)


@dataclass
class BSeg:
    regset: str      # 'raw' | 'X' | 'Y'
    kind: int
    nk: int
    wt: int          # parameter tensor (the weight is used transposed: A[m][k] = W[r0 + c(k)][col0 + m])
    ld: int
    r0: int
    cmin: int
    cmax: int
    reg0: int = 0


@dataclass
class BOp:
    name: str
    segs: List[BSeg]
    ntiles: int
    col0: int
    mask: Optional[int]     # ReLU mask layer index, None = no activation
    out: str                # 'X' | 'Y'
    gblock: Optional[int]   # first G block id of the output tiles (None: the output is not stored)

    @property
    def nk(self):
        return sum(s.nk for s in self.segs)


@dataclass
class WJob:
    name: str
    a_blocks: List[int]
    b_blocks: List[int]
    rowmap: list            # [a_idx][32] -> (wt, row, ld) or None
    colmap: list            # [b_idx][32] -> col or -1
    biasmap: Optional[list] = None   # [a_idx][32] -> (bt, index) or None
    cost: float = 1.0       # relative HBM bytes per wave tile (for the split heuristic)
    b_src: int = 0          # 0: B blocks are T-blocks of the saved-activation buffer HT; 1 (pre-GEMM plans): blocks of the ENCODING
                            # FRAGMENT buffer the off-axis IPE kernel wrote (block b = its k-steps 2b, 2b+1: lane = sample, 8 features per
                            # lane), read through transposing LDS loads -- no T-block copy of the 672-wide encoding is ever written


@dataclass
class TrainPlan:
    fwd: Plan
    h_blocks: dict = field(default_factory=dict)     # name -> (first id, n, kind)
    g_blocks: dict = field(default_factory=dict)
    NH: int = 0
    NG: int = 0
    NMASK: int = 0
    fwd_out: list = field(default_factory=list)      # per fwd op: (H block0 or None, mask layer or None)
    bops: List[BOp] = field(default_factory=list)
    bchunks: list = field(default_factory=list)      # (op, tile, ks); padded with None
    jobs: List[WJob] = field(default_factory=list)
    pre_gemm: bool = False                           # training form of the two-kernel bf16 MLP (mlp_pre_plan.py), see build()
    NE: int = 0                                      # pre-GEMM plans: encoding blocks per wave tile in the fragment buffer

    @staticmethod
    def build(arch: Arch = None, pre_gemm: bool = False) -> "TrainPlan":
        """pre_gemm (round 5): the training kernels of the architecture whose encoding is too wide for k_mlp_bf16's wave-private LDS area
        (the unbounded-scene model, 672 features).  Forward-with-save = k_pre_gemm (unchanged: layer 0 and the encoding half of the skip
        layer) + a TRUNK forward-with-save that starts from the preloaded register set X = bf16(relu(layer 0)) -- whose T-blocks and ReLU
        mask it derives from those registers -- and initialises the skip layer's accumulators from k_pre_gemm's partial sums; the dgrad
        stream is the standard one (no gradient flows into the encoding: mip_nerf.py:83-90 concatenates a constant); the two weight
        matrices that multiply the encoding get weight-gradient jobs whose B blocks are the encoding FRAGMENTS (WJob.b_src = 1)."""
        fwd = Plan.build(arch or Arch(), pre_gemm=pre_gemm)
        a = fwd.arch
        if a.net_depth_condition < 1 or (a.net_depth_condition != 1 and pre_gemm):
            raise NotImplementedError("training kernels need at least one view layer (exactly one in the two-kernel form)")
        if a.xyz_dim % TILE:
            raise NotImplementedError("training kernels need xyz_dim to be a multiple of 32")
        if max(a.net_width, a.net_width_condition) > 256:
            raise NotImplementedError("training kernels are generated for widths <= 256 (two waves per SIMD; the 512-wide trunk has an inference kernel only)")
        tp = TrainPlan(fwd, pre_gemm=pre_gemm)
        D, W, Wc, E = a.net_depth, a.net_width, a.net_width_condition, a.xyz_dim
        nW, nC, nE = W // TILE, Wc // TILE, E // TILE
        names = [n for n, _ in a.param_shapes()]
        pid = {n: i for i, n in enumerate(names)}
        # ---- T-block ids ---------------------------------------------------------------------------
        hid = 0

        def hadd(name, n, kind):
            nonlocal hid
            tp.h_blocks[name] = (hid, n, kind)
            hid += n
        if pre_gemm:
            tp.NE = nE              # the encoding stays in the fragment buffer (its own block index space 0 .. nE-1)
        else:
            hadd("enc", nE, NATURAL)
        for i in range(1, D + 1):
            hadd(f"x{i}", nW, DLAYOUT)
        views = bool(a.use_viewdirs)       # False: MLP.forward(x, None) -- colour head on the trunk output, no bottleneck / view layer
        nV = a.net_depth_condition if views else 0     # view layers (mip_nerf.py:62-69); "hv" = the output of the LAST one (the colour head's input),
        if views:                                      # hv0 .. hv{nV-2} the ones before it (round 5: two view layers train in bf16 too)
            hadd("view", 1, NATURAL)
            for i in range(nV - 1):
                hadd(f"hv{i}", nC, DLAYOUT)
            hadd("hv", nC, DLAYOUT)
        tp.NH = hid
        gid = 0

        def gadd(name, n, kind):
            nonlocal gid
            tp.g_blocks[name] = (gid, n, kind)
            gid += n
        gadd("raw", 1, NATURAL)
        if views:
            for i in range(nV - 1, 0, -1):
                gadd(f"gv{i}", nC, DLAYOUT)      # delta at the pre-activation of view layer i
            gadd("gv", nC, DLAYOUT)              # ... of view layer 0 (the one that reads the bottleneck + view directions)
        for i in range(D, 0, -1):
            gadd(f"g{i}", nW, DLAYOUT)
        tp.NG = gid
        tp.NMASK = D + nV                        # mask row D + i = ReLU bits of view layer i
        # ---- what the forward-with-save kernel stores per op ---------------------------------------------
        for op in fwd.ops:
            if op.name.startswith("layer"):
                i = int(op.name[5:])
                tp.fwd_out.append((tp.h_blocks[f"x{i + 1}"][0], i))
            elif op.name == "head":
                tp.fwd_out.append((None, None))      # bottleneck: linear, never stored (see post_process)
            elif op.name.startswith("view"):
                i = int(op.name[4:])
                tp.fwd_out.append((tp.h_blocks["hv" if i == nV - 1 else f"hv{i}"][0], D + i))
            else:
                tp.fwd_out.append((None, None))
        # ---- dgrad ops -------------------------------------------------------------------------------------
        G = {k: v[0] for k, v in tp.g_blocks.items()}
        nrgb = a.num_rgb
        cur, other = "Y", "X"
        if views:
            tp.bops.append(BOp("dcolor", [BSeg("raw", NATURAL, 1, pid["color_layer.weight"], Wc, 0, 0, nrgb)],
                               nC, 0, D + nV - 1, "Y", G["gv" if nV == 1 else f"gv{nV - 1}"]))
            for i in range(nV - 1, 0, -1):       # through view layer i (Wc -> Wc): delta at the pre-activation of view layer i - 1
                tp.bops.append(BOp(f"dview{i}", [BSeg(cur, DLAYOUT, Wc // KSTEP, pid[f"view_layers.{i}.0.weight"], Wc, 0, 0, Wc)],
                                   nC, 0, D + i - 1, other, G["gv" if i == 1 else f"gv{i - 1}"]))
                cur, other = other, cur
            tp.bops.append(BOp("dview", [BSeg(cur, DLAYOUT, Wc // KSTEP, pid["view_layers.0.0.weight"],
                                              W + a.view_dim, 0, 0, Wc)], nW, 0, None, other, None))
            cur, other = other, cur
            tp.bops.append(BOp("dhead", [BSeg(cur, DLAYOUT, W // KSTEP, pid["extra_layer.weight"], W, 0, 0, W),
                                         BSeg("raw", NATURAL, 1, pid["density_layer.weight"], W, -nrgb, nrgb, nrgb + 1)],
                               nW, 0, D - 1, other, G[f"g{D}"]))
            cur, other = other, cur
        else:
            # both heads read the trunk output: delta_D = (d_rgb W_color + d_density W_density) * relu'(x_D); two k-steps on the
            # same d_raw register, one per weight tensor
            tp.bops.append(BOp("dhead", [BSeg("raw", NATURAL, 1, pid["color_layer.weight"], Wc, 0, 0, nrgb),
                                         BSeg("raw", NATURAL, 1, pid["density_layer.weight"], W, -nrgb, nrgb, nrgb + 1)],
                               nW, 0, D - 1, "Y", G[f"g{D}"]))
        shapes = dict(a.param_shapes())
        for i in range(D - 1, 0, -1):
            ld = shapes[f"layers.{i}.0.weight"][1]
            tp.bops.append(BOp(f"dlayer{i}", [BSeg(cur, DLAYOUT, W // KSTEP, pid[f"layers.{i}.0.weight"], ld, 0, 0, W)],
                               nW, 0, i - 1, other, G[f"g{i}"]))
            cur, other = other, cur
        for oi, op in enumerate(tp.bops):
            for t0 in range(0, op.ntiles, 2):
                t1 = t0 + 1 if t0 + 1 < op.ntiles else None
                for ks in range(op.nk):
                    tp.bchunks.append((oi, t0, ks))
                    if t1 is not None:
                        tp.bchunks.append((oi, t1, ks))
        tp.n_bchunks_real = len(tp.bchunks)
        while len(tp.bchunks) % (GROUP * SLOTS):
            tp.bchunks.append(None)
        # ---- wgrad jobs ---------------------------------------------------------------------------------
        H = tp.h_blocks

        def blocks(tab, name):
            b0, n, _ = tab[name]
            return list(range(b0, b0 + n))

        def rows_d(wt, ld, n):
            return [[(wt, TILE * ai + colfeat(DLAYOUT, m), ld) for m in range(32)] for ai in range(n)]

        def bias_d(bt, n):
            return [[(bt, TILE * ai + colfeat(DLAYOUT, m)) for m in range(32)] for ai in range(n)]

        def cols(name, col0, ncols):
            b0, n, kind = H[name]
            out = []
            for bi in range(n):
                row = []
                for c in range(32):
                    f = TILE * bi + colfeat(kind, c)
                    row.append(col0 + f if f < ncols else -1)
                out.append(row)
            return out

        def enc_jobs(name, gname, wname, bname, ld, col0):
            """pre-GEMM plans: delta (8 blocks) x the encoding's nE fragment blocks, <= 8 per job; natural feature order (column n of block
            b <-> feature 32 b + n); the bias gradient rides on the first job"""
            for c0 in range(0, nE, MAX_JOB_BLOCKS):
                bl = list(range(c0, min(c0 + MAX_JOB_BLOCKS, nE)))
                colmap = [[col0 + TILE * b + n for n in range(32)] for b in bl]
                tp.jobs.append(WJob(f"{name}.{c0 // MAX_JOB_BLOCKS}", blocks(tp.g_blocks, gname), bl, rows_d(pid[wname], ld, nW), colmap,
                                    bias_d(pid[bname], nW) if (bname is not None and c0 == 0) else None, cost=(nW + len(bl)) / 16, b_src=1))

        for i in range(D):
            wname, bname = f"layers.{i}.0.weight", f"layers.{i}.0.bias"
            ld = shapes[wname][1]
            src = "enc" if i == 0 else f"x{i}"
            ncols = E if i == 0 else W
            if pre_gemm and i == 0:
                enc_jobs("L0", "g1", wname, bname, ld, 0)
                continue
            tp.jobs.append(WJob(f"L{i}", blocks(tp.g_blocks, f"g{i + 1}"), blocks(H, src),
                                rows_d(pid[wname], ld, nW), cols(src, 0, ncols), bias_d(pid[bname], nW),
                                cost=(nW + len(blocks(H, src))) / 16))
            if ld > ncols:      # skip layer: the appended encoding columns
                if pre_gemm:
                    enc_jobs(f"L{i}e", f"g{i + 1}", wname, None, ld, W)
                else:
                    tp.jobs.append(WJob(f"L{i}e", blocks(tp.g_blocks, f"g{i + 1}"), blocks(H, "enc"),
                                        rows_d(pid[wname], ld, nW), cols("enc", W, E), None, cost=(nW + nE) / 16))
        _, nparams = fwd.param_offsets()
        raw_bias = [[(pid["color_layer.bias"], m) if m < nrgb else
                     ((pid["density_layer.bias"], 0) if m == nrgb else None) for m in range(32)]]
        if views:
            # scratch region behind the parameters: M = sum_s delta_view x8^T [Wc, W], then db_view of THIS call [Wc]
            tp.scratch_M, tp.scratch_dbv = nparams, nparams + Wc * W
            tp.n_scratch = Wc * W + Wc
            tp.post = dict(W=W, Wc=Wc, ldv=W + a.view_dim, extra_w=pid["extra_layer.weight"], extra_b=pid["extra_layer.bias"],
                           view_w=pid["view_layers.0.0.weight"], view_b=pid["view_layers.0.0.bias"])
            SCR = -1      # pseudo tensor id: row * ld + col is an offset into the scratch region
            m_rows = [[(SCR, TILE * ai + colfeat(DLAYOUT, m), W) for m in range(32)] for ai in range(nC)]
            m_bias = [[(SCR, Wc * W + TILE * ai + colfeat(DLAYOUT, m)) for m in range(32)] for ai in range(nC)]
            raw_rows_density = [[(pid["density_layer.weight"], 0, W) if m == nrgb else None for m in range(32)]]
            tp.jobs.append(WJob("M", blocks(tp.g_blocks, "gv") + blocks(tp.g_blocks, "raw"), blocks(H, f"x{D}"),
                                m_rows + raw_rows_density, cols(f"x{D}", 0, W), m_bias + raw_bias, cost=(nC + 1 + nW) / 16))
            vw = pid["view_layers.0.0.weight"]
            tp.jobs.append(WJob("viewd", blocks(tp.g_blocks, "gv"), blocks(H, "view"), rows_d(vw, W + a.view_dim, nC),
                                cols("view", W, a.view_dim), None, cost=(nC + 1) / 16))
            for i in range(1, nV):               # view layer i: delta_i x (output of view layer i - 1)
                src = f"hv{i - 1}"
                tp.jobs.append(WJob(f"view{i}", blocks(tp.g_blocks, f"gv{i}"), blocks(H, src), rows_d(pid[f"view_layers.{i}.0.weight"], Wc, nC),
                                    cols(src, 0, Wc), bias_d(pid[f"view_layers.{i}.0.bias"], nC), cost=(2 * nC) / 16))
            raw_rows_color = [[(pid["color_layer.weight"], m, Wc) if m < nrgb else None for m in range(32)]]
            tp.jobs.append(WJob("color", blocks(tp.g_blocks, "raw"), blocks(H, "hv"), raw_rows_color,
                                cols("hv", 0, Wc), None, cost=(1 + nC) / 16))
        else:
            # no bottleneck: nothing to post-process, no scratch; extra_layer / view_layers get no gradient (autograd leaves
            # them None; the C entry points zero them unless accumulating)
            tp.scratch_M = tp.scratch_dbv = nparams
            tp.n_scratch = 0
            tp.post = None
            head_rows = [[(pid["color_layer.weight"], m, Wc) if m < nrgb else
                          ((pid["density_layer.weight"], 0, W) if m == nrgb else None) for m in range(32)]]
            tp.jobs.append(WJob("heads", blocks(tp.g_blocks, "raw"), blocks(H, f"x{D}"), head_rows, cols(f"x{D}", 0, W),
                                raw_bias, cost=(1 + nW) / 16))
        for j in tp.jobs:
            assert len(j.a_blocks) <= MAX_JOB_BLOCKS and len(j.b_blocks) <= MAX_JOB_BLOCKS
        return tp

    # ---- helpers ---------------------------------------------------------------------------------------
    def bseg_of(self, op: BOp, ks: int):
        for s in op.segs:
            if ks < s.nk:
                return s, ks
            ks -= s.nk
        raise IndexError

    def bpack_table(self) -> np.ndarray:
        """int32 [n_bchunks, 64, 8]: flat parameter index of the dgrad stream element, -1 = zero."""
        offs, _ = self.fwd.param_offsets()
        tab = np.full((len(self.bchunks), 64, 8), -1, dtype=np.int32)
        for ci, ch in enumerate(self.bchunks):
            if ch is None:
                continue
            oi, t, ks = ch
            op = self.bops[oi]
            seg, ksl = self.bseg_of(op, ks)
            for hi in range(2):
                for j in range(8):
                    c = Plan.kmap(seg.kind, ksl, hi, j)
                    if not (seg.cmin <= c < seg.cmax):
                        continue
                    m = np.arange(32)
                    tab[ci, hi * 32 + m, j] = offs[seg.wt] + (seg.r0 + c) * seg.ld + op.col0 + 32 * t + m
        return tab

    def wgrad_out_table(self) -> np.ndarray:
        """int32 [njobs, 8 waves, 9 slots, 64 lanes, 16 regs]: flat parameter index fed by each fp32
        partial position, or -1.  Slot 8 = bias MFMA (column 0 of A x ones)."""
        offs, _ = self.fwd.param_offsets()
        tab = np.full((len(self.jobs), WG_WAVES, JOB_SLOTS, 64, 16), -1, dtype=np.int32)
        for ji, job in enumerate(self.jobs):
            for ai in range(len(job.a_blocks)):
                for hi in range(2):
                    for r in range(16):
                        m = Plan.drow(hi, r)
                        rm = job.rowmap[ai][m]
                        if rm is not None:
                            wt, row, ld = rm
                            base = self.scratch_M if wt < 0 else offs[wt]
                            for bi in range(len(job.b_blocks)):
                                for n in range(32):
                                    c = job.colmap[bi][n]
                                    if c >= 0:
                                        tab[ji, ai, bi, hi * 32 + n, r] = base + row * ld + c
                        if job.biasmap is not None and job.biasmap[ai][m] is not None:
                            bt, bidx = job.biasmap[ai][m]
                            tab[ji, ai, BIAS_SLOT, hi * 32 + 0, r] = (self.scratch_M if bt < 0 else offs[bt]) + bidx
        return tab

    def job_table(self) -> np.ndarray:
        """int32 [njobs, 20]: nA, nB, with_bias, b_src, a_blocks[8], b_blocks[8] (unused entries repeat the last)."""
        out = np.zeros((len(self.jobs), 20), dtype=np.int32)
        for ji, job in enumerate(self.jobs):
            nA, nB = len(job.a_blocks), len(job.b_blocks)
            out[ji, 0:4] = (nA, nB, int(job.biasmap is not None), job.b_src)
            for w in range(8):
                out[ji, 4 + w] = job.a_blocks[min(w, nA - 1)]
                out[ji, 12 + w] = job.b_blocks[min(w, nB - 1)]
        return out

    def job_splits(self, total_wgs: int) -> List[int]:
        """workgroups per job, proportional to the job's HBM bytes per wave tile (>= 1 each)."""
        cost = np.array([j.cost for j in self.jobs])
        raw = cost / cost.sum() * total_wgs
        sp = np.maximum(1, np.floor(raw)).astype(int)
        return [int(x) for x in sp]

    def blob(self) -> bytes:
        """Binary tables linked into libmipnerf_hip.so: header (int32 x 16) + sections."""
        bp = self.bpack_table().ravel()
        jt = self.job_table().ravel()
        ot = self.wgrad_out_table().ravel()
        hdr = np.zeros(16, dtype=np.int32)
        hdr[0] = 0x54524E31          # 'TRN1'
        hdr[1] = len(self.bchunks)
        hdr[2] = len(self.jobs)
        hdr[3] = self.NH
        hdr[4] = self.NG
        hdr[5] = self.NMASK | (self.NE << 16)      # NE: encoding fragment blocks per wave tile (pre-GEMM plans), else 0
        hdr[6] = JOB_FLOATS
        hdr[7] = self.fwd.param_offsets()[1]
        hdr[8] = bp.size
        hdr[9] = jt.size
        hdr[10] = ot.size
        hdr[11] = self.n_scratch
        offs, _ = self.fwd.param_offsets()
        po = self.post
        if po is not None:
            hdr[12:16] = (offs[po["extra_w"]], offs[po["extra_b"]], offs[po["view_w"]], offs[po["view_b"]])
        return hdr.tobytes() + bp.astype(np.int32).tobytes() + jt.astype(np.int32).tobytes() + ot.astype(np.int32).tobytes()


# ---- numpy emulation of the three kernels (tests only) ------------------------------------------------------
_LH = np.repeat(np.arange(2), 32)
_LN = np.tile(np.arange(32), 2)


def _mfma(a, b):
    """32x32x16 semantics on fragments [64, 8]: D[m, n] = sum_{hi, j} A[(hi, m), j] * B[(hi, n), j]."""
    return np.einsum("hmj,hnj->mn", a.reshape(2, 32, 8).astype(np.float64),
                     b.reshape(2, 32, 8).astype(np.float64)).astype(np.float32)


def _to_acc(D):
    """D [32, 32] -> accumulator registers [64 lanes, 16]: lane (hi, n) reg r = D[drow(hi, r), n]."""
    acc = np.zeros((64, 16), np.float32)
    for hi in range(2):
        for r in range(16):
            acc[hi * 32:(hi + 1) * 32, r] = D[Plan.drow(hi, r), :]
    return acc


def _selectors():
    P1 = np.zeros((64, 8), np.float32)
    P2 = np.zeros((64, 8), np.float32)
    for hi in range(2):
        for n in range(32):
            for j in range(8):
                P1[hi * 32 + n, j] = float(n == hi * 8 + j)
                P2[hi * 32 + n, j] = float(n == 16 + hi * 8 + j)
    return P1, P2


def tblock(r0, r1):
    """The two T fragments [2, 64, 8] the kernels derive from the k-step registers r0, r1 [64, 8] of a tile:
    D = r0 x P1 + r1 x P2 (register operand is the A matrix), fragment f = accumulator regs 8f..8f+7."""
    P1, P2 = _selectors()
    acc = _to_acc(_mfma(r0, P1) + _mfma(r1, P2))
    return np.stack([acc[:, 0:8], acc[:, 8:16]])


def pack_mask(r0, r1):
    """uint32 per lane: bit p = reg 2p of the tile is > 0, bit 16+p = reg 2p+1 (p = 0..7)."""
    regs = np.concatenate([r0, r1], axis=1) > 0          # [64, 16]
    m = np.zeros(64, np.uint32)
    for p in range(8):
        m |= regs[:, 2 * p].astype(np.uint32) << np.uint32(p)
        m |= regs[:, 2 * p + 1].astype(np.uint32) << np.uint32(16 + p)
    return m


def unpack_mask(word, t):
    """[64, 16] bool of tile t from the per-layer dwords [64, 4] (dword q = tiles 2q | 2q+1 << 8)."""
    w = (word[:, t >> 1] >> np.uint32(8 * (t & 1)))
    out = np.zeros((64, 16), bool)
    for p in range(8):
        out[:, 2 * p] = (w >> np.uint32(p)) & 1
        out[:, 2 * p + 1] = (w >> np.uint32(16 + p)) & 1
    return out


def emulate_train_tile(tp: TrainPlan, flat_params, enc, view, d_raw, valid, round_bf16=False, return_masks=False):
    """One wave tile (32 samples): forward-with-save + dgrad exactly as the generated kernels move data.
    enc [32, xyz], view [32, 32] (padded), d_raw [32, 4], valid [32] bool.
    Returns HT [NH, 2, 64, 8], GT [NG, 2, 64, 8], raw [32, 4]."""
    plan = tp.fwd
    rnd = bf16_round if round_bf16 else (lambda z: z.astype(np.float32))
    fp = np.concatenate([flat_params.astype(np.float32), np.zeros(1, np.float32)])
    stream = rnd(fp[plan.pack_table()])
    bias = fp[plan.bias_table()]

    def natural(src, nk):
        out = np.zeros((nk, 64, 8), np.float32)
        for ks in range(nk):
            for j in range(8):
                out[ks, :, j] = src[_LN, ks * 16 + _LH * 8 + j]
        return rnd(out)
    HT = np.zeros((tp.NH, 2, 64, 8), np.float32)
    masks = np.zeros((tp.NMASK, 64, 4), np.uint32)
    ET, pre_acc = None, None
    if tp.pre_gemm:
        # k_pre_gemm (unchanged) hands over X and the skip layer's accumulator images; the trunk kernel derives x1's T-blocks and the ReLU
        # mask of layer 0 from X (x1 = bf16(relu(.)) > 0 <=> the pre-activation was positive, up to values that round to zero in bf16);
        # the weight-gradient kernel reads the encoding as the B operand  [(hi, n), j] = enc[sample frag_sample(f, hi, j), feature 32 b + n]
        from .mlp_pre_plan import PrePlan, emulate_pre_gemm
        x, pre_acc = emulate_pre_gemm(PrePlan.build(plan.arch, split=False), flat_params, enc, round_bf16)
        regs = {"X": x, "view": natural(view, 2)}
        x1 = tp.h_blocks["x1"][0]
        for t in range(plan.arch.net_width // TILE):
            HT[x1 + t] = tblock(x[2 * t], x[2 * t + 1])
            masks[0, :, t >> 1] |= pack_mask(x[2 * t], x[2 * t + 1]) << np.uint32(8 * (t & 1))
        er = rnd(enc)
        ET = np.zeros((tp.NE, 2, 64, 8), np.float32)
        for b in range(tp.NE):
            for f in range(2):
                for j in range(8):
                    ET[b, f, :, j] = er[Plan.drow(_LH, 8 * f + j), 32 * b + _LN]
    else:
        regs = {"enc": natural(enc, plan.arch.xyz_dim // 16), "view": natural(view, 2)}
        e0 = tp.h_blocks["enc"][0]
        for b in range(plan.arch.xyz_dim // 32):
            HT[e0 + b] = tblock(regs["enc"][2 * b], regs["enc"][2 * b + 1])
    if "view" in tp.h_blocks:
        HT[tp.h_blocks["view"][0]] = tblock(regs["view"][0], regs["view"][1])
    ci = 0
    raw = np.zeros((32, 4), np.float32)
    for oi, op in enumerate(plan.ops):
        nt = len(op.tiles)
        acc = np.zeros((nt, 64, 16), np.float32)
        for ti in range(nt):
            acc[ti] = pre_acc[ti] if op.pre else bias[op.first_tile + ti][_LH]
        for (t0, t1) in plan.panels(op):
            for ks in range(op.nk):
                seg, ksl = plan.seg_of(op, ks)
                b = regs[seg.regset][seg.reg0 + ksl]
                for t in ((t0,) if t1 is None else (t0, t1)):
                    assert plan.chunks[ci] == (oi, t, ks)
                    acc[t] += _to_acc(_mfma(stream[ci], b))
                    ci += 1
        if op.relu:
            acc = np.maximum(acc, 0)
        if op.out in ("X", "Y"):
            nto = nt - (1 if op.name == "head" else 0)
            newreg = np.zeros((2 * nto, 64, 8), np.float32)
            for t in range(nto):
                newreg[2 * t] = acc[t, :, 0:8]
                newreg[2 * t + 1] = acc[t, :, 8:16]
            newreg = rnd(newreg)
            regs[op.out] = newreg
            hb, ml = tp.fwd_out[oi]
            for t in range(nto):
                if hb is not None:
                    HT[hb + t] = tblock(newreg[2 * t], newreg[2 * t + 1])
                if ml is not None:
                    masks[ml, :, t >> 1] |= pack_mask(newreg[2 * t], newreg[2 * t + 1]) << np.uint32(8 * (t & 1))
            if op.name == "head":
                raw[:, 3] = acc[nt - 1, 0:32, 0]
        else:
            for c in range(plan.arch.num_rgb):
                raw[:, c] = acc[0, 0:32, c]
    # ---- dgrad ----------------------------------------------------------------------------------------------
    bstream = rnd(fp[tp.bpack_table()])
    R = np.zeros((64, 8), np.float32)
    R[0:32, 0:4] = np.where(valid[:, None], d_raw, 0.0)
    R = rnd(R)
    GT = np.zeros((tp.NG, 2, 64, 8), np.float32)
    GT[tp.g_blocks["raw"][0]] = tblock(R, np.zeros_like(R))
    bregs = {"raw": R[None]}
    ci = 0
    for oi, op in enumerate(tp.bops):
        acc = np.zeros((op.ntiles, 64, 16), np.float32)
        for t0 in range(0, op.ntiles, 2):
            pair = (t0,) if t0 + 1 >= op.ntiles else (t0, t0 + 1)
            for ks in range(op.nk):
                seg, ksl = tp.bseg_of(op, ks)
                b = bregs[seg.regset][seg.reg0 + ksl]
                for t in pair:
                    assert tp.bchunks[ci] == (oi, t, ks)
                    acc[t] += _to_acc(_mfma(bstream[ci], b))
                    ci += 1
        newreg = np.zeros((2 * op.ntiles, 64, 8), np.float32)
        for t in range(op.ntiles):
            a_t = acc[t]
            if op.mask is not None:
                a_t = np.where(unpack_mask(masks[op.mask], t), a_t, 0.0)
            newreg[2 * t] = a_t[:, 0:8]
            newreg[2 * t + 1] = a_t[:, 8:16]
        newreg = rnd(newreg)
        bregs[op.out] = newreg
        if op.gblock is not None:
            for t in range(op.ntiles):
                GT[op.gblock + t] = tblock(newreg[2 * t], newreg[2 * t + 1])
    assert ci == tp.n_bchunks_real
    if return_masks:
        return HT, GT, raw, ET, masks
    if tp.pre_gemm:
        return HT, GT, raw, ET
    return HT, GT, raw


def emulate_wgrad(tp: TrainPlan, HT_all, GT_all, ET_all=None):
    """HT_all [ntiles, NH, 2, 64, 8], GT_all [ntiles, NG, 2, 64, 8] (pre-GEMM plans: ET_all [ntiles, NE, 2, 64, 8], the encoding blocks as
    the weight-gradient MFMA receives them) -> flat gradient (all parameters)."""
    _, nparams = tp.fwd.param_offsets()
    total = nparams + tp.n_scratch
    flat = np.zeros(total, np.float64)
    seen = np.zeros(total, np.int32)
    otab = tp.wgrad_out_table()
    ones = np.ones((64, 8), np.float32)
    for ji, job in enumerate(tp.jobs):
        part = np.zeros((WG_WAVES, JOB_SLOTS, 64, 16), np.float64)
        for wt in range(HT_all.shape[0]):
            for ai, ab in enumerate(job.a_blocks):
                for f in range(2):
                    A = GT_all[wt, ab, f]
                    for bi, bb in enumerate(job.b_blocks):
                        part[ai, bi] += _to_acc(_mfma(A, (ET_all if job.b_src else HT_all)[wt, bb, f]))
                    if job.biasmap is not None:
                        part[ai, BIAS_SLOT] += _to_acc(_mfma(A, ones))
        idx = otab[ji].ravel()
        ok = idx >= 0
        np.add.at(flat, idx[ok], part.ravel()[ok])
        np.add.at(seen, idx[ok], 1)
    return flat.astype(np.float32), seen


def post_process(tp: TrainPlan, flat_params, flat):
    """The fp32 chain-rule step that replaces the bottleneck T-blocks (kernels_wgrad.hip k_wgrad_post): consumes the
    scratch region of `flat` (M, db_view of this call), returns the parameter gradients [nparams]."""
    offs, nparams = tp.fwd.param_offsets()
    po = tp.post
    if po is None:            # no bottleneck (use_viewdirs=False): the partials are the gradients
        return flat[:nparams].copy()
    W, Wc, ldv = po["W"], po["Wc"], po["ldv"]
    fp = flat_params.astype(np.float32)
    We = fp[offs[po["extra_w"]]:offs[po["extra_w"]] + W * W].reshape(W, W)
    be = fp[offs[po["extra_b"]]:offs[po["extra_b"]] + W]
    Wv = fp[offs[po["view_w"]]:offs[po["view_w"]] + Wc * ldv].reshape(Wc, ldv)[:, :W]
    M = flat[tp.scratch_M:tp.scratch_M + Wc * W].reshape(Wc, W).astype(np.float32)
    dbv = flat[tp.scratch_dbv:tp.scratch_dbv + Wc].astype(np.float32)
    out = flat[:nparams].copy()
    dWv = out[offs[po["view_w"]]:offs[po["view_w"]] + Wc * ldv].reshape(Wc, ldv)
    dWv[:, :W] += M @ We.T + np.outer(dbv, be)
    out[offs[po["view_b"]]:offs[po["view_b"]] + Wc] += dbv
    out[offs[po["extra_w"]]:offs[po["extra_w"]] + W * W] += (Wv.T @ M).ravel()
    out[offs[po["extra_b"]]:offs[po["extra_b"]] + W] += Wv.T @ dbv
    return out


def emulate_train(tp: TrainPlan, flat_params, enc, view, d_raw, round_bf16=False):
    """enc [S, xyz], view [S, 32], d_raw [S, 4] (S arbitrary; padded to wave tiles like the kernels do)."""
    S = enc.shape[0]
    nt = (S + 31) // 32
    HTs, GTs, raws, ETs = [], [], [], []
    for t in range(nt):
        idx = np.minimum(np.arange(t * 32, t * 32 + 32), S - 1)
        valid = np.arange(t * 32, t * 32 + 32) < S
        res = emulate_train_tile(tp, flat_params, enc[idx], view[idx], d_raw[idx], valid, round_bf16)
        HTs.append(res[0])
        GTs.append(res[1])
        raws.append(res[2])
        if tp.pre_gemm:
            ETs.append(res[3])
    flat, seen = emulate_wgrad(tp, np.stack(HTs), np.stack(GTs), np.stack(ETs) if ETs else None)
    return post_process(tp, flat_params, flat), seen, np.concatenate(raws)[:S]
