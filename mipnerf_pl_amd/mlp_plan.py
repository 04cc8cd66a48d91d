"""Static execution plan of the Mip-NeRF MLP for the gfx950 MFMA kernels.

The reference MLP (models/mip_nerf.py:14-111) is 8 x (Linear 256 + ReLU) with the 96-d
encoding concatenated after layer 4, a density head (256->1), a bottleneck (256->256, no
activation), one view layer (256+27 -> 128, ReLU) and a colour head (128->3).

bf16 kernel (mlp_bf16_gen.hip, generated from this plan by gen_mlp_bf16.py)
---------------------------------------------------------------------------
Every wavefront owns 32 samples and keeps their activations IN REGISTERS for the whole
network.  A layer is computed "swapped": D[out, sample] = W[out, k] * X^T[k, sample] with
v_mfma_f32_32x32x16_bf16, so

  * the A operand is a 32(out) x 16(k) slab of the weight matrix ("chunk", 1 KiB = one
    16-byte vector per lane),
  * the B operand is the wave's own activations: lane (hi, n) holds 8 k-values of sample n,
  * the D tile leaves lane (hi, n) holding output features (r&3) + 8*(r>>2) + 4*hi, r<16, of
    sample n -- which, after bias/ReLU/bf16 packing, IS a valid B operand of the next layer
    provided the next layer's weights are packed with the matching k-permutation
    ("dlayout" below).  No cross-lane movement, no LDS round trip for activations.

All weights of the network therefore form one linear stream of chunks in the exact order
every wave consumes them; the kernel DMAs that stream through an LDS ring shared by the 8
waves of a workgroup (global_load_lds_dwordx4), so each chunk is fetched from L2 once per
256 samples.  This module defines that order (`chunks`), the index table used to pack the
fp32 master weights into the bf16 stream, and the bias table layout.

fp32 kernel (kernels_mlp_f32.hip) keeps activations in LDS in natural feature order and
uses v_mfma_f32_32x32x2_f32; its packing is the same chunk idea with the natural k map and
8 fp32 per lane (2 KiB chunks), in [layer][tile][kblock] order (`f32_layers`).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Tuple

import numpy as np

import os

CHAIN = os.environ.get("MLP_CHAIN", "0") == "1"   # weight-stream order knob (inference kernel experiments)
RING_MULTIPLE = 64  # chunks: 2 ring slots x 32-chunk groups (gen_mlp_bf16.py)
TILE = 32          # MFMA M/N
KSTEP = 16         # MFMA K (bf16 32x32x16)
NATURAL, DLAYOUT = 0, 1

PARAM_ORDER_DOC = "layers.{0..D-1}.0.{weight,bias}, density_layer, extra_layer, view_layers.{i}.0, color_layer"


@dataclass
class Seg:
    """A run of k-steps of a layer input taken from one register set."""
    regset: str        # 'enc' | 'view' | 'X' | 'Y' | 'encg' (one-kernel form of wide encodings: streamed from global memory through a wave-private LDS ring)
    kind: int          # NATURAL | DLAYOUT
    nk: int            # k-steps (16 features each)
    col0: int          # first column of the torch weight this segment multiplies
    ncols: int         # valid columns (rest of the padded k-steps are zero weights)
    reg0: int = 0      # first k-step register of the set


@dataclass
class TileSrc:
    """Where the 32 output rows of one tile come from."""
    wt: int            # parameter-tensor index of the weight
    bt: int            # parameter-tensor index of the bias
    row0: int
    nrows: int         # valid rows (<= 32)
    ld: int            # in_features of that weight


@dataclass
class Op:
    name: str
    segs: List[Seg]
    tiles: List[TileSrc]
    relu: bool
    out: str           # register set written: 'X' | 'Y' | 'head' | 'rgb'
    first_tile: int = 0   # global tile index of tiles[0] (bias table row)
    pre: bool = False     # accumulators start from pre-activations computed by the pre-GEMM kernel (mlp_pre_plan.py), not from the bias
    kmajor: bool = False  # all output tiles of the op accumulate at once, chunks in [k-step][tile] order (Plan.fused: the ops that read the wide encoding)

    @property
    def nk(self):
        return sum(s.nk for s in self.segs)


@dataclass
class Arch:
    net_depth: int = 8
    net_width: int = 256
    net_depth_condition: int = 1
    net_width_condition: int = 128
    skip_index: int = 4
    num_rgb: int = 3
    num_density: int = 1
    xyz_dim: int = 96
    view_dim: int = 27
    use_viewdirs: bool = True      # False: MLP.forward(x, None) -- colour head on the trunk output (mip_nerf.py:99-110)
    feat_per_deg: int = 6          # encoding features per frequency: 6 = axis-aligned IPE (sin, cos) x 3; 42 = off-axis IPE on 21 directions
    bf16_kernels: bool = True      # False: plan tables + fp32 kernels only (an encoding too wide for the bf16 kernels' wave-private
                                   # LDS area: the 672 off-axis features of the unbounded-scene model); not part of key()

    def key(self):
        return (self.net_depth, self.net_width, self.net_depth_condition, self.net_width_condition, self.skip_index,
                self.num_rgb, self.num_density, self.xyz_dim, self.view_dim, int(self.use_viewdirs))

    def param_shapes(self):
        shapes = []
        for i in range(self.net_depth):
            if i == 0:
                din = self.xyz_dim
            elif (i - 1) % self.skip_index == 0 and i > 1:
                din = self.net_width + self.xyz_dim
            else:
                din = self.net_width
            shapes += [(f"layers.{i}.0.weight", (self.net_width, din)), (f"layers.{i}.0.bias", (self.net_width,))]
        shapes += [("density_layer.weight", (self.num_density, self.net_width)),
                   ("density_layer.bias", (self.num_density,)),
                   ("extra_layer.weight", (self.net_width, self.net_width)),
                   ("extra_layer.bias", (self.net_width,))]
        for i in range(self.net_depth_condition):
            din = self.net_width + self.view_dim if i == 0 else self.net_width_condition
            shapes += [(f"view_layers.{i}.0.weight", (self.net_width_condition, din)),
                       (f"view_layers.{i}.0.bias", (self.net_width_condition,))]
        shapes += [("color_layer.weight", (self.num_rgb, self.net_width_condition)),
                   ("color_layer.bias", (self.num_rgb,))]
        return shapes


def _ceil(a, b):
    return (a + b - 1) // b


@dataclass
class Plan:
    arch: Arch
    ops: List[Op] = field(default_factory=list)
    chunks: List[Tuple[int, int, int]] = field(default_factory=list)   # (op, tile_in_op, ks); op = -1: zero padding
    n_tiles: int = 0
    n_real_chunks: int = 0
    pre_gemm: bool = False     # trunk of the two-kernel bf16 form (mlp_pre_plan.py): no encoding segments, layer 0 done elsewhere
    fused: bool = False        # ONE-kernel form of a wide encoding (round 6): layer 0 and the skip layer are k-step-major ops with all 8 output
                               # tiles live, their encoding k-steps streamed global -> wave-private LDS ring -> B operand

    # ---- construction -----------------------------------------------------------------
    @staticmethod
    def build(arch: Arch = None, pre_gemm: bool = False, fused: bool = False) -> "Plan":
        """pre_gemm: the TRUNK of the two-kernel bf16 form used for encodings too wide for the wave-private LDS area (mlp_pre_plan.py):
        layer 0 and the encoding part of the skip layer are a separate k-step-major GEMM kernel; this plan starts at layer 1 with the
        register set X preloaded from memory (bf16(relu(layer 0))) and the skip layer's accumulators initialised from the GEMM's fp32
        partial sums (Op.pre) instead of the bias."""
        a = arch or Arch()
        assert not (pre_gemm and fused)
        wmax = 256 if (pre_gemm or fused) else 512      # (above 256 the bf16 kernel runs one wave per SIMD: gen_mlp_bf16.waves_of)
        if a.net_width % TILE or a.net_width_condition % TILE or a.net_width > wmax or a.net_width_condition > wmax:
            raise NotImplementedError("MFMA kernels need widths that are multiples of 32 and <= 512 (<= 256 for the two-kernel trunk form)")
        if a.xyz_dim % KSTEP or a.view_dim > 32 or a.num_rgb > 4 or a.num_density != 1:
            raise NotImplementedError("unsupported encoding / head size for the MFMA kernels")
        if a.net_depth >= 2 and (a.net_depth - 1) % a.skip_index == 0 and a.net_depth - 1 > 0:
            raise NotImplementedError("skip concat after the last trunk layer (reference would fail too)")
        if not a.use_viewdirs and a.net_width_condition != a.net_width:
            raise NotImplementedError("use_viewdirs=False feeds the trunk output (net_width) to color_layer "
                                      "(net_width_condition inputs): the reference fails unless the two widths are equal")
        p = Plan(a, pre_gemm=pre_gemm, fused=fused)
        encset = "encg" if fused else "enc"
        names = [n for n, _ in a.param_shapes()]
        pid = {n: i for i, n in enumerate(names)}
        W, E = a.net_width, a.xyz_dim
        cur, other = None, "X"
        for i in range(a.net_depth):
            segs = []
            is_skip = (i - 1) % a.skip_index == 0 and i > 1
            if i == 0:
                if pre_gemm:                       # X = bf16(relu(layer 0)) arrives from the pre-GEMM kernel
                    cur, other = "X", "Y"
                    continue
                segs.append(Seg(encset, NATURAL, E // KSTEP, 0, E))
                ld = E
            else:
                segs.append(Seg(cur, DLAYOUT, W // KSTEP, 0, W))
                ld = W
                if is_skip:
                    if fused:
                        # encoding part FIRST: the order of the two-kernel form (k_pre_gemm's partial sums, then the trunk's 16 k-steps on
                        # top), so both forms add the same products in the same order and agree bit for bit
                        segs.insert(0, Seg(encset, NATURAL, E // KSTEP, W, E))
                    elif not pre_gemm:
                        segs.append(Seg("enc", NATURAL, E // KSTEP, W, E))
                    ld = W + E
            tiles = [TileSrc(pid[f"layers.{i}.0.weight"], pid[f"layers.{i}.0.bias"], t * TILE, TILE, ld)
                     for t in range(W // TILE)]
            p.ops.append(Op(f"layer{i}", segs, tiles, True, other, pre=pre_gemm and is_skip, kmajor=fused and (i == 0 or is_skip)))
            cur, other = other, ("Y" if other == "X" else "X")
        # head: bottleneck (no activation) + density row as an extra tile; without view directions only the density row
        # (extra_layer and view_layers stay unused parameters, mip_nerf.py:99-110)
        tiles = [TileSrc(pid["extra_layer.weight"], pid["extra_layer.bias"], t * TILE, TILE, W)
                 for t in range(W // TILE)] if a.use_viewdirs else []
        tiles.append(TileSrc(pid["density_layer.weight"], pid["density_layer.bias"], 0, a.num_density, W))
        p.ops.append(Op("head", [Seg(cur, DLAYOUT, W // KSTEP, 0, W)], tiles, False, other))
        if a.use_viewdirs:
            cur, other = other, cur
        Wc = a.net_width_condition
        for i in range(a.net_depth_condition if a.use_viewdirs else 0):
            if i == 0:
                segs = [Seg(cur, DLAYOUT, W // KSTEP, 0, W), Seg("view", NATURAL, 2, W, a.view_dim)]
                ld = W + a.view_dim
            else:
                segs = [Seg(cur, DLAYOUT, Wc // KSTEP, 0, Wc)]
                ld = Wc
            tiles = [TileSrc(pid[f"view_layers.{i}.0.weight"], pid[f"view_layers.{i}.0.bias"], t * TILE, TILE, ld)
                     for t in range(Wc // TILE)]
            p.ops.append(Op(f"view{i}", segs, tiles, True, other))
            cur, other = other, cur
        p.ops.append(Op("color", [Seg(cur, DLAYOUT, Wc // KSTEP, 0, Wc)],
                        [TileSrc(pid["color_layer.weight"], pid["color_layer.bias"], 0, a.num_rgb, Wc)],
                        False, "rgb"))
        # chunk order: panels of two tiles interleaved per k-step, odd tile alone (CHAIN: tile after tile, i.e.
        # back-to-back MFMAs on the SAME accumulator -- an experiment knob of the inference kernel, see gen_mlp_bf16.py)
        gt = 0
        for oi, op in enumerate(p.ops):
            op.first_tile = gt
            gt += len(op.tiles)
            if op.kmajor:                          # [k-step][tile]: every k-step's B operand feeds all tiles of the op back to back
                for ks in range(op.nk):
                    for t in range(len(op.tiles)):
                        p.chunks.append((oi, t, ks))
                continue
            for (t0, t1) in p.panels(op):
                if CHAIN:
                    for t in (t0, t1):
                        if t is not None:
                            for ks in range(op.nk):
                                p.chunks.append((oi, t, ks))
                    continue
                for ks in range(op.nk):
                    p.chunks.append((oi, t0, ks))
                    if t1 is not None:
                        p.chunks.append((oi, t1, ks))
        p.n_tiles = gt
        # the kernel streams whole ring groups and needs a stable tile-to-tile ring phase: pad the stream with zero chunks
        p.n_real_chunks = len(p.chunks)
        while len(p.chunks) % RING_MULTIPLE:
            p.chunks.append((-1, 0, 0))
        return p

    @staticmethod
    def panels(op: Op):
        n = len(op.tiles)
        out = [(t, t + 1) for t in range(0, n - 1, 2)]
        if n % 2:
            out.append((n - 1, None))
        return out

    # ---- k maps -----------------------------------------------------------------------
    @staticmethod
    def kmap(kind: int, ks_local: int, hi: int, j: int) -> int:
        """feature index held by lane-half `hi`, slot j of k-step ks_local of a segment"""
        if kind == NATURAL:
            return ks_local * 16 + hi * 8 + j
        t, u = ks_local >> 1, ks_local & 1
        return 32 * t + 8 * (2 * u + (j >> 2)) + 4 * hi + (j & 3)

    @staticmethod
    def drow(hi: int, r: int) -> int:
        """row of a 32x32 D tile held by lane-half hi in accumulator register r"""
        return (r & 3) + 8 * (r >> 2) + 4 * hi

    def seg_of(self, op: Op, ks: int):
        for s in op.segs:
            if ks < s.nk:
                return s, ks
            ks -= s.nk
        raise IndexError

    # ---- tables -----------------------------------------------------------------------
    def param_offsets(self):
        offs, o = [], 0
        for _, shp in self.arch.param_shapes():
            offs.append(o)
            o += int(np.prod(shp))
        return offs, o

    def pack_table(self) -> np.ndarray:
        """int32 [n_chunks, 64, 8]: flat index into the concatenated fp32 parameters of the
        value that goes to (chunk, lane, slot), or -1 for zero padding."""
        offs, _ = self.param_offsets()
        tab = np.full((len(self.chunks), 64, 8), -1, dtype=np.int32)
        for ci, (oi, ti, ks) in enumerate(self.chunks):
            if oi < 0:
                continue
            op = self.ops[oi]
            tile = op.tiles[ti]
            seg, ksl = self.seg_of(op, ks)
            for hi in range(2):
                for j in range(8):
                    c = self.kmap(seg.kind, ksl, hi, j)
                    if c >= seg.ncols:
                        continue
                    col = seg.col0 + c
                    m = np.arange(tile.nrows)
                    tab[ci, hi * 32 + m, j] = offs[tile.wt] + (tile.row0 + m) * tile.ld + col
        return tab

    def bias_table(self) -> np.ndarray:
        """int32 [n_tiles, 2, 16]: flat parameter index of the bias added to accumulator
        register r of lane-half hi for global tile g, or -1."""
        offs, _ = self.param_offsets()
        tab = np.full((self.n_tiles, 2, 16), -1, dtype=np.int32)
        for op in self.ops:
            if op.pre:                  # the bias is part of the pre-activations the accumulators start from
                continue
            for ti, tile in enumerate(op.tiles):
                for hi in range(2):
                    for r in range(16):
                        row = self.drow(hi, r)
                        if row < tile.nrows:
                            tab[op.first_tile + ti, hi, r] = offs[tile.bt] + tile.row0 + row
        return tab

    # ---- fp32 kernel layout (natural k order, activations in LDS) -----------------------
    def f32_layers(self):
        """Per layer of the LDS-resident fp32 kernel (kernels_mlp_f32.hip): K segments (x_in0, kb0), (x_in1, kb1), the
        output column x_out, tiles, relu, kind.  LDS row layout per sample: cols [0,W) activation buffer B, [W,2W) buffer A
        (layers alternate: layer i writes A when i is even), [2W, 2W+max(xyz,32)) the encoding, later the padded view
        features; `runs` = (x_col0, w_col0, ncols) per segment in k order."""
        a = self.arch
        W = a.net_width
        enc_col = 2 * W
        layers = []
        cur = None                      # LDS column of the buffer holding the current activation
        for oi, op in enumerate(self.ops):
            out_col = 0 if cur == W else W
            cols = []
            for s in op.segs:
                cols.append((enc_col if s.regset in ("enc", "view") else cur, s.col0, s.ncols))
            ktot = sum(_ceil(c[2], KSTEP) * KSTEP for c in cols)
            segs = [(c[0], _ceil(c[2], KSTEP)) for c in cols] + [(0, 0)]
            kind = 1 if op.name == "head" else (2 if op.out == "rgb" else 0)
            layers.append(dict(name=op.name, x_in0=segs[0][0], kb0=segs[0][1], x_in1=segs[1][0], kb1=segs[1][1],
                               x_out=out_col, kb=ktot // KSTEP, tiles=op.tiles, relu=op.relu, runs=cols,
                               first_tile=op.first_tile, x_in=segs[0][0], kind=kind,
                               stage_view=int(kind == 1 and a.use_viewdirs)))
            if kind == 0 or (kind == 1 and len(op.tiles) > 1):
                cur = out_col           # a head that is only the density row leaves the trunk output where it is
        return layers

    def pack_table_f32(self) -> np.ndarray:
        """int32 [n_chunks_f32, 64, 8] for the fp32 stream: chunk order [layer][tile][kb],
        natural k map: lane (hi, m), slot j <-> W[row0+m][col(kb*16 + hi*8 + j)]."""
        offs, _ = self.param_offsets()
        rows = []
        for L in self.f32_layers():
            # natural column list of this layer input (padded per run to 16)
            colmap = []
            for (_, wcol0, n) in L["runs"]:
                pad = _ceil(n, KSTEP) * KSTEP
                colmap += [wcol0 + c if c < n else -1 for c in range(pad)]
            for tile in L["tiles"]:
                for kb in range(L["kb"]):
                    tab = np.full((64, 8), -1, dtype=np.int32)
                    for hi in range(2):
                        for j in range(8):
                            col = colmap[kb * 16 + hi * 8 + j]
                            if col < 0:
                                continue
                            m = np.arange(tile.nrows)
                            tab[hi * 32 + m, j] = offs[tile.wt] + (tile.row0 + m) * tile.ld + col
                    rows.append(tab)
        return np.stack(rows)


# ---- numpy emulation of the bf16 kernel's dataflow (used by tests, not by the product) -----
def bf16_round(x: np.ndarray) -> np.ndarray:
    """float32 -> nearest-even bfloat16, returned as float32."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    r = ((u >> 16) & 1) + np.uint32(0x7FFF)
    out = ((u + r) & np.uint32(0xFFFF0000)).astype(np.uint32)
    return out.view(np.float32)


def emulate_wave(plan: Plan, flat_params: np.ndarray, enc: np.ndarray, view: np.ndarray,
                 round_bf16: bool = False, pre_x: np.ndarray = None, pre_acc: np.ndarray = None):
    """Run the register-level dataflow of one wavefront (32 samples) exactly as the generated
    kernel does: stream chunks in plan order, MFMA 32x32x16 semantics, D -> B repacking.
    enc [32, xyz_dim], view [32, 32] (already padded).  Returns raw (rgb[32,3], density[32]).
    Trunk plans (Plan.pre_gemm): pre_x [16, 64, 8] = the preloaded register set X, pre_acc [tiles, 64, 16] = the accumulator images
    of the skip layer, both as the pre-GEMM kernel leaves them (mlp_pre_plan.emulate_pre_gemm); enc is not read."""
    rnd = bf16_round if round_bf16 else (lambda z: z.astype(np.float32))
    ptab = plan.pack_table()
    btab = plan.bias_table()
    fp = np.concatenate([flat_params.astype(np.float32), np.zeros(1, np.float32)])
    stream = rnd(fp[ptab])                        # [-1] -> the appended zero
    bias = fp[btab]                               # [tiles, 2, 16]
    lanes_hi = np.repeat(np.arange(2), 32)
    lanes_n = np.tile(np.arange(32), 2)
    regs = {}
    # natural-order B operands
    def natural(src, nk):
        out = np.zeros((nk, 64, 8), np.float32)
        for ks in range(nk):
            for j in range(8):
                out[ks, :, j] = src[lanes_n, ks * 16 + lanes_hi * 8 + j]
        return rnd(out)
    if plan.pre_gemm:
        regs["X"] = pre_x
    else:
        regs["encg" if plan.fused else "enc"] = natural(enc, plan.arch.xyz_dim // 16)
    regs["view"] = natural(view, 2)
    ci = 0
    result = {}
    for op in plan.ops:
        nt = len(op.tiles)
        acc = np.zeros((nt, 64, 16), np.float32)
        for ti in range(nt):
            acc[ti] = pre_acc[ti] if op.pre else bias[op.first_tile + ti][lanes_hi]      # accumulator initialised with bias
        for tiles_of_panel in ([tuple(range(nt))] if op.kmajor else [((t0,) if t1 is None else (t0, t1)) for (t0, t1) in plan.panels(op)]):
            for ks in range(op.nk):
                seg, ksl = plan.seg_of(op, ks)
                b = regs[seg.regset][seg.reg0 + ksl]               # [64, 8]
                for t in tiles_of_panel:
                    assert plan.chunks[ci] == (plan.ops.index(op), t, ks)
                    a = stream[ci]
                    ci += 1
                    # D[m, n] += sum_{hi, j} A[(hi, m), j] * B[(hi, n), j]
                    A = a.reshape(2, 32, 8)
                    Bm = b.reshape(2, 32, 8)
                    D = np.einsum("hmj,hnj->mn", A.astype(np.float64), Bm.astype(np.float64)).astype(np.float32)
                    for hi in range(2):
                        for r in range(16):
                            acc[t, hi * 32:(hi + 1) * 32, r] += D[Plan.drow(hi, r), :]
        if op.relu:
            acc = np.maximum(acc, 0)
        if op.out in ("X", "Y") or op.name == "head":
            ntile_out = nt - (1 if op.name == "head" else 0)
            newreg = np.zeros((2 * ntile_out, 64, 8), np.float32)
            for t in range(ntile_out):
                newreg[2 * t] = acc[t, :, 0:8]
                newreg[2 * t + 1] = acc[t, :, 8:16]
            if ntile_out:
                regs[op.out] = rnd(newreg)
            if op.name == "head":
                result["density"] = acc[nt - 1, 0:32, 0].copy()     # lanes hi=0, register 0 -> row 0
        else:
            result["rgb"] = np.stack([acc[0, 0:32, r] for r in range(plan.arch.num_rgb)], axis=-1)
    assert ci == plan.n_real_chunks and all(c[0] < 0 for c in plan.chunks[ci:])
    return result["rgb"], result["density"]
