"""Drop-in `MipNerf` / `MLP` modules with the constructor signatures, parameter names / shapes
(state_dict keys) and `forward` contracts of the reference (models/mip_nerf.py:14-248), whose
compute runs in the hand-written gfx950 kernels of libmipnerf_hip.so.

    model = MipNerf(**same_kwargs_as_reference).cuda()
    ret = model(rays, randomized, white_bkgd)     # list of (comp_rgb, distance, acc, weights, t_samples)

The torch modules below only OWN the fp32 master parameters (so checkpoints, optimizers and
DDP see ordinary nn.Parameters); no torch op touches activations on the hot path.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

from . import _lib as L
from . import ops
from .rays import Rays

_PREC = {"bf16": L.PREC_BF16, "bfloat16": L.PREC_BF16, "fp32": L.PREC_FP32, "float32": L.PREC_FP32}


def _xavier_init(linear):
    torch.nn.init.xavier_uniform_(linear.weight.data)


def param_layout(arch: dict, use_viewdirs: bool = True):
    """[(name, (out, in) or (out,), column split)] of the reference MLP's parameters in state_dict order (mip_nerf.py:19-73).
    `column split` = width of the FIRST part of a concatenated input ([trunk | encoding] of a skip layer, [bottleneck | view
    encoding] of the first view layer), else None."""
    w, wc, x, v = arch["net_width"], arch["net_width_condition"], arch["xyz_dim"], arch["view_dim"]
    out = []
    for i in range(arch["net_depth"]):
        if i == 0:
            din, split = x, None
        elif (i - 1) % arch["skip_index"] == 0 and i > 1:
            din, split = w + x, w
        else:
            din, split = w, None
        out += [(f"layers.{i}.0.weight", (w, din), split), (f"layers.{i}.0.bias", (w,), None)]
    out += [("density_layer.weight", (arch["num_density_channels"], w), None), ("density_layer.bias", (arch["num_density_channels"],), None),
            ("extra_layer.weight", (w, w), None), ("extra_layer.bias", (w,), None)]
    for i in range(arch["net_depth_condition"]):
        din, split = (w + v, w) if i == 0 else (wc, None)
        out += [(f"view_layers.{i}.0.weight", (wc, din), split), (f"view_layers.{i}.0.bias", (wc,), None)]
    out += [("color_layer.weight", (arch["num_rgb_channels"], wc), None), ("color_layer.bias", (arch["num_rgb_channels"],), None)]
    return out


class WidthPadding:
    """An MLP whose widths are not among the generated kernel shapes runs on the smallest generated shape that CONTAINS it
    (same depth, skip period, view-layer count and encodings; net_width / net_width_condition rounded up): the true parameters
    are scattered into zero-initialised tensors of the generated shape.  The padded network computes the same function -- a
    padded unit has zero weights and bias, outputs relu(0) = 0 and feeds zero weights -- and its gradient w.r.t. every padded
    entry is exactly zero (no upstream delta passes relu'(0) = 0, and the inputs it multiplies are 0), so the gradient of the
    true parameters is a gather of the padded gradient.  `index` holds, for every element of the concatenated true parameters,
    its position in the concatenated padded parameters."""

    def __init__(self, arch: dict, padded_arch: dict, use_viewdirs: bool = True):
        self.arch, self.padded_arch = dict(arch), dict(padded_arch)
        true_l, pad_l = param_layout(arch, use_viewdirs), param_layout(padded_arch, use_viewdirs)
        assert [t[0] for t in true_l] == [t[0] for t in pad_l]
        idx, off = [], 0
        self.padded_shapes, self.padded_offsets = [], []
        for (name, ts, tsplit), (_, ps, psplit) in zip(true_l, pad_l):
            if len(ts) == 1:
                assert ts[0] <= ps[0], name
                flat = torch.arange(ts[0])
            else:
                assert ts[0] <= ps[0] and ts[1] <= ps[1], name
                col = torch.arange(ts[1])
                if tsplit is not None:                      # [first part | rest]: the rest starts behind the PADDED first part
                    col = torch.where(col < tsplit, col, col + (psplit - tsplit))
                flat = (torch.arange(ts[0])[:, None] * ps[1] + col[None, :]).reshape(-1)
            idx.append(flat + off)
            self.padded_shapes.append(tuple(ps))
            self.padded_offsets.append(off)
            off += int(torch.Size(ps).numel())
        self.index = torch.cat(idx)
        self.true_numel, self.padded_numel = int(self.index.numel()), off

    def scatter(self, params, out_flat: torch.Tensor) -> None:
        """true parameters -> their places in the (zero elsewhere) flat padded buffer"""
        out_flat[self.index] = torch.cat([p.detach().reshape(-1) for p in params])

    def gather(self, padded_flat: torch.Tensor) -> torch.Tensor:
        return padded_flat[self.index]

    def padded_views(self, flat: torch.Tensor):
        return [flat[o:o + int(torch.Size(s).numel())].view(s) for o, s in zip(self.padded_offsets, self.padded_shapes)]

    def to(self, device):
        self.index = self.index.to(device)
        return self


def containing_variant(arch: dict, use_viewdirs: bool, unbounded: bool, want_bf16: bool):
    """The generated architecture variant an MLP of shape `arch` runs on: (variant arch dict, exact: bool), or None.  Exact match
    first; else the smallest variant with the same structure and net_width / net_width_condition >= the asked ones (with
    use_viewdirs=False the colour layer reads the trunk, so both stay equal).  Prefers variants that have kernels for the asked
    precision."""
    lib = L.lib()
    best = None
    for v in range(int(lib.mipnerf_num_variants())):
        cfg, has_bf16 = L.Config(), C.c_int(0)
        L.check(lib.mipnerf_variant_arch(v, C.byref(cfg), C.byref(has_bf16)), "variant_arch")
        same = (cfg.net_depth == arch["net_depth"] and cfg.skip_index == arch["skip_index"] and
                cfg.net_depth_condition == arch["net_depth_condition"] and cfg.num_rgb_channels == arch["num_rgb_channels"] and
                cfg.num_density_channels == arch["num_density_channels"] and bool(cfg.use_viewdirs) == bool(use_viewdirs) and
                bool(cfg.unbounded) == bool(unbounded) and
                (42 if cfg.unbounded else 6) * (cfg.max_deg_point - cfg.min_deg_point) == arch["xyz_dim"] and
                3 + 6 * cfg.deg_view == arch["view_dim"])
        if not same or cfg.net_width < arch["net_width"] or cfg.net_width_condition < arch["net_width_condition"]:
            continue
        exact = cfg.net_width == arch["net_width"] and cfg.net_width_condition == arch["net_width_condition"]
        if not use_viewdirs and not exact and cfg.net_width != cfg.net_width_condition:
            continue
        key = (not exact, bool(want_bf16) and not has_bf16.value, cfg.net_width * cfg.net_width + cfg.net_width * cfg.net_width_condition)
        if best is None or key < best[0]:
            padded = dict(arch, net_width=int(cfg.net_width), net_width_condition=int(cfg.net_width_condition))
            best = (key, padded, exact)
    return None if best is None else (best[1], best[2])


class NativeContext:
    """Owns one `mipnerf_ctx` (configuration + packed weight streams on the current device)."""

    def __init__(self, cfg: L.Config, device: torch.device, padding: Optional[WidthPadding] = None):
        self.device = device
        self.cfg = cfg
        self.padding = padding.to(device) if padding is not None else None
        self._padded_flat = None
        self._h = C.c_void_p()
        with torch.cuda.device(device):
            L.check(L.lib().mipnerf_create(C.byref(cfg), C.byref(self._h)), "mipnerf_create")
        self._packed_key = None
        self._ws = None

    def __del__(self):
        try:
            if getattr(self, "_h", None) is not None and self._h.value:
                L.lib().mipnerf_destroy(self._h)
                self._h = None
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    def sync_params(self, params) -> None:
        """Re-pack the MFMA operand streams when any master parameter changed (in-place updates bump
        tensor._version; load_state_dict / .to() change version or storage)."""
        key = tuple((p.data_ptr(), p._version) for p in params)
        if key == self._packed_key:
            return
        want = int(L.lib().mipnerf_num_param_tensors(self._h))
        if len(params) != want:
            raise NotImplementedError(f"expected {want} parameter tensors for this architecture, got {len(params)}")
        if self.padding is not None:
            # widths between the generated shapes: the kernels see zero-padded copies (WidthPadding); one cat + one scatter
            for i, p in enumerate(params):
                if not p.is_cuda or p.device != self.device or p.dtype != torch.float32:
                    raise RuntimeError(f"parameter {i} must be float32 on {self.device} (is {p.dtype} on {p.device})")
            with torch.cuda.device(self.device):
                if self._padded_flat is None:
                    self._padded_flat = torch.zeros(self.padding.padded_numel, device=self.device, dtype=torch.float32)
                self.padding.scatter(params, self._padded_flat)
            params = self.padding.padded_views(self._padded_flat)
        keep = []
        arr = (C.c_void_p * want)()
        for i, p in enumerate(params):
            if not p.is_cuda or p.device != self.device:
                raise RuntimeError(f"parameter {i} is on {p.device}, context is on {self.device}")
            d = p.detach()
            if d.dtype != torch.float32:
                raise TypeError("master parameters must be float32")
            if not d.is_contiguous():
                # the library keeps these pointers as the fp32 master weights of the backward kernels
                raise RuntimeError(f"parameter {i} is not contiguous; the native path needs contiguous master parameters")
            keep.append(d)
            arr[i] = d.data_ptr()
        with torch.cuda.device(self.device):
            L.check(L.lib().mipnerf_set_params(self._h, arr, ops._stream()), "mipnerf_set_params")
        self._keep = keep
        self._packed_key = key

    def workspace(self, num_rays: int) -> torch.Tensor:
        need = int(L.lib().mipnerf_workspace_bytes(self._h, num_rays))
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws

    def train_sizes(self, num_points: int):
        """(act, masks, delta, partials) buffer sizes in bytes of the native MLP training step."""
        v = [C.c_size_t() for _ in range(4)]
        L.check(L.lib().mipnerf_mlp_train_sizes(self._h, num_points, *[C.byref(x) for x in v]), "mlp_train_sizes")
        return tuple(int(x.value) for x in v)

    def scratch(self, name: str, nbytes: int) -> torch.Tensor:
        """Cached device scratch buffer (grown on demand, reused across steps)."""
        if not hasattr(self, "_scratch"):
            self._scratch = {}
        t = self._scratch.get(name)
        if t is None or t.numel() < nbytes:
            # MIPNERF_ZERO_SCRATCH=1: timing experiments whose kernels leave parts of a buffer unwritten (build.py: WRONG_RESULT_KNOBS)
            alloc = torch.zeros if os.environ.get("MIPNERF_ZERO_SCRATCH") == "1" else torch.empty
            t = alloc(nbytes, dtype=torch.uint8, device=self.device)
            self._scratch[name] = t
        return t

    def grad_numel(self, shapes) -> int:
        """Elements of the flat gradient buffer the native backward writes (the generated shape's, which is the parameters' own
        unless the widths are padded)."""
        if self.padding is not None:
            return self.padding.padded_numel
        return sum(int(torch.Size(s).numel()) for s in shapes)

    def split_grads(self, grad_flat: torch.Tensor, shapes):
        """flat native gradient -> one tensor per parameter, in the parameters' own shapes"""
        if self.padding is not None:
            grad_flat = self.padding.gather(grad_flat)
        grads, off = [], 0
        for shp in shapes:
            n = int(torch.Size(shp).numel())
            grads.append(grad_flat[off:off + n].view(shp))
            off += n
        return grads

    def set_option(self, option: int, value: int) -> None:
        L.check(L.lib().mipnerf_set_option(self._h, option, value), "mipnerf_set_option")


class MLP(torch.nn.Module):
    """Parameter container + stand-alone entry to the MFMA MLP kernel (models/mip_nerf.py:14-111)."""

    def __init__(self, net_depth: int, net_width: int, net_depth_condition: int, net_width_condition: int,
                 skip_index: int, num_rgb_channels: int, num_density_channels: int, activation: str,
                 xyz_dim: int, view_dim: int):
        super().__init__()
        if activation != "relu":
            raise NotImplementedError  # mip_nerf.py:50,70
        self.skip_index = skip_index
        self.arch = dict(net_depth=net_depth, net_width=net_width, net_depth_condition=net_depth_condition,
                         net_width_condition=net_width_condition, skip_index=skip_index,
                         num_rgb_channels=num_rgb_channels, num_density_channels=num_density_channels,
                         xyz_dim=xyz_dim, view_dim=view_dim)
        layers = []
        for i in range(net_depth):
            if i == 0:
                dim_in = xyz_dim
            elif (i - 1) % skip_index == 0 and i > 1:
                dim_in = net_width + xyz_dim
            else:
                dim_in = net_width
            linear = torch.nn.Linear(dim_in, net_width)
            _xavier_init(linear)
            layers.append(torch.nn.Sequential(linear, torch.nn.ReLU(True)))
        self.layers = torch.nn.ModuleList(layers)
        self.density_layer = torch.nn.Linear(net_width, num_density_channels)
        _xavier_init(self.density_layer)
        self.extra_layer = torch.nn.Linear(net_width, net_width)
        _xavier_init(self.extra_layer)
        layers = []
        for i in range(net_depth_condition):
            dim_in = net_width + view_dim if i == 0 else net_width_condition
            linear = torch.nn.Linear(dim_in, net_width_condition)
            _xavier_init(linear)
            layers.append(torch.nn.Sequential(linear, torch.nn.ReLU(True)))
        self.view_layers = torch.nn.Sequential(*layers)
        self.color_layer = torch.nn.Linear(net_width_condition, num_rgb_channels)   # default init (mip_nerf.py:73)
        self._ctx: Optional[NativeContext] = None
        self._cfg_extra = {}
        self.precision = L.PREC_BF16

    # -- flat parameter / gradient storage (opt-in: FlatAdam, FlatGradAllReduce, zero-copy wgrad reduction) -------
    def flatten_parameters(self):
        """Re-home the 24 parameters (and their .grad) as views of ONE flat fp32 buffer each, in state_dict order.
        Call after the module is on its device.  Names, shapes, leaf-ness and state_dict() are unchanged.

        Flat mode is for this package's own loop (FlatAdam, FlatGradAllReduce, GraphedTrainStep): the native backward
        writes the gradient into the flat buffer as a SIDE EFFECT and returns no per-parameter gradients to autograd, so
        AccumulateGrad nodes, tensor hooks and therefore torch's DistributedDataParallel reducer never fire, and
        torch.autograd.grad() returns None for these parameters.  Under DDP / Lightning's DDP strategy keep the default
        (non-flat) mode: there the backward returns ordinary gradients.  A parameter with a tensor hook makes the backward
        fall back to returned gradients automatically."""
        params = self.ordered_params()
        dev = params[0].device
        e = self._cfg_extra
        found = containing_variant(self.arch, bool(e.get("use_viewdirs", 1)), bool(e.get("unbounded", 0)), self.precision == L.PREC_BF16) \
            if dev.type == "cuda" else None
        if found is not None and not found[1]:
            raise NotImplementedError("flat parameter mode (FlatAdam / GraphedTrainStep / zero-copy gradient reduction) needs an MLP of a "
                                      "generated shape; this one runs zero-padded on a wider one (csrc/gen_mlp_bf16.py: VARIANTS)")
        total = sum(p.numel() for p in params)
        flat = torch.empty(total, device=dev, dtype=torch.float32)
        gflat = torch.zeros(total, device=dev, dtype=torch.float32)
        off = 0
        for p in params:
            n = p.numel()
            flat[off:off + n].copy_(p.detach().reshape(-1))
            p.data = flat[off:off + n].view(p.shape)
            p.grad = gflat[off:off + n].view(p.shape)
            off += n
        self._flat_param, self._flat_grad = flat, gflat
        self._flat_grad_valid = False
        self.invalidate_packed()
        return self

    def is_flat(self) -> bool:
        flat = getattr(self, "_flat_param", None)
        if flat is None:
            return False
        off = 0
        for p in self.ordered_params():
            if p.data_ptr() != flat.data_ptr() + 4 * off:
                return False
            off += p.numel()
        return True

    def grads_are_flat(self) -> bool:
        g = getattr(self, "_flat_grad", None)
        if g is None or not self.is_flat():
            return False
        off = 0
        for p in self.ordered_params():
            if p.grad is None or p.grad.data_ptr() != g.data_ptr() + 4 * off:
                return False
            off += p.numel()
        return True

    def gather_foreign_grads(self):
        """If autograd / a caller replaced some p.grad (e.g. zero_grad(set_to_none=True) followed by a backward through
        the generic path), copy those into the flat buffer and re-attach the views."""
        g = getattr(self, "_flat_grad", None)
        if g is None:
            return
        if all(p.grad is None for p in self.ordered_params()):
            self._flat_grad_valid = False       # somebody reset the gradients (zero_grad(set_to_none=True)): nothing accumulated
        off, any_grad = 0, False
        for p in self.ordered_params():
            n = p.numel()
            view = g[off:off + n].view(p.shape)
            if p.grad is not None and p.grad.data_ptr() != view.data_ptr():
                view.copy_(p.grad)
                any_grad = True
            if p.grad is None or p.grad.data_ptr() != view.data_ptr():
                p.grad = view
            off += n
        if any_grad:
            self._flat_grad_valid = True

    def invalidate_packed(self):
        """Force a re-pack of the MFMA weight streams (parameters changed without bumping tensor versions)."""
        if self._ctx is not None:
            self._ctx._packed_key = None

    # -- native context ------------------------------------------------------------------------
    def ordered_params(self):
        """state_dict order of the reference MLP = order mipnerf_set_params expects."""
        return list(self.parameters())

    def native(self, device: torch.device) -> NativeContext:
        if self._ctx is None or self._ctx.device != device:
            a = self.arch
            deg_point = a["xyz_dim"] // (42 if self._cfg_extra.get("unbounded", 0) else 6)
            deg_view = (a["view_dim"] - 3) // 6
            e = self._cfg_extra
            padding = None
            found = containing_variant(a, bool(e.get("use_viewdirs", 1)), bool(e.get("unbounded", 0)), self.precision == L.PREC_BF16)
            if found is not None and not found[1]:
                padding = WidthPadding(a, found[0], bool(e.get("use_viewdirs", 1)))
                if self.is_flat():               # (flatten_parameters' own guard only runs when the module already sat on the GPU)
                    raise NotImplementedError("flat parameter mode needs an MLP of a generated shape; this one runs zero-padded on "
                                              f"{found[0]['net_width']} / {found[0]['net_width_condition']}: the native backward would "
                                              "write the padded gradient past the end of the true-size flat buffer")
                import warnings
                warnings.warn(f"MLP {a['net_width']} / {a['net_width_condition']} is not a generated shape: it runs zero-padded on the "
                              f"{found[0]['net_width']} / {found[0]['net_width_condition']} kernels at "
                              f"{padding.padded_numel / padding.true_numel:.2f}x the multiply-adds, plus one cat + scatter of all parameters "
                              "per optimiser step (add the shape to csrc/gen_mlp_bf16.py: VARIANTS and rebuild to avoid both)", stacklevel=3)
                a = found[0]                     # the context is created for the containing generated shape
            cfg = L.Config(
                num_samples=e.get("num_samples", 128), num_levels=e.get("num_levels", 2),
                min_deg_point=e.get("min_deg_point", 0), max_deg_point=e.get("max_deg_point", deg_point),
                deg_view=e.get("deg_view", deg_view), use_viewdirs=e.get("use_viewdirs", 1),
                disparity=e.get("disparity", 0), disable_integration=e.get("disable_integration", 0),
                net_depth=a["net_depth"], net_width=a["net_width"], net_depth_condition=a["net_depth_condition"],
                net_width_condition=a["net_width_condition"], skip_index=a["skip_index"],
                num_rgb_channels=a["num_rgb_channels"], num_density_channels=a["num_density_channels"],
                resample_padding=e.get("resample_padding", 0.01), density_bias=e.get("density_bias", -1.0),
                rgb_padding=e.get("rgb_padding", 0.001), density_noise=e.get("density_noise", 0.0), unbounded=e.get("unbounded", 0))
            self._ctx = NativeContext(cfg, device, padding)
        self._ctx.sync_params(self.ordered_params())
        return self._ctx

    def forward(self, x, view_direction=None, precision: Optional[int] = None, return_activated: bool = False):
        """x: [B, N, xyz_dim] encodings, view_direction: [B, view_dim] -> (raw_rgb [B,N,3], raw_density [B,N,1])."""
        if (view_direction is None) != (not self._cfg_extra.get("use_viewdirs", 1)):
            raise ValueError("MLP.forward: view_direction must be given exactly when the model was built with use_viewdirs=True "
                             "(the native context is specialised on it)")
        prec = self.precision if precision is None else precision
        dt = torch.bfloat16 if prec == L.PREC_BF16 else torch.float32
        if not x.is_cuda:
            raise RuntimeError("MLP.forward needs HIP device tensors; there is no CPU fallback")
        B, N, _ = x.shape
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()) and not return_activated:
            # differentiable w.r.t. the parameters (like the reference module under autograd): native bf16 training
            # kernels, or torch's fp32 Linear ops in parity mode (autograd.py)
            from .autograd import mlp_native, mlp_native_f32
            venc = torch.zeros(B, 32, device=x.device, dtype=dt)
            if view_direction is not None:
                venc[:, :view_direction.shape[-1]] = view_direction.to(dt)
            raw = mlp_native(self, x.to(dt), venc) if prec == L.PREC_BF16 else mlp_native_f32(self, x.to(dt), venc)
            return raw[..., :3], raw[..., 3:4]
        ctx = self.native(x.device)
        enc = x.to(dt).contiguous()
        venc = torch.zeros(B, 32, device=x.device, dtype=dt)
        if view_direction is not None:
            venc[:, :view_direction.shape[-1]] = view_direction.to(dt)
        rgb_sigma = torch.empty(B, N, 4, device=x.device, dtype=torch.float32)
        raw = torch.empty_like(rgb_sigma)
        L.check(L.lib().mipnerf_mlp_forward(ctx.handle, B * N, N, enc.data_ptr(), venc.data_ptr(), prec,
                                            rgb_sigma.data_ptr(), raw.data_ptr(), ops._stream()), "mlp_forward")
        if return_activated:
            return raw[..., :3], raw[..., 3:4], rgb_sigma
        return raw[..., :3], raw[..., 3:4]


class MipNerf(torch.nn.Module):
    """Nerf NN Model with both coarse and fine MLPs (reference: models/mip_nerf.py:114-248).

    Extra keywords (not in the reference): `precision` = 'bf16' (default; bf16 MFMA with fp32
    accumulation, BASELINE configs[1]) or 'fp32' (exact-fp32 MFMA, parity mode / configs[3]);
    `unbounded=True` = the unbounded-scene (mip-NeRF 360) path the reference's dead code aims at
    (mip.py:106-124 sample_along_rays_360, :292-319 integrated_pos_enc_360, :424-447 contract / parameterization): fence posts
    uniform in inverse depth, fine-level resampling over the inverse-depth fence posts, conical frustums lifted to
    FULL-covariance Gaussians, contracted into the radius-2 ball, encoded with the off-axis IPE on 21 directions -- 42 features
    per degree, so the MLP's first layer / skip concat are 42 * (max_deg_point - min_deg_point) wide (672 for 16 degrees).  Default
    precision fp32 (forward, rendering and training through autograd); precision='bf16' runs the MLP as two kernels (csrc/gen_pre_gemm.py +
    a trunk kernel; forward about seven times faster) and, since round 5, TRAINS through autograd on bf16 kernels too (forward-with-save,
    dgrad, weight-gradient jobs over the encoding; 7.4 ms against 51 ms per 4096-ray step).  `disparity` / `disable_integration` do not apply."""

    def __init__(self, num_samples: int = 128, num_levels: int = 2, resample_padding: float = 0.01,
                 stop_resample_grad: bool = True, use_viewdirs: bool = True, disparity: bool = False,
                 ray_shape: str = 'cone', min_deg_point: int = 0, max_deg_point: int = 16, deg_view: int = 4,
                 density_activation: str = 'softplus', density_noise: float = 0., density_bias: float = -1.,
                 rgb_activation: str = 'sigmoid', rgb_padding: float = 0.001, disable_integration: bool = False,
                 append_identity: bool = True, mlp_net_depth: int = 8, mlp_net_width: int = 256,
                 mlp_net_depth_condition: int = 1, mlp_net_width_condition: int = 128, mlp_skip_index: int = 4,
                 mlp_num_rgb_channels: int = 3, mlp_num_density_channels: int = 1, mlp_net_activation: str = 'relu',
                 precision: Optional[str] = None, unbounded: bool = False):
        super().__init__()
        self.unbounded = bool(unbounded)
        self.num_levels = num_levels
        self.num_samples = num_samples
        self.disparity = disparity
        self.ray_shape = ray_shape
        self.disable_integration = disable_integration
        self.min_deg_point = min_deg_point
        self.max_deg_point = max_deg_point
        self.use_viewdirs = use_viewdirs
        self.deg_view = deg_view
        self.density_noise = density_noise
        self.density_bias = density_bias
        self.resample_padding = resample_padding
        self.stop_resample_grad = stop_resample_grad
        self.rgb_padding = rgb_padding
        if ray_shape != 'cone':
            raise NotImplementedError  # mip.py:97-98
        if rgb_activation != 'sigmoid' or density_activation != 'softplus':
            raise NotImplementedError  # mip_nerf.py:165,170
        if not use_viewdirs and mlp_net_width_condition != mlp_net_width:
            # MLP.forward(x, None) feeds the trunk output to color_layer (mip_nerf.py:99-110): a shape error in the reference
            raise NotImplementedError("use_viewdirs=False needs mlp_net_width_condition == mlp_net_width "
                                      "(color_layer reads the trunk output; the reference fails on other shapes too)")
        mlp_xyz_dim = (max_deg_point - min_deg_point) * (42 if unbounded else 3 * 2)
        if unbounded:
            if disparity or disable_integration:
                raise NotImplementedError("unbounded=True samples in inverse depth and always integrates (disparity / disable_integration do not apply)")
            if not stop_resample_grad:
                raise NotImplementedError("unbounded=True implements the shipped stop-gradient resampler")
            # fp32 unless asked otherwise.  precision='bf16': the 672-wide encoding runs as k_pre_gemm + a trunk kernel (csrc/gen_pre_gemm.py),
            # in inference and (round 5) in training: under autograd and in the one-call native step (train_step_native)
            precision = precision or os.environ.get("MIPNERF_PRECISION", "fp32")
        mlp_view_dim = deg_view * 3 * 2
        mlp_view_dim = mlp_view_dim + 3 if append_identity else mlp_view_dim
        if not append_identity:
            raise NotImplementedError("append_identity=False: forward always appends (mip_nerf.py:224), as here")
        if not stop_resample_grad and (precision or os.environ.get("MIPNERF_PRECISION", "bf16")) in ("bf16", "bfloat16"):
            # inference is unaffected; training with the resampler in the graph needs dL/d(encoding), which only the fp32 path has
            import warnings
            warnings.warn("MipNerf(stop_resample_grad=False, precision='bf16'): forward / rendering work, but TRAINING with the "
                          "resampler in the autograd graph needs precision='fp32' (the first backward-enabled forward raises "
                          "NotImplementedError)", stacklevel=2)
        self.mlp = MLP(mlp_net_depth, mlp_net_width, mlp_net_depth_condition, mlp_net_width_condition,
                       mlp_skip_index, mlp_num_rgb_channels, mlp_num_density_channels, mlp_net_activation,
                       mlp_xyz_dim, mlp_view_dim)
        precision = precision or os.environ.get("MIPNERF_PRECISION", "bf16")
        if precision not in _PREC:
            raise ValueError(f"precision must be one of {sorted(_PREC)}")
        self.precision = _PREC[precision]
        self.mlp.precision = self.precision
        self.mlp._cfg_extra = dict(num_samples=num_samples, num_levels=num_levels, min_deg_point=min_deg_point,
                                   max_deg_point=max_deg_point, deg_view=deg_view, use_viewdirs=int(use_viewdirs),
                                   disparity=int(disparity), disable_integration=int(disable_integration),
                                   resample_padding=resample_padding, density_bias=density_bias,
                                   rgb_padding=rgb_padding, density_noise=float(density_noise), unbounded=int(self.unbounded))

    def _density_randn(self, randomized, B, dev, density_randn):
        """mip_nerf.py:232-233: standard-normal draws [num_levels, B, N] when randomized and density_noise > 0 (the
        reference draws them on the CPU; here torch's device generator), else None."""
        if not (randomized and self.density_noise > 0):
            return None
        if density_randn is None:
            return torch.randn(self.num_levels, B, self.num_samples, device=dev)
        z = ops._f32c(density_randn, "density_randn")
        if z.numel() != self.num_levels * B * self.num_samples:
            raise ValueError("density_randn must have num_levels x B x num_samples elements")
        return z

    def set_precision(self, precision: str) -> "MipNerf":
        """Switch the arithmetic of all later calls: 'bf16' | 'fp32'.  The native context packs the weight streams of both precisions, so
        nothing is rebuilt.  Typical use: a model trained in one precision rendered in the other (e.g. fp32 parity training, bf16 rendering)
        (`model.eval(); model.set_precision('bf16')` -- about six times faster)."""
        if precision not in _PREC:
            raise ValueError(f"precision must be one of {sorted(_PREC)}")
        self.precision = _PREC[precision]
        self.mlp.precision = self.precision
        return self

    def forward(self, rays: Rays, randomized: bool, white_bkgd: bool, t_rand=None, u_rand=None, density_randn=None):
        """rays: Rays of [B,k] float32 HIP tensors.  Returns [(comp_rgb [B,3], distance [B], acc [B],
        weights [B,N], t_samples [B,N+1])] * num_levels (mip_nerf.py:246).  `t_rand` / `u_rand` / `density_randn`
        optionally inject the random draws of the randomized path (tests)."""
        o = rays.origins
        if not o.is_cuda:
            raise RuntimeError("MipNerf.forward needs rays on a HIP device; there is no CPU fallback "
                               "(the CPU restatement lives in oracle/ and is test-only)")
        with torch.cuda.device(o.device):      # native launches go to the CURRENT device's stream
            if o.shape[0] == 0:
                return self._forward_empty(o.device)
            if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
                # (round 5: MipNerf(unbounded=True, precision='bf16') trains through this route too -- k_pre_gemm + a trunk forward-with-save,
                # the standard dgrad, weight-gradient jobs over the row-major encoding -- and through the one-call train_step_native)
                from .autograd import mipnerf_forward_train
                return mipnerf_forward_train(self, rays, randomized, white_bkgd, t_rand, u_rand, density_randn)
            return self._forward_native(rays, randomized, white_bkgd, t_rand, u_rand, density_randn)

    def _forward_empty(self, dev):
        """Zero rays (an empty shard of a partitioned frame, an empty last chunk): the reference's torch ops run on empty
        tensors and return empty per-level tuples (mip_nerf.py:172-248); the C ABI rejects B < 1, so the host answers.
        Under autograd the outputs hang off the parameters, so a `.backward()` on this rank leaves zero gradients and a
        data-parallel wrapper sees the same set of reduced tensors as on the other ranks."""
        N = self.num_samples
        z = None
        if torch.is_grad_enabled():
            ps = [p for p in self.parameters() if p.requires_grad]
            if ps:
                z = torch.stack([p.reshape(-1)[:1].sum() for p in ps]).sum() * 0.0
        ret = []
        for _ in range(self.num_levels):
            tens = [torch.zeros(shape, device=dev) for shape in ((0, 3), (0,), (0,), (0, N), (0, N + 1))]
            if z is not None:
                tens = [t + z for t in tens[:4]] + tens[4:]         # the fence posts carry no gradient (stop-gradient resampler)
            ret.append(tuple(tens))
        return ret

    def native_step_supported(self, device) -> bool:
        """True when the one-call native training step (mipnerf_train_step) serves this model on `device`: bf16 precision, the shipped
        stop-gradient resampler, an MLP shape with generated bf16 training kernels."""
        if self.precision != L.PREC_BF16 or not self.stop_resample_grad or torch.device(device).type != "cuda":
            return False
        try:
            ctx = self.mlp.native(torch.device(device))
        except NotImplementedError:
            return False
        return int(L.lib().mipnerf_train_workspace_bytes(ctx.handle, 64)) > 0

    def _train_step_call(self, rays: Rays, gt_rgb, randomized, white_bkgd, coarse_loss_mult, distloss_mult, disable_multiscale_loss,
                         t_rand, u_rand, density_randn, grad, accumulate, return_outputs=False):
        """One mipnerf_train_step launch sequence: loss scalars [6] + the flat gradient written (accumulate = 0) or added (1) into `grad`."""
        dev = rays.origins.device
        B, N = rays.origins.shape[0], self.num_samples
        ctx = self.mlp.native(dev)
        f = [ops._f32c(getattr(rays, k), k) for k in Rays._fields]
        rp = L.RaysPtrs(*[t.data_ptr() for t in f])
        gt = ops._f32c(gt_rgb[..., :3], "gt_rgb")
        if randomized:
            t_rand = torch.rand(B, N + 1, device=dev) if t_rand is None else ops._f32c(t_rand, "t_rand")
            u_rand = torch.rand(B, N + 1, device=dev) if u_rand is None else ops._f32c(u_rand, "u_rand")
        dz = self._density_randn(randomized, B, dev, density_randn)
        need = int(L.lib().mipnerf_train_workspace_bytes(ctx.handle, B))
        if need == 0:
            raise NotImplementedError("the one-call native training step has no bf16 training kernels for this MLP shape "
                                      "(csrc/gen_mlp_train.py); it trains through autograd (MipNerf.forward + loss.backward())")
        ws = ctx.scratch("train_step", need)
        scalars = torch.empty(6, device=dev, dtype=torch.float32)
        outs, ret = None, None
        if return_outputs:
            outs = (L.LevelOut * self.num_levels)()
            ret = []
            for lvl in range(self.num_levels):
                tens = (torch.empty(B, 3, device=dev), torch.empty(B, device=dev), torch.empty(B, device=dev),
                        torch.empty(B, N, device=dev), torch.empty(B, N + 1, device=dev))
                outs[lvl] = L.LevelOut(*[t.data_ptr() for t in tens])
                ret.append(tens)
        flags = L.FLAG_WHITE_BKGD if white_bkgd else 0
        L.check(L.lib().mipnerf_train_step(ctx.handle, B, C.byref(rp), gt.data_ptr(), t_rand.data_ptr() if randomized else None,
                                           u_rand.data_ptr() if randomized else None, None if dz is None else dz.data_ptr(),
                                           flags, float(coarse_loss_mult),
                                           float(distloss_mult), int(bool(disable_multiscale_loss)), ws.data_ptr(), ws.numel(),
                                           grad.data_ptr(), int(accumulate), scalars.data_ptr(), outs, ops._stream()), "train_step")
        return scalars, ret

    def _check_native_step(self, rays):
        if self.precision != L.PREC_BF16:
            raise NotImplementedError("train_step_native is the bf16 path; fp32 parity mode trains through autograd")
        if not self.stop_resample_grad:
            raise NotImplementedError("stop_resample_grad=False trains through autograd in fp32 precision (the one-call native step "
                                      "implements the shipped stop-gradient resampler)")
        if not rays.origins.is_cuda:
            raise RuntimeError("train_step_native needs rays on a HIP device; there is no CPU fallback")

    def train_step_native(self, rays: Rays, gt_rgb, randomized: bool, white_bkgd: bool, coarse_loss_mult: float = 0.1,
                          distloss_mult: float = 0.01, disable_multiscale_loss: bool = False, t_rand=None, u_rand=None,
                          return_outputs: bool = False, density_randn=None):
        """forward + loss (nerf_system.py:99-111) + backward of the whole hot path in ONE native call
        (mipnerf_train_step): no autograd graph.  The gradient of the loss lands in the parameters' .grad (zero-copy
        when the MLP is in flat mode, `mlp.flatten_parameters()`).  Returns (scalars [6] tensor = loss, mse_c, mse_f,
        distloss_c, distloss_f, psnr_fine, outputs or None).  bf16 precision only."""
        self._check_native_step(rays)
        dev = rays.origins.device
        mlp = self.mlp
        ctx = mlp.native(dev)
        dropped = all(p.grad is None for p in mlp.ordered_params())
        flat_mode = mlp.grads_are_flat() or (mlp.is_flat() and dropped)
        if flat_mode:
            if dropped:                         # zero_grad(set_to_none=True) of a foreign optimizer: start from zero
                mlp._flat_grad_valid = False
            mlp.gather_foreign_grads()          # re-attaches the .grad views
            grad, accumulate = mlp._flat_grad, 1 if mlp._flat_grad_valid else 0
        else:
            total = ctx.grad_numel([p.shape for p in mlp.ordered_params()])
            grad, accumulate = torch.empty(total, device=dev, dtype=torch.float32), 0
        scalars, ret = self._train_step_call(rays, gt_rgb, randomized, white_bkgd, coarse_loss_mult, distloss_mult, disable_multiscale_loss,
                                             t_rand, u_rand, density_randn, grad, accumulate, return_outputs)
        if flat_mode:
            mlp._flat_grad_valid = True
        else:
            ps = mlp.ordered_params()
            for p, g in zip(ps, ctx.split_grads(grad, [p.shape for p in ps])):
                p.grad = g if p.grad is None else p.grad.add_(g)
        return scalars, ret

    def loss_native(self, rays: Rays, gt_rgb, randomized: bool, white_bkgd: bool, coarse_loss_mult: float = 0.1,
                    distloss_mult: float = 0.01, disable_multiscale_loss: bool = False, t_rand=None, u_rand=None, density_randn=None):
        """The training loss of nerf_system.py:99-111 as ONE autograd node over the parameters (round 6): its forward is the one-call
        native step (forward of both levels + loss + the whole backward: the 3-kernel bf16 path), its backward hands the gradient that
        call already computed -- scaled by the incoming dL/dloss -- to the parameters.  `loss.backward()`, AccumulateGrad, tensor hooks
        and therefore DistributedDataParallel's reducer and any torch optimizer see ordinary per-parameter gradients, so a loop that
        knows nothing about this package (Lightning's automatic optimisation, train.py:48-64) runs the fast path.
        Returns (loss [] attached to the graph, scalars [6] detached = loss, mse_c, mse_f, distloss_c, distloss_f, psnr_fine)."""
        self._check_native_step(rays)
        from .autograd import native_step_loss
        return native_step_loss(self, rays, gt_rgb, randomized, white_bkgd, coarse_loss_mult, distloss_mult, disable_multiscale_loss,
                                t_rand, u_rand, density_randn)

    def _forward_native(self, rays, randomized, white_bkgd, t_rand=None, u_rand=None, density_randn=None, out=None, ws=None):
        """`out` (optional): preallocated per-level tuples (comp_rgb, distance, acc, weights, t_samples) to write into
        (contiguous fp32 HIP tensors of the right shapes) instead of fresh tensors.  `ws` (optional): a private workspace
        (uint8, >= mipnerf_workspace_bytes) for a caller that keeps several forwards in flight on different streams."""
        dev = rays.origins.device
        B, N = rays.origins.shape[0], self.num_samples
        ctx = self.mlp.native(dev)
        f = [ops._f32c(getattr(rays, k), k) for k in Rays._fields]
        rp = L.RaysPtrs(*[t.data_ptr() for t in f])
        if randomized:
            # mip.py:159 / 201: the two uniform draws, from torch's device RNG
            t_rand = torch.rand(B, N + 1, device=dev) if t_rand is None else ops._f32c(t_rand, "t_rand")
            u_rand = torch.rand(B, N + 1, device=dev) if u_rand is None else ops._f32c(u_rand, "u_rand")
        outs = (L.LevelOut * self.num_levels)()
        ret = []
        for lvl in range(self.num_levels):
            if out is not None:
                comp_rgb, distance, acc, weights, t_samples = out[lvl]
                for t_, shp in ((comp_rgb, (B, 3)), (distance, (B,)), (acc, (B,)), (weights, (B, N)), (t_samples, (B, N + 1))):
                    if tuple(t_.shape) != shp or t_.dtype != torch.float32 or not t_.is_contiguous() or t_.device != dev:
                        raise ValueError(f"_forward_native: bad preallocated output {tuple(t_.shape)} (want {shp}, contiguous fp32 on {dev})")
            else:
                comp_rgb = torch.empty(B, 3, device=dev)
                distance = torch.empty(B, device=dev)
                acc = torch.empty(B, device=dev)
                weights = torch.empty(B, N, device=dev)
                t_samples = torch.empty(B, N + 1, device=dev)
            outs[lvl] = L.LevelOut(comp_rgb.data_ptr(), distance.data_ptr(), acc.data_ptr(), weights.data_ptr(),
                                   t_samples.data_ptr())
            ret.append((comp_rgb, distance, acc, weights, t_samples))
        if ws is None:
            ws = ctx.workspace(B)
        flags = L.FLAG_WHITE_BKGD if white_bkgd else 0
        dz = self._density_randn(randomized, B, dev, density_randn)     # mip_nerf.py:232-233
        L.check(L.lib().mipnerf_forward(ctx.handle, B, C.byref(rp), t_rand.data_ptr() if randomized else None,
                                        u_rand.data_ptr() if randomized else None, None if dz is None else dz.data_ptr(),
                                        flags, self.precision, ws.data_ptr(), ws.numel(), outs, ops._stream()), "mipnerf_forward")
        return ret


class GraphedForward:
    """ONE batch's `MipNerf.forward` (inference: randomized=False) replayed from a captured hipGraph (round 5): the ~6 launches of a
    forward (ray prologue, coarse MLP, compositing + resampling, fine MLP, compositing) over static input / output / workspace
    buffers, so a step is one graph launch and no Python between the kernels.
        gf = GraphedForward(model, B, white_bkgd=True)
        ret = gf(rays)          # copies the 7 ray tensors into gf.static_in, replays; returns the static per-level tuples
        ret = gf.replay()       # rays already in gf.static_in (e.g. written there by the device-side ray generator)
    The returned tensors are the graph's own buffers: valid until the next call.  Same bits as the eager forward (same launches);
    a parameter change is re-packed outside the graph before the replay, a `set_precision` re-captures."""

    def __init__(self, model: "MipNerf", num_rays: int, white_bkgd: bool, device: Optional[torch.device] = None):
        device = torch.device(device) if device is not None else next(model.parameters()).device
        if num_rays < 1:
            raise ValueError("GraphedForward needs at least one ray")
        self.model, self.n, self.white_bkgd, self.dev = model, int(num_rays), bool(white_bkgd), device
        N = model.num_samples
        self.static_in = Rays(*[torch.zeros(self.n, k, device=device) for k in (3, 3, 3, 1, 1, 1, 1)])
        for k in ("directions", "viewdirs"):
            getattr(self.static_in, k)[:, 2] = 1.0
        self.static_in.radii.fill_(1e-3)
        self.static_in.near.fill_(2.0)
        self.static_in.far.fill_(6.0)
        self.out = [(torch.zeros(self.n, 3, device=device), torch.zeros(self.n, device=device), torch.zeros(self.n, device=device),
                     torch.zeros(self.n, N, device=device), torch.zeros(self.n, N + 1, device=device)) for _ in range(model.num_levels)]
        self.graph = None
        self.capture_error = None
        self._ws = None

    def _launch(self):
        self.model._forward_native(self.static_in, False, self.white_bkgd, out=self.out, ws=self._ws)

    def _capture(self):
        ctx = self.model.mlp.native(self.dev)
        need = int(L.lib().mipnerf_workspace_bytes(ctx.handle, self.n))
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.dev)
        s = torch.cuda.Stream(device=self.dev)
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s), torch.no_grad():     # warm-up on a side stream: lazy init, packing
            self._launch()
        torch.cuda.current_stream().wait_stream(s)
        graph = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(graph, capture_error_mode="thread_local"), torch.no_grad():
                self._launch()
            self.graph = graph
        except RuntimeError as e:
            import sys
            print(f"[mipnerf_pl_amd] hipGraph capture of the forward failed ({e}); launching eagerly", file=sys.stderr)
            torch.cuda.synchronize()
            self.graph, self.capture_error = False, str(e)

    def replay(self):
        with torch.cuda.device(self.dev):
            self.model.mlp.native(self.dev)         # re-pack OUTSIDE the graph when a parameter changed
            if self.graph is not None and self._captured_precision != self.model.precision:
                self.graph = None                   # MipNerf.set_precision since the capture
            if self.graph is None:
                self._captured_precision = self.model.precision
                self._capture()
            if self.graph is False:
                with torch.no_grad():
                    self._launch()
            else:
                self.graph.replay()
        return self.out

    def __call__(self, rays: Rays):
        if rays.origins.shape[0] != self.n:
            raise ValueError(f"GraphedForward captured for {self.n} rays, got {rays.origins.shape[0]}")
        for dst, src in zip(self.static_in, rays):
            dst.copy_(src)
        return self.replay()


class GraphedFrame:
    """Whole-frame rendering from ONE captured hipGraph (BASELINE configs[4]: 800 x 800 = 640,000 rays in 8192-ray chunks =
    79 chunk forwards, ~20 kernels each): every chunk launch reads its slice of a static full-frame ray buffer and writes its
    slice of static full-frame outputs, so a frame is 7 ray copies + one graph launch -- no per-chunk copies, clones or
    Python.  `__call__(rays)` takes flattened [n, k] rays and returns (coarse_rgb [n,3], fine_rgb [n,3], distance [n])
    views of the static outputs (valid until the next call).
    `lanes` (default 2, env MIPNERF_FRAME_LANES): the chunks are independent, so the capture forks into that many streams that
    take the chunks round-robin, each with its own workspace and per-sample scratch: the small kernels and the kernel boundaries
    of one lane run while the other lane's MLP kernel owns the CUs."""

    def __init__(self, model: "MipNerf", num_rays: int, chunk: int, white_bkgd: bool, device: torch.device, lanes: Optional[int] = None):
        self.model, self.n, self.chunk, self.white_bkgd, self.dev = model, int(num_rays), int(chunk), bool(white_bkgd), device
        self.lanes = max(1, int(lanes if lanes is not None else os.environ.get("MIPNERF_FRAME_LANES", "2")))
        self.static_in = Rays(*[torch.zeros(self.n, k, device=device) for k in (3, 3, 3, 1, 1, 1, 1)])
        for k in ("directions", "viewdirs"):
            getattr(self.static_in, k)[:, 2] = 1.0
        self.static_in.radii.fill_(1e-3)
        self.static_in.near.fill_(2.0)
        self.static_in.far.fill_(6.0)
        L_ = model.num_levels
        self.rgb = [torch.zeros(self.n, 3, device=device) for _ in range(L_)]
        self.dist = [torch.zeros(self.n, device=device) for _ in range(L_)]
        self.acc = [torch.zeros(self.n, device=device) for _ in range(L_)]
        self.graph = None

    def _run_chunks(self):
        m, N = self.model, self.model.num_samples
        bounds = [(lo, min(self.n, lo + self.chunk)) for lo in range(0, self.n, self.chunk)]
        lanes = min(self.lanes, len(bounds))
        ctx = m.mlp.native(self.dev)
        need = int(L.lib().mipnerf_workspace_bytes(ctx.handle, min(self.n, self.chunk)))
        if getattr(self, "_lane_ws", None) is None or len(self._lane_ws) != lanes or self._lane_ws[0].numel() < need:
            self._lane_ws = [torch.empty(need, dtype=torch.uint8, device=self.dev) for _ in range(lanes)]
            self._lane_streams = [torch.cuda.Stream(device=self.dev) for _ in range(lanes)] if lanes > 1 else []
            self._scratch = [dict() for _ in range(lanes)]
        cur = torch.cuda.current_stream()
        for s_ in self._lane_streams:            # fork (inside a capture the lanes join the captured graph)
            s_.wait_stream(cur)
        for ci, (lo, hi) in enumerate(bounds):
            lane = ci % lanes
            b = hi - lo
            scratch = self._scratch[lane]
            if b not in scratch:      # per-sample outputs nobody reads after the chunk: shared by all chunks of that size on this lane
                scratch[b] = [(torch.empty(b, N, device=self.dev), torch.empty(b, N + 1, device=self.dev)) for _ in range(m.num_levels)]
            out = [(self.rgb[l][lo:hi], self.dist[l][lo:hi], self.acc[l][lo:hi], scratch[b][l][0], scratch[b][l][1])
                   for l in range(m.num_levels)]
            rays = Rays(*[x[lo:hi] for x in self.static_in])
            if lanes > 1:
                with torch.cuda.stream(self._lane_streams[lane]):
                    m._forward_native(rays, False, self.white_bkgd, out=out, ws=self._lane_ws[lane])
            else:
                m._forward_native(rays, False, self.white_bkgd, out=out, ws=self._lane_ws[0])
        for s_ in self._lane_streams:            # join
            cur.wait_stream(s_)

    def _capture(self):
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s), torch.no_grad():     # warm-up on a side stream: lazy init, workspace growth, packing
            self.model._forward_native(Rays(*[x[:min(self.n, self.chunk)] for x in self.static_in]), False, self.white_bkgd)
        torch.cuda.current_stream().wait_stream(s)
        graph = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(graph, capture_error_mode="thread_local"), torch.no_grad():
                self._run_chunks()
            self.graph = graph
        except RuntimeError as e:
            import sys
            print(f"[mipnerf_pl_amd] hipGraph capture of the frame failed ({e}); rendering eagerly", file=sys.stderr)
            torch.cuda.synchronize()
            self.graph = False

    def __call__(self, rays: Rays):
        if rays.origins.shape[0] != self.n:
            raise ValueError(f"GraphedFrame captured for {self.n} rays, got {rays.origins.shape[0]}")
        self.model.mlp.native(self.dev)         # re-pack OUTSIDE the graph when a parameter changed
        if self.graph is not None and getattr(self, "_captured_precision", None) != self.model.precision:
            self.graph = None                   # MipNerf.set_precision since the capture: the graph holds the other precision's launches
        if self.graph is None:
            self._captured_precision = self.model.precision
            self._capture()
        for dst, src in zip(self.static_in, rays):
            dst.copy_(src)
        if self.graph is False:
            with torch.no_grad():
                self._run_chunks()
        else:
            self.graph.replay()
        return self.rgb[0], self.rgb[-1], self.dist[-1]
