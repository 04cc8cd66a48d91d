"""Host-side mirror of the free functions of the reference's models/mip.py, same names and
argument meaning, executing on MI355X through the C ABI of libmipnerf_hip.so.

Every function takes / returns torch tensors that live on a HIP device (`tensor.is_cuda`);
torch is only the owner of device memory and of the current stream.  There is no CPU path:
CPU tensors raise.  Randomised variants draw their uniform noise with torch's device RNG
(`torch.rand`) and hand it to the kernels, which apply it exactly like mip.py:155-160 / 198-204.
"""
from __future__ import annotations

import torch

from . import _lib as L


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _f32c(t: torch.Tensor, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError(f"{name}: the MI355X-native path needs HIP device tensors (got {t.device}); "
                           "there is no CPU fallback")
    if t.dtype != torch.float32:
        raise TypeError(f"{name}: expected float32, got {t.dtype}")
    return t.contiguous()


def _ptr(t):
    return None if t is None else t.data_ptr()


def _torch_dtype(precision: int):
    return torch.bfloat16 if precision == L.PREC_BF16 else torch.float32


# ---------------------------------------------------------------------------------------------
def cast_rays(t_samples, origins, directions, radii, ray_shape="cone", diagonal=True):
    """models/mip.py:81-103 -> (means [B,N,3], covs [B,N,3])."""
    if ray_shape != "cone" or not diagonal:
        raise NotImplementedError  # mip.py:97-98 ('cylinder'); full covariances are dead code upstream
    t_samples = _f32c(t_samples, "t_samples")
    B, N1 = t_samples.shape
    N = N1 - 1
    means = torch.empty(B, N, 3, device=t_samples.device, dtype=torch.float32)
    covs = torch.empty_like(means)
    # contiguous copies (if any) are bound to locals: a temporary would be freed, and its block re-used by the next
    # temporary, before the launch
    o, d, r = _f32c(origins, "origins"), _f32c(directions, "directions"), _f32c(radii, "radii")
    L.check(L.lib().mipnerf_cast_rays(B, N, _ptr(t_samples), _ptr(o), _ptr(d), _ptr(r), _ptr(means), _ptr(covs), _stream()),
            "cast_rays")
    return means, covs


def sample_t(num_samples, near, far, randomized, disparity, t_rand=None):
    """t part of sample_along_rays (mip.py:143-163): [B, N+1]."""
    near = _f32c(near, "near")
    far = _f32c(far, "far")
    B = near.shape[0]
    if randomized and t_rand is None:
        t_rand = torch.rand(B, num_samples + 1, device=near.device)   # mip.py:159
    t = torch.empty(B, num_samples + 1, device=near.device, dtype=torch.float32)
    tr = _f32c(t_rand, "t_rand") if randomized else None
    L.check(L.lib().mipnerf_sample_along_rays(B, num_samples, _ptr(near), _ptr(far), _ptr(tr),
                                              int(bool(disparity)), _ptr(t), _stream()), "sample_along_rays")
    return t


def sample_along_rays(origins, directions, radii, num_samples, near, far, randomized, disparity, ray_shape,
                      t_rand=None):
    """models/mip.py:127-165 -> (t_samples [B,N+1], (means, covs))."""
    t = sample_t(num_samples, near, far, randomized, disparity, t_rand)
    return t, cast_rays(t, origins, directions, radii, ray_shape)


def sorted_piecewise_constant_pdf(bins, weights, num_samples, randomized, u_rand=None):
    """models/mip.py:168-229.  `weights` is NOT mutated (the reference pads it in place)."""
    bins = _f32c(bins, "bins")
    weights = _f32c(weights, "weights")
    B, nb = weights.shape
    if randomized and u_rand is None:
        u_rand = torch.rand(B, num_samples, device=bins.device)       # stands in for uniform_(mip.py:201)
    out = torch.empty(B, num_samples, device=bins.device, dtype=torch.float32)
    ur = _f32c(u_rand, "u_rand") if randomized else None
    L.check(L.lib().mipnerf_sorted_piecewise_constant_pdf(
        B, nb, _ptr(bins), _ptr(weights), num_samples, _ptr(ur), _ptr(out), _stream()),
        "sorted_piecewise_constant_pdf")
    return out


def resample_t(t_samples, weights, randomized, resample_padding, u_rand=None):
    """t part of resample_along_rays (mip.py:250-271): blur-pool + padding + PDF inversion."""
    t_samples = _f32c(t_samples, "t_samples")
    weights = _f32c(weights.detach(), "weights")     # no graph here; the differentiable form is autograd._ResampleT
    B, N = weights.shape
    if randomized and u_rand is None:
        u_rand = torch.rand(B, N + 1, device=weights.device)
    out = torch.empty(B, N + 1, device=weights.device, dtype=torch.float32)
    ur = _f32c(u_rand, "u_rand") if randomized else None
    L.check(L.lib().mipnerf_resample_along_rays(
        B, N, _ptr(t_samples), _ptr(weights), _ptr(ur), float(resample_padding), _ptr(out), _stream()), "resample_along_rays")
    return out


def resample_along_rays(origins, directions, radii, t_samples, weights, randomized, ray_shape, stop_grad,
                        resample_padding, u_rand=None):
    """models/mip.py:232-280 -> (new_t_vals [B,N+1], (means, covs))."""
    if not stop_grad and torch.is_grad_enabled() and weights.requires_grad:
        # mip.py:265-279: new_t_vals stays differentiable w.r.t. the weights (native backward, autograd._ResampleT); the Gaussians
        # below are computed without a graph -- inside MipNerf the fused encoding (autograd._CastIPE) carries that gradient
        from .autograd import _ResampleT
        if randomized and u_rand is None:
            u_rand = torch.rand(weights.shape[0], weights.shape[1] + 1, device=weights.device)
        t = _ResampleT.apply(t_samples, weights, resample_padding, u_rand if randomized else None)
        # the reference keeps means / covs differentiable w.r.t. t here (mip.py:265-280).  The native backward of that link is
        # fused with the encoding (autograd._CastIPE, what MipNerf uses); the stand-alone Gaussians stay attached to the graph
        # through a node whose backward RAISES, so composing this op with integrated_pos_enc can never silently drop the gradient
        from .autograd import _CastRaysNoBackward
        means, covs = _CastRaysNoBackward.apply(t, origins, directions, radii, ray_shape)
        return t, (means, covs)
    t = resample_t(t_samples, weights, randomized, resample_padding, u_rand)
    return t, cast_rays(t, origins, directions, radii, ray_shape)


def integrated_pos_enc(means_covs, min_deg, max_deg, diagonal=True, precision=L.PREC_FP32):
    """models/mip.py:322-350 -> [B, N, 6*(max_deg-min_deg)] (float32, or bfloat16 for the bf16 MLP)."""
    if not diagonal:
        raise NotImplementedError
    means, covs = means_covs
    means = _f32c(means, "means")
    covs = _f32c(covs, "covs")
    M = means.numel() // 3
    enc = torch.empty(*means.shape[:-1], 6 * (max_deg - min_deg), device=means.device, dtype=_torch_dtype(precision))
    L.check(L.lib().mipnerf_integrated_pos_enc(M, min_deg, max_deg, _ptr(means), _ptr(covs), _ptr(enc),
                                               precision, _stream()), "integrated_pos_enc")
    return enc


def cast_ipe(t_samples, origins, directions, radii, min_deg, max_deg, disable_integration=False,
             precision=L.PREC_FP32):
    """cast_rays + integrated_pos_enc fused (what MipNerf.forward uses): [B, N, 6L]."""
    t_samples = _f32c(t_samples, "t_samples")
    B, N1 = t_samples.shape
    enc = torch.empty(B, N1 - 1, 6 * (max_deg - min_deg), device=t_samples.device, dtype=_torch_dtype(precision))
    o, d, r = _f32c(origins, "origins"), _f32c(directions, "directions"), _f32c(radii, "radii")   # kept alive until after the launch
    L.check(L.lib().mipnerf_cast_ipe(B, N1 - 1, min_deg, max_deg, int(bool(disable_integration)), _ptr(t_samples),
                                     _ptr(o), _ptr(d), _ptr(r), _ptr(enc), precision, _stream()), "cast_ipe")
    return enc


def pos_enc(x, min_deg, max_deg, append_identity=True, precision=L.PREC_FP32, ld=None):
    """models/mip.py:353-363 for min_deg == 0, append_identity=True -> [B, 3 + 6*max_deg] (row stride ld)."""
    if min_deg != 0 or not append_identity:
        raise NotImplementedError("pos_enc is implemented for min_deg=0, append_identity=True (the only call "
                                  "on the hot path, mip_nerf.py:220-225)")
    x = _f32c(x, "x")
    B = x.shape[0]
    width = 3 + 6 * max_deg
    ld = width if ld is None else ld
    out = torch.empty(B, ld, device=x.device, dtype=_torch_dtype(precision))
    L.check(L.lib().mipnerf_pos_enc(B, max_deg, _ptr(x), _ptr(out), ld, precision, _stream()), "pos_enc")
    return out


def volumetric_rendering(rgb, density, t_samples, dirs, white_bkgd):
    """models/mip.py:366-401 -> (comp_rgb [B,3], distance [B], acc [B], weights [B,N])."""
    rgb_sigma = torch.cat([_f32c(rgb, "rgb"), _f32c(density, "density")], dim=-1).contiguous()
    return volumetric_rendering_packed(rgb_sigma, t_samples, dirs, white_bkgd)


def volumetric_rendering_packed(rgb_sigma, t_samples, dirs, white_bkgd):
    """Same, on the MLP kernel's packed output [B, N, 4] = (r, g, b, sigma)."""
    rgb_sigma = _f32c(rgb_sigma, "rgb_sigma")
    t_samples = _f32c(t_samples, "t_samples")
    B, N1 = t_samples.shape
    N = N1 - 1
    dev = t_samples.device
    comp_rgb = torch.empty(B, 3, device=dev)
    distance = torch.empty(B, device=dev)
    acc = torch.empty(B, device=dev)
    weights = torch.empty(B, N, device=dev)
    dirs = _f32c(dirs, "dirs")
    L.check(L.lib().mipnerf_volumetric_rendering(B, N, _ptr(rgb_sigma), _ptr(t_samples), _ptr(dirs),
                                                 int(bool(white_bkgd)), _ptr(comp_rgb), _ptr(distance), _ptr(acc),
                                                 _ptr(weights), _stream()), "volumetric_rendering")
    return comp_rgb, distance, acc, weights


def camera_record(c2w, width, height, near, far, focal=None, pix2cam=None, lossmult=1.0, dtype=torch.float32):
    """One row of the camera table of `generate_rays` (32 numbers, include/mipnerf_hip.h): Blender pinhole camera
    when `focal` is given (datasets.py:226-228), Multicam when `pix2cam` is given (datasets.py:125-131).
    dtype=torch.float64: a table for float64 ray arithmetic (RenderGen, render_video.py:29-112)."""
    rec = torch.zeros(32, dtype=dtype)
    rec[0:12] = torch.as_tensor(c2w, dtype=dtype)[:3, :4].reshape(-1)
    if (focal is None) == (pix2cam is None):
        raise ValueError("give exactly one of focal (Blender) / pix2cam (Multicam)")
    if pix2cam is not None:
        rec[12:21] = torch.as_tensor(pix2cam, dtype=dtype)[:3, :3].reshape(-1)
        rec[26] = 1.0
    else:
        rec[27] = float(focal)
    rec[21], rec[22], rec[23], rec[24], rec[25] = float(width), float(height), float(near), float(far), float(lossmult)
    return rec


def generate_rays(cameras, num_rays=None, cam_idx=None, pix_idx=None):
    """Rays of datasets.py:214-263 / 116-168 computed on the device: `cameras` [ncam, 32] float32 HIP tensor of
    `camera_record` rows, ray i = pixel pix_idx[i] (y*W + x; None = i) of camera cam_idx[i] (None = 0).
    A float64 table selects float64 arithmetic (mipnerf_generate_rays_f64); the rays are float32 either way."""
    from .rays import Rays
    f64 = cameras.dtype == torch.float64
    if f64:
        if not cameras.is_cuda:
            raise RuntimeError("cameras: needs a HIP device tensor")
        cameras = cameras.contiguous().reshape(-1, 32)
    else:
        cameras = _f32c(cameras, "cameras").reshape(-1, 32)
    dev = cameras.device
    if num_rays is None:
        num_rays = int(pix_idx.numel()) if pix_idx is not None else int(cameras[0, 21].item() * cameras[0, 22].item())

    def i32(t, name):
        if t is None:
            return None
        if not t.is_cuda:
            raise RuntimeError(f"{name}: needs a HIP device tensor")
        return t.to(torch.int32).contiguous()
    ci, pi = i32(cam_idx, "cam_idx"), i32(pix_idx, "pix_idx")
    out = [torch.empty(num_rays, k, device=dev, dtype=torch.float32) for k in (3, 3, 3, 1, 1, 1, 1)]
    rp = L.RaysPtrs(*[t.data_ptr() for t in out])
    import ctypes as C
    fn = L.lib().mipnerf_generate_rays_f64 if f64 else L.lib().mipnerf_generate_rays
    L.check(fn(num_rays, _ptr(cameras), _ptr(ci), _ptr(pi), C.byref(rp), _stream()), "generate_rays")
    return Rays(*out)


def eval_errors(pred_color, batch_pixels):
    """utils/metrics.py:191-197: (psnr, ssim) of one rendered frame, pred/gt [1,H,W,3] (or [H,W,3]) fp32 HIP tensors;
    one fused kernel instead of six conv2d passes.  Returns two 0-d tensors."""
    a = _f32c(pred_color, "pred_color")
    b = _f32c(batch_pixels, "batch_pixels")
    if a.dim() == 4:
        if a.shape[0] != 1:
            raise NotImplementedError("eval_errors: one frame at a time (eval.py evaluates image by image)")
        a, b = a[0], b[0]
    if a.shape != b.shape or a.dim() != 3 or a.shape[-1] != 3:
        raise ValueError("eval_errors: expected matching [H,W,3] images")
    H, W = int(a.shape[0]), int(a.shape[1])
    ws = torch.empty(int(L.lib().mipnerf_eval_workspace_floats(H, W)), device=a.device, dtype=torch.float32)
    out = torch.empty(2, device=a.device, dtype=torch.float32)
    a, b = a.contiguous(), b.contiguous()
    L.check(L.lib().mipnerf_eval_errors(H, W, _ptr(a), _ptr(b), _ptr(ws), _ptr(out), _stream()),
            "eval_errors")
    return out[0], out[1]


def selftest() -> str:
    """Run the hardware self-test (MFMA lane layouts, LDS DMA); returns the report, raises on failure."""
    rc = L.lib().mipnerf_selftest(_stream())
    msg = L.last_error()
    if rc != 0:
        raise RuntimeError(f"gfx950 self-test failed (code {rc:#x}): {msg}")
    return msg


# ---- unbounded scenes (mip-NeRF 360) -------------------------------------------------------------------------------------
# Working versions of the reference's dead functions (models/mip.py:106-124, 292-319, 424-447); they follow the paper the
# dead code aims at (Barron et al., "Mip-NeRF 360", CVPR 2022) -- csrc/raymath360.hpp explains what is wrong upstream.
def sample_t_360(num_samples, near, far, randomized, t_rand=None):
    """Fence posts uniform in normalised inverse depth: (t_inv [B,N+1], t [B,N+1]) (mip.py:106-121)."""
    near, far = _f32c(near, "near"), _f32c(far, "far")
    B = near.shape[0]
    if randomized and t_rand is None:
        t_rand = torch.rand(B, num_samples + 1, device=near.device)
    tr = _f32c(t_rand, "t_rand") if randomized else None
    t_inv = torch.empty(B, num_samples + 1, device=near.device, dtype=torch.float32)
    t = torch.empty_like(t_inv)
    L.check(L.lib().mipnerf_sample_along_rays_360(B, num_samples, _ptr(near), _ptr(far), _ptr(tr), _ptr(t_inv), _ptr(t),
                                                  _stream()), "sample_along_rays_360")
    return t_inv, t


def cast_rays_360(t_samples, origins, directions, radii, contracted=False):
    """Conical frustums -> Gaussians with FULL covariance, optionally contracted: (means [B,N,3], covs [B,N,3,3])."""
    t_samples = _f32c(t_samples, "t_samples")
    B, N1 = t_samples.shape
    means = torch.empty(B, N1 - 1, 3, device=t_samples.device, dtype=torch.float32)
    covs = torch.empty(B, N1 - 1, 3, 3, device=t_samples.device, dtype=torch.float32)
    o, d, r = _f32c(origins, "origins"), _f32c(directions, "directions"), _f32c(radii, "radii")
    L.check(L.lib().mipnerf_cast_ipe_360(B, N1 - 1, 0, 1, int(bool(contracted)), _ptr(t_samples), _ptr(o), _ptr(d), _ptr(r), None,
                                         L.PREC_FP32, _ptr(means), _ptr(covs), _stream()), "cast_rays_360")
    return means, covs


def sample_along_rays_360(origins, directions, radii, num_samples, near, far, randomized, disparity=False, ray_shape="cone",
                          t_rand=None):
    """models/mip.py:106-124 (same signature and return): (t_inv [B,N+1], (means [B,N,3], covs [B,N,3,3]))."""
    if ray_shape != "cone":
        raise NotImplementedError
    t_inv, t = sample_t_360(num_samples, near, far, randomized, t_rand)
    return t_inv, cast_rays_360(t, origins, directions, radii, contracted=False)


def cast_ipe_360(t_samples, origins, directions, radii, min_deg, max_deg, contracted=True, precision=L.PREC_FP32, fragments=False):
    """Fused frustum -> full-covariance Gaussian -> contraction -> off-axis IPE: [B, N, 2*21*(max_deg-min_deg)]; what
    `integrated_pos_enc_360(parameterization(cast_rays(...)))` of the reference is meant to compute (mip.py:292-319, 431-447).
    fragments=True (bf16 only): the MFMA B-operand fragment layout the two-kernel MLP form reads fastest -- an opaque
    [ceil(B N / 256) * 256, features] bf16 buffer for `autograd.mlp_native(..., frag_shape=(B, N))`, not a row-major tensor."""
    t_samples = _f32c(t_samples, "t_samples")
    B, N1 = t_samples.shape
    F = 42 * (max_deg - min_deg)
    o, d, r = _f32c(origins, "origins"), _f32c(directions, "directions"), _f32c(radii, "radii")
    if fragments:
        if precision != L.PREC_BF16:
            raise ValueError("cast_ipe_360(fragments=True) is a bf16 layout")
        M = B * (N1 - 1)
        enc = torch.empty((M + 255) // 256 * 256, F, device=t_samples.device, dtype=torch.bfloat16)
        L.check(L.lib().mipnerf_cast_ipe_360(B, N1 - 1, min_deg, max_deg, int(bool(contracted)), _ptr(t_samples), _ptr(o), _ptr(d),
                                             _ptr(r), _ptr(enc), L.OUT_BF16_FRAGMENTS, None, None, _stream()), "cast_ipe_360")
        return enc
    enc = torch.empty(B, N1 - 1, F, device=t_samples.device, dtype=_torch_dtype(precision))
    L.check(L.lib().mipnerf_cast_ipe_360(B, N1 - 1, min_deg, max_deg, int(bool(contracted)), _ptr(t_samples), _ptr(o), _ptr(d),
                                         _ptr(r), _ptr(enc), precision, None, None, _stream()), "cast_ipe_360")
    return enc


def parameterization(means, covs):
    """models/mip.py:431-447 done right: the scene contraction pushed through Gaussians, (contract(mean), J cov J^T)."""
    means, covs = _f32c(means, "means"), _f32c(covs, "covs")
    M = means.numel() // 3
    if covs.numel() != 9 * M:
        raise ValueError("parameterization: covs must be [..., 3, 3] full covariances")
    mo, co = torch.empty_like(means), torch.empty_like(covs)
    L.check(L.lib().mipnerf_gauss_360(M, 0, 1, 1, _ptr(means), _ptr(covs), None, L.PREC_FP32, _ptr(mo), _ptr(co), _stream()),
            "parameterization")
    return mo, co


def contract(x):
    """models/mip.py:424-426: x inside the unit ball, (2 - 1/|x|) x/|x| outside (mip-NeRF 360 eq. 10); x [..., 3]."""
    x = _f32c(x, "x")
    M = x.numel() // 3
    out = torch.empty_like(x)
    zero = torch.zeros(M, 9, device=x.device, dtype=torch.float32)
    L.check(L.lib().mipnerf_gauss_360(M, 0, 1, 1, _ptr(x), _ptr(zero), None, L.PREC_FP32, _ptr(out), None, _stream()), "contract")
    return out


def integrated_pos_enc_360(means_covs, min_deg=0, max_deg=1, contracted=True, precision=L.PREC_FP32):
    """models/mip.py:292-319 done right: off-axis integrated positional encoding of (means [...,3], covs [...,3,3]) on the 21
    icosahedron directions at the frequencies 2^min_deg .. 2^(max_deg-1) (the dead upstream code has a single frequency =
    the defaults here), contracting the Gaussians first like its `parameterization` call: [..., 2*21*(max_deg-min_deg)]."""
    means, covs = means_covs
    means, covs = _f32c(means, "means"), _f32c(covs, "covs")
    M = means.numel() // 3
    enc = torch.empty(*means.shape[:-1], 42 * (max_deg - min_deg), device=means.device, dtype=_torch_dtype(precision))
    L.check(L.lib().mipnerf_gauss_360(M, min_deg, max_deg, int(bool(contracted)), _ptr(means), _ptr(covs), _ptr(enc), precision,
                                      None, None, _stream()), "integrated_pos_enc_360")
    return enc
