"""Training path of MipNerf.forward (gradients w.r.t. the 24 MLP parameter tensors).

Native (HIP) in both directions: sampling, conical-frustum + IPE, view encoding, the MLP (bf16 mode:
forward-with-save, dgrad and wgrad MFMA kernels, see mlp_train_plan.py), activations, volumetric rendering and
its backward (fused with the activation derivatives), resampling, distloss.  torch.autograd only chains the
native pieces (custom Functions) and owns the buffers.
fp32 ("parity") mode: fused exact-fp32 MFMA forward that saves the layer outputs + GEMM-based dgrad / wgrad on the
fp32 matrix instruction (kernels_gemm_f32.hip) -- the instrument of the gradient-parity tests against the reference.
"""
from __future__ import annotations

import torch

from . import _lib as L
from . import ops


class _RenderFromRaw(torch.autograd.Function):
    """activations (mip_nerf.py:236-238) + volumetric_rendering (mip.py:366-401) on raw [B,N,4]."""

    @staticmethod
    def forward(ctx, raw, t_samples, dirs, white_bkgd, rgb_padding, density_bias, density_randn=None, density_noise=0.0):
        raw = ops._f32c(raw, "raw")
        B, N = raw.shape[0], raw.shape[1]
        rgb_sigma = torch.empty_like(raw)
        dz = None if density_randn is None else ops._f32c(density_randn, "density_randn")     # mip_nerf.py:232-233
        L.check(L.lib().mipnerf_activate(B * N, raw.data_ptr(), float(rgb_padding), float(density_bias),
                                         None if dz is None else dz.data_ptr(), float(density_noise),
                                         rgb_sigma.data_ptr(), ops._stream()), "activate")
        comp_rgb, distance, acc, weights = ops.volumetric_rendering_packed(rgb_sigma, t_samples, dirs, white_bkgd)
        ctx.save_for_backward(rgb_sigma, ops._f32c(t_samples, "t"), ops._f32c(dirs, "dirs"))
        ctx.white = bool(white_bkgd)
        ctx.rgb_padding = float(rgb_padding)
        return comp_rgb, distance, acc, weights

    @staticmethod
    def backward(ctx, g_rgb, g_dist, g_acc, g_w):
        rgb_sigma, t, dirs = ctx.saved_tensors
        B, N = rgb_sigma.shape[0], rgb_sigma.shape[1]
        d_raw = torch.empty_like(rgb_sigma)

        def p(g):
            return None if g is None else g.contiguous().float().data_ptr()
        keep = [g.contiguous().float() if g is not None else None for g in (g_rgb, g_dist, g_acc, g_w)]
        if ctx.needs_input_grad[1]:        # t_samples in the graph (stop_resample_grad=False): also dL/dt_samples
            d_t = torch.empty_like(t)
            L.check(L.lib().mipnerf_volumetric_rendering_bwd_t(
                B, N, rgb_sigma.data_ptr(), t.data_ptr(), dirs.data_ptr(), int(ctx.white),
                *[None if k is None else k.data_ptr() for k in keep], ctx.rgb_padding, d_raw.data_ptr(), d_t.data_ptr(),
                ops._stream()), "volumetric_rendering_bwd_t")
            return d_raw, d_t, None, None, None, None, None, None
        L.check(L.lib().mipnerf_volumetric_rendering_bwd(
            B, N, rgb_sigma.data_ptr(), t.data_ptr(), dirs.data_ptr(), int(ctx.white),
            *[None if k is None else k.data_ptr() for k in keep], ctx.rgb_padding, d_raw.data_ptr(), ops._stream()),
            "volumetric_rendering_bwd")
        return d_raw, None, None, None, None, None, None, None


class _DistLossRays(torch.autograd.Function):
    """Per-ray distortion loss (mip.py:8-20 before the batch mean), O(N) per ray, no [B,N,N] tensors."""

    @staticmethod
    def forward(ctx, weights, t_samples):
        weights = ops._f32c(weights, "weights")
        t_samples = ops._f32c(t_samples, "t_samples")
        B, N = weights.shape
        out = torch.empty(B, device=weights.device)
        L.check(L.lib().mipnerf_distloss(B, N, weights.data_ptr(), t_samples.data_ptr(), out.data_ptr(), None, None,
                                         ops._stream()), "distloss")
        ctx.save_for_backward(weights, t_samples)
        return out

    @staticmethod
    def backward(ctx, g_ray):
        weights, t_samples = ctx.saved_tensors
        B, N = weights.shape
        g = g_ray.contiguous().float()
        d_w = torch.empty_like(weights)
        if ctx.needs_input_grad[1]:        # t_samples in the graph (stop_resample_grad=False)
            d_t = torch.empty_like(t_samples)
            L.check(L.lib().mipnerf_distloss_bwd(B, N, weights.data_ptr(), t_samples.data_ptr(), g.data_ptr(), d_w.data_ptr(),
                                                 d_t.data_ptr(), ops._stream()), "distloss_bwd_t")
            return d_w, d_t
        L.check(L.lib().mipnerf_distloss(B, N, weights.data_ptr(), t_samples.data_ptr(), None, g.data_ptr(),
                                         d_w.data_ptr(), ops._stream()), "distloss_bwd")
        return d_w, None


class _MLPNative(torch.autograd.Function):
    """MLP.forward (models/mip_nerf.py:75-111) on the bf16 MFMA kernels, differentiable w.r.t. the parameters.
    forward: k_mlp_bf16_trainfwd (saves transposed activations + ReLU bit masks); backward: k_mlp_bf16_dgrad ->
    k_mlp_wgrad -> k_wgrad_reduce, one flat fp32 gradient buffer sliced into the 24 parameter gradients."""

    @staticmethod
    def forward(ctx, mlp, enc, venc, frag_shape, *params):
        dev = enc.device
        nctx = mlp.native(dev)                    # re-packs the weight streams if a parameter changed
        # frag_shape = (B, N): enc is the fragment buffer of ops.cast_ipe_360(fragments=True) (two-kernel variants), else [B, N, xyz_dim]
        B, N = frag_shape if frag_shape is not None else (enc.shape[0], enc.shape[1])
        M = B * N
        if enc.dtype != torch.bfloat16 or venc.dtype != torch.bfloat16 or venc.shape[-1] != 32:
            raise TypeError("native MLP training path: enc [B,N,xyz_dim] and viewenc [B,32] must be bfloat16")
        enc = enc.contiguous()
        venc = venc.contiguous()
        sz = nctx.train_sizes(M)
        act = torch.empty(sz[0], dtype=torch.uint8, device=dev)
        masks = torch.empty(sz[1], dtype=torch.uint8, device=dev)
        raw = torch.empty(B, N, 4, device=dev, dtype=torch.float32)
        rgb_sigma = torch.empty_like(raw)
        # fragment encodings go through their own entry point (ABI 6; a per-context toggle was not safe across threads / streams)
        fwd = L.lib().mipnerf_mlp_forward_train_fragments if frag_shape is not None else L.lib().mipnerf_mlp_forward_train
        L.check(fwd(nctx.handle, M, N, enc.data_ptr(), venc.data_ptr(), rgb_sigma.data_ptr(), raw.data_ptr(), act.data_ptr(), masks.data_ptr(),
                    ops._stream()), "mlp_forward_train")
        ctx.save_for_backward(act, masks)
        # wide encodings (the unbounded-scene model's two-kernel form): the weight-gradient kernel reads the ENCODING itself (rows or fragments) --
        # the act buffer records its address -- so it must outlive the forward; the standard shapes transposed their 96 features into `act`
        ctx.enc_keepalive = enc if enc.shape[-1] > 96 else None
        ctx.nctx, ctx.M, ctx.sizes, ctx.mlp = nctx, M, sz, mlp
        ctx.shapes = [p.shape for p in params]
        return raw

    @staticmethod
    def backward(ctx, d_raw):
        act, masks = ctx.saved_tensors
        nctx = ctx.nctx
        dev = act.device
        d_raw = d_raw.contiguous().float()
        delta = nctx.scratch("delta", ctx.sizes[2])
        partials = nctx.scratch("partials", ctx.sizes[3])
        total = nctx.grad_numel(ctx.shapes)
        mlp = ctx.mlp
        hooked = any(getattr(p, "_backward_hooks", None) for p in mlp.ordered_params())     # somebody wants to SEE the gradients
        if mlp.grads_are_flat() and not hooked:
            # flat mode (MLP.flatten_parameters): the split reduction writes / adds straight into the buffer the
            # parameters' .grad alias -- no per-tensor gradient tensors, no autograd accumulation kernels
            L.check(L.lib().mipnerf_mlp_backward(nctx.handle, ctx.M, d_raw.data_ptr(), act.data_ptr(), masks.data_ptr(),
                                                 delta.data_ptr(), partials.data_ptr(), mlp._flat_grad.data_ptr(),
                                                 1 if mlp._flat_grad_valid else 0, ops._stream()), "mlp_backward")
            mlp._flat_grad_valid = True
            return (None, None, None, None) + (None,) * len(ctx.shapes)
        grad_flat = torch.empty(total, device=dev, dtype=torch.float32)
        L.check(L.lib().mipnerf_mlp_backward(nctx.handle, ctx.M, d_raw.data_ptr(), act.data_ptr(), masks.data_ptr(),
                                             delta.data_ptr(), partials.data_ptr(), grad_flat.data_ptr(), 0, ops._stream()),
                "mlp_backward")
        return (None, None, None, None, *nctx.split_grads(grad_flat, ctx.shapes))


class _MLPNativeF32(torch.autograd.Function):
    """Parity-mode (fp32) MLP under autograd: fused exact-fp32 MFMA forward that saves every layer output, backward =
    dgrad / wgrad GEMMs on v_mfma_f32_32x32x2_f32 (kernels_gemm_f32.hip).  No library GEMM."""

    @staticmethod
    def forward(ctx, mlp, enc, venc, *params):
        dev = enc.device
        nctx = mlp.native(dev)
        B, N = enc.shape[0], enc.shape[1]
        M = B * N
        if enc.dtype != torch.float32 or venc.dtype != torch.float32 or venc.shape[-1] != 32:
            raise TypeError("fp32 MLP training path: enc [B,N,96] and viewenc [B,32] must be float32")
        enc, venc = enc.contiguous(), venc.contiguous()
        import ctypes as C
        sb, wb = C.c_size_t(), C.c_size_t()
        L.lib().mipnerf_mlp_train_f32_bytes(nctx.handle, M, C.byref(sb), C.byref(wb))
        save = torch.empty(int(sb.value), dtype=torch.uint8, device=dev)
        raw = torch.empty(B, N, 4, device=dev, dtype=torch.float32)
        rgb_sigma = torch.empty_like(raw)
        L.check(L.lib().mipnerf_mlp_forward_train_f32(nctx.handle, M, N, enc.data_ptr(), venc.data_ptr(), rgb_sigma.data_ptr(),
                                                      raw.data_ptr(), save.data_ptr(), ops._stream()), "mlp_forward_train_f32")
        ctx.save_for_backward(save, enc, venc)
        ctx.nctx, ctx.M, ctx.N, ctx.ws_bytes = nctx, M, N, int(wb.value)
        ctx.shapes = [p.shape for p in params]
        return raw

    @staticmethod
    def backward(ctx, d_raw):
        save, enc, venc = ctx.saved_tensors
        nctx = ctx.nctx
        d_raw = d_raw.contiguous().float()
        ws = nctx.scratch("f32_bwd", ctx.ws_bytes)
        grad_flat = torch.empty(nctx.grad_numel(ctx.shapes), device=save.device, dtype=torch.float32)
        d_enc = None
        if ctx.needs_input_grad[1]:        # the encoding is in the graph (stop_resample_grad=False): input gradient too
            d_enc = torch.empty_like(enc)
            L.check(L.lib().mipnerf_mlp_backward_f32_enc(nctx.handle, ctx.M, ctx.N, d_raw.data_ptr(), enc.data_ptr(),
                                                         venc.data_ptr(), save.data_ptr(), ws.data_ptr(), grad_flat.data_ptr(), 0,
                                                         d_enc.data_ptr(), ops._stream()), "mlp_backward_f32_enc")
        else:
            L.check(L.lib().mipnerf_mlp_backward_f32(nctx.handle, ctx.M, ctx.N, d_raw.data_ptr(), enc.data_ptr(), venc.data_ptr(),
                                                     save.data_ptr(), ws.data_ptr(), grad_flat.data_ptr(), 0, ops._stream()),
                    "mlp_backward_f32")
        return (None, d_enc, None, *nctx.split_grads(grad_flat, ctx.shapes))


class _NativeStepLoss(torch.autograd.Function):
    """The whole training step's loss as ONE autograd node (MipNerf.loss_native): forward = mipnerf_train_step (forward of both levels,
    the loss of nerf_system.py:99-111, compositing / MLP backward, weight gradients) into a private flat gradient buffer; backward =
    that buffer times the incoming scalar gradient, sliced into the parameters' shapes.  Nothing is recomputed and no activation is kept
    alive between the two: what the node saves is the 2.45-MB gradient."""

    @staticmethod
    def forward(ctx, model, rays, gt_rgb, cfg, draws, *params):
        dev = rays.origins.device
        mlp = model.mlp
        nctx = mlp.native(dev)
        shapes = [p.shape for p in params]
        grad = torch.empty(nctx.grad_numel(shapes), device=dev, dtype=torch.float32)
        randomized, white_bkgd, clm, dlm, dml = cfg
        scalars, _ = model._train_step_call(rays, gt_rgb, randomized, white_bkgd, clm, dlm, dml, draws[0], draws[1], draws[2], grad, 0)
        ctx.nctx, ctx.shapes, ctx.mlp = nctx, shapes, mlp
        ctx.save_for_backward(grad)
        ctx.mark_non_differentiable(scalars)
        return scalars[0].clone(), scalars

    @staticmethod
    def backward(ctx, g_loss, _g_scalars):
        (grad,) = ctx.saved_tensors
        flat = grad * g_loss                                     # one launch over the flat buffer (dL/dloss is 1 under loss.backward())
        mlp = ctx.mlp
        hooked = any(getattr(p, "_backward_hooks", None) for p in mlp.ordered_params())
        if mlp.grads_are_flat() and not hooked:
            # flat mode (MLP.flatten_parameters: this package's own loops): .grad aliases the flat buffer; add / write there, nothing to return
            if mlp._flat_grad_valid:
                mlp._flat_grad.add_(flat)
            else:
                mlp._flat_grad.copy_(flat)
            mlp._flat_grad_valid = True
            return (None,) * 5 + (None,) * len(ctx.shapes)
        return (None,) * 5 + tuple(ctx.nctx.split_grads(flat, ctx.shapes))


def native_step_loss(model, rays, gt_rgb, randomized, white_bkgd, coarse_loss_mult, distloss_mult, disable_multiscale_loss,
                     t_rand=None, u_rand=None, density_randn=None):
    cfg = (bool(randomized), bool(white_bkgd), float(coarse_loss_mult), float(distloss_mult), bool(disable_multiscale_loss))
    return _NativeStepLoss.apply(model, rays, gt_rgb, cfg, (t_rand, u_rand, density_randn), *model.mlp.ordered_params())


def mlp_native_f32(mlp, samples_enc, viewdirs_enc):
    """Differentiable fp32 MLP: samples_enc [B,N,96] fp32, viewdirs_enc [B,32] fp32 (27 + zero pad) -> raw [B,N,4]."""
    return _MLPNativeF32.apply(mlp, samples_enc, viewdirs_enc, *mlp.ordered_params())


# The unbounded model's bf16 training forward writes its 672-wide encoding as MFMA fragments (faster for k_pre_gemm and the weight-gradient
# jobs).  False = row-major rows, the layout of the per-stage C ABI: same arithmetic, same bits -- the parity tests run both and compare.
FRAGMENT_ENCODINGS = True


def mlp_native(mlp, samples_enc, viewdirs_enc, frag_shape=None):
    """Differentiable bf16 MLP: samples_enc [B,N,xyz_dim] bf16 (or, frag_shape = (B, N), the fragment buffer of
    ops.cast_ipe_360(fragments=True)), viewdirs_enc [B,32] bf16 -> raw [B,N,4] fp32."""
    return _MLPNative.apply(mlp, samples_enc, viewdirs_enc, frag_shape, *mlp.ordered_params())


class _CastIPE(torch.autograd.Function):
    """cast_rays + integrated_pos_enc (mip.py:81-103, 322-350), fp32, differentiable w.r.t. t_samples: the route by which the
    fine level's loss reaches the resampled fence posts when stop_resample_grad=False."""

    @staticmethod
    def forward(ctx, t_samples, origins, directions, radii, min_deg, max_deg, disable_integration):
        t = ops._f32c(t_samples, "t_samples")
        o, d, r = ops._f32c(origins, "origins"), ops._f32c(directions, "directions"), ops._f32c(radii, "radii")
        enc = ops.cast_ipe(t, o, d, r, min_deg, max_deg, disable_integration, precision=L.PREC_FP32)
        ctx.save_for_backward(t, o, d, r)
        ctx.cfg = (int(min_deg), int(max_deg), int(bool(disable_integration)))
        return enc

    @staticmethod
    def backward(ctx, d_enc):
        t, o, d, r = ctx.saved_tensors
        B, N = t.shape[0], t.shape[1] - 1
        g = d_enc.contiguous().float()
        d_t = torch.zeros_like(t)
        L.check(L.lib().mipnerf_cast_ipe_bwd(B, N, ctx.cfg[0], ctx.cfg[1], ctx.cfg[2], t.data_ptr(), o.data_ptr(), d.data_ptr(),
                                             r.data_ptr(), g.data_ptr(), d_t.data_ptr(), ops._stream()), "cast_ipe_bwd")
        return d_t, None, None, None, None, None, None


class _CastRaysNoBackward(torch.autograd.Function):
    """cast_rays(t) of ops.resample_along_rays(stop_grad=False): forward only.  A gradient arriving here means the caller
    differentiates means / covs w.r.t. the resampled t outside MipNerf (which routes that link through _CastIPE)."""

    @staticmethod
    def forward(ctx, t, origins, directions, radii, ray_shape):
        from . import ops
        means, covs = ops.cast_rays(t.detach(), origins, directions, radii, ray_shape)
        return means, covs

    @staticmethod
    def backward(ctx, g_means, g_covs):
        raise NotImplementedError("gradient of cast_rays w.r.t. the resampled t_samples: use ops.cast_ipe / MipNerf "
                                  "(stop_resample_grad=False, fp32), whose fused cast_rays + integrated_pos_enc has a native backward")


class _ResampleT(torch.autograd.Function):
    """The t part of resample_along_rays (mip.py:232-280: blur pool, padding, sorted_piecewise_constant_pdf), differentiable
    w.r.t. the coarse weights (stop_grad=False branch, mip.py:265-279)."""

    @staticmethod
    def forward(ctx, t_samples, weights, padding, u_rand):
        t, w = ops._f32c(t_samples, "t_samples"), ops._f32c(weights, "weights")
        u = None if u_rand is None else ops._f32c(u_rand, "u_rand")
        t_new = ops.resample_t(t, w, u is not None, padding, u)
        ctx.save_for_backward(t, w, *([u] if u is not None else []))
        ctx.padding = float(padding)
        return t_new

    @staticmethod
    def backward(ctx, d_t_new):
        t, w = ctx.saved_tensors[0], ctx.saved_tensors[1]
        u = ctx.saved_tensors[2] if len(ctx.saved_tensors) > 2 else None
        B, N = w.shape
        g = d_t_new.contiguous().float()
        d_w = torch.empty_like(w)
        L.check(L.lib().mipnerf_resample_along_rays_bwd(B, N, t.data_ptr(), w.data_ptr(), None if u is None else u.data_ptr(),
                                                        ctx.padding, g.data_ptr(), d_w.data_ptr(), ops._stream()),
                "resample_along_rays_bwd")
        return None, d_w, None, None


def distloss(weight, samples):
    """Drop-in for models/mip.py:8-20: scalar distortion loss (t_samples must be sorted, as they always are)."""
    return _DistLossRays.apply(weight, samples).mean()


def render_from_raw(raw, t_samples, dirs, white_bkgd, rgb_padding=0.001, density_bias=-1.0, density_randn=None,
                    density_noise=0.0):
    return _RenderFromRaw.apply(raw, t_samples, dirs, white_bkgd, rgb_padding, density_bias, density_randn, density_noise)


def mipnerf_forward_train(model, rays, randomized, white_bkgd, t_rand=None, u_rand=None, density_randn=None):
    """Differentiable MipNerf.forward (mip_nerf.py:172-248): list of (comp_rgb, distance, acc, weights, t_samples)."""
    dev = rays.origins.device
    native = model.precision == L.PREC_BF16
    N = model.num_samples
    model.mlp.native(dev)      # raises NotImplementedError for an MLP shape the kernels were not generated for
    dz = model._density_randn(randomized, rays.origins.shape[0], dev, density_randn)
    if dz is not None:
        dz = dz.reshape(model.num_levels, -1)
    with torch.no_grad():
        venc = ops.pos_enc(rays.viewdirs, 0, model.deg_view, True, precision=model.precision, ld=32)
    through = not model.stop_resample_grad          # mip.py:265-279: keep the resampler in the autograd graph
    if through and native:
        raise NotImplementedError("stop_resample_grad=False needs the gradient w.r.t. the MLP's input encoding, which the bf16 "
                                  "dgrad kernel does not produce; use precision='fp32' for this option")
    ret = []
    t_samples, weights, t_inv = None, None, None
    unbounded = getattr(model, "unbounded", False)
    for lvl in range(model.num_levels):
        if unbounded:
            # same sequence as the unbounded branch of mipnerf_forward (capi.hip), no gradient through the sampler
            with torch.no_grad():
                if lvl == 0:
                    t_inv, t_samples = ops.sample_t_360(N, rays.near, rays.far, randomized, t_rand)
                else:
                    t_inv = ops.resample_t(t_inv, weights.detach(), randomized, model.resample_padding, u_rand)
                    t_samples = 1.0 / t_inv
                # bf16: the encoding as MFMA fragments (k_pre_gemm and the weight-gradient jobs read them faster than rows: 7.4 -> 6.9 ms per step)
                enc = ops.cast_ipe_360(t_samples, rays.origins, rays.directions, rays.radii, model.min_deg_point, model.max_deg_point,
                                       contracted=True, precision=model.precision, fragments=native and FRAGMENT_ENCODINGS)
        elif through and lvl > 0:
            B = rays.origins.shape[0]
            u = None
            if randomized:
                u = torch.rand(B, N + 1, device=dev) if u_rand is None else u_rand
            t_samples = _ResampleT.apply(t_samples, weights, model.resample_padding, u)
            enc = _CastIPE.apply(t_samples, rays.origins, rays.directions, rays.radii, model.min_deg_point, model.max_deg_point,
                                 model.disable_integration)
        else:
            with torch.no_grad():
                if lvl == 0:
                    t_samples = ops.sample_t(N, rays.near, rays.far, randomized, model.disparity, t_rand)
                else:
                    t_samples = ops.resample_t(t_samples, weights.detach(), randomized, model.resample_padding, u_rand)
                enc = ops.cast_ipe(t_samples, rays.origins, rays.directions, rays.radii, model.min_deg_point,
                                   model.max_deg_point, model.disable_integration, precision=model.precision)
        if native:
            raw = mlp_native(model.mlp, enc, venc, frag_shape=(t_samples.shape[0], N) if (unbounded and FRAGMENT_ENCODINGS) else None)
        else:
            raw = mlp_native_f32(model.mlp, enc, venc)
        comp_rgb, distance, acc, weights = render_from_raw(raw, t_samples, rays.directions, white_bkgd,
                                                           model.rgb_padding, model.density_bias,
                                                           None if dz is None else dz[lvl], model.density_noise)
        ret.append((comp_rgb, distance, acc, weights, t_samples))
    return ret
