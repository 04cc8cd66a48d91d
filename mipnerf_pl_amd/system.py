"""`MipNeRFSystem`: host-side mirror of the reference LightningModule (models/nerf_system.py:13-177) with
`self.mip_nerf` swapped for the MI355X-native module.  Hooks, hyper-parameter keys, loss and the chunked
`render_image` follow the reference line by line; Lightning itself stays third-party: if
`pytorch_lightning` is importable the class derives from `LightningModule`, otherwise from a minimal
shim (nn.Module + hparams/log) so the hooks can be driven by a plain loop (bench.py --mode train).
`setup` / `train_dataloader` / `val_dataloader` read the reference's dataset directories through
`mipnerf_pl_amd.datasets` (images + camera table resident on the device, rays generated per batch by the HIP kernel; SURVEY
8f-1) and hand Lightning a device-side `RayLoader` instead of a host `DataLoader`."""
from __future__ import annotations

import torch

from .autograd import distloss
from .lr_schedule import MipLRDecay
from .model import MipNerf
from .rays import Rays, Rays_keys

try:  # pragma: no cover - not installed in the build container
    from pytorch_lightning import LightningModule as _Base
    _HAVE_PL = True
except Exception:  # noqa: BLE001
    _HAVE_PL = False

    class _Base(torch.nn.Module):
        """The few LightningModule facilities the hooks use."""

        def __init__(self):
            super().__init__()
            self.hparams = {}
            self.logged = {}
            self.global_step = 0

        def save_hyperparameters(self, hparams):
            self.hparams = dict(hparams)

        def log(self, name, value, **kw):
            self.logged[name] = value.detach() if torch.is_tensor(value) else value

        # Lightning checkpoint format (what the reference's train.py writes and eval.py / render_video.py read with
        # MipNeRFSystem.load_from_checkpoint): {'state_dict': {'mip_nerf.mlp...': tensor}, 'hyper_parameters': {...}}
        def save_checkpoint(self, path):
            torch.save({"state_dict": {k: v.detach().cpu() for k, v in self.state_dict().items()},
                        "hyper_parameters": dict(self.hparams), "global_step": self.global_step}, path)

        @classmethod
        def load_from_checkpoint(cls, checkpoint_path, map_location=None, **kwargs):
            ckpt = torch.load(checkpoint_path, map_location=map_location or "cpu", weights_only=False)
            hp = dict(ckpt.get("hyper_parameters", {}))
            hp.update(kwargs.pop("hparams", {}) or {})
            system = cls(hp, **kwargs)
            system.load_state_dict(ckpt["state_dict"], strict=True)
            system.global_step = int(ckpt.get("global_step", 0))
            return system


def calc_psnr(x: torch.Tensor, y: torch.Tensor):
    """utils/metrics.py:182-188."""
    return -10.0 * torch.log10(torch.mean((x - y) ** 2))


def rearrange_render_image(rays, chunk_size=4096):
    """models/mip.py:404-421: flatten [1,H,W,k] rays and cut them into chunks; val_mask = lossmult (pre-flatten)."""
    single_image_rays = [getattr(rays, key) for key in Rays_keys]
    val_mask = single_image_rays[-3]
    single_image_rays = [a.reshape(-1, a.shape[-1]) for a in single_image_rays]
    length = single_image_rays[0].shape[0]
    chunks = [[a[i:i + chunk_size] for i in range(0, length, chunk_size)] for a in single_image_rays]
    n = len(chunks[0])
    return [Rays(*[c[i] for c in chunks]) for i in range(n)], val_mask


DEFAULT_HPARAMS = {   # configs/lego.yaml, flattened like configs/config.py:62-92
    'train.randomized': True, 'val.randomized': False, 'train.white_bkgd': True, 'val.white_bkgd': True,
    'val.chunk_size': 8192, 'train.batch_size': 3072, 'train.num_work': 4,
    'nerf.num_samples': 128, 'nerf.num_levels': 2, 'nerf.resample_padding': 0.01, 'nerf.stop_resample_grad': True,
    'nerf.use_viewdirs': True, 'nerf.disparity': False, 'nerf.ray_shape': 'cone', 'nerf.min_deg_point': 0,
    'nerf.max_deg_point': 16, 'nerf.deg_view': 4, 'nerf.density_activation': 'softplus', 'nerf.density_noise': 0.,
    'nerf.density_bias': -1., 'nerf.rgb_activation': 'sigmoid', 'nerf.rgb_padding': 0.001,
    'nerf.disable_integration': False, 'nerf.append_identity': True, 'nerf.mlp.net_depth': 8,
    'nerf.mlp.net_width': 256, 'nerf.mlp.net_depth_condition': 1, 'nerf.mlp.net_width_condition': 128,
    'nerf.mlp.skip_index': 4, 'nerf.mlp.num_rgb_channels': 3, 'nerf.mlp.num_density_channels': 1,
    'nerf.mlp.net_activation': 'relu', 'optimizer.lr_init': 5e-4, 'optimizer.lr_final': 5e-6,
    'optimizer.lr_delay_steps': 2500, 'optimizer.lr_delay_mult': 0.01, 'optimizer.max_steps': 1000000,
    'loss.disable_multiscale_loss': False, 'loss.coarse_loss_mult': 0.1,
    'nerf.unbounded': False,      # not a reference key: the unbounded-scene (mip-NeRF 360) path, MipNerf(unbounded=True)
}


class MipNeRFSystem(_Base):
    def __init__(self, hparams, precision=None):
        super().__init__()
        self.save_hyperparameters(hparams)
        hp = hparams
        self.train_randomized = hp['train.randomized']
        self.val_randomized = hp['val.randomized']
        self.white_bkgd = hp['train.white_bkgd']
        self.val_chunk_size = hp['val.chunk_size']
        self.batch_size = hp['train.batch_size']
        self.mip_nerf = MipNerf(   # nerf_system.py:22-48, same keyword list
            num_samples=hp['nerf.num_samples'], num_levels=hp['nerf.num_levels'],
            resample_padding=hp['nerf.resample_padding'], stop_resample_grad=hp['nerf.stop_resample_grad'],
            use_viewdirs=hp['nerf.use_viewdirs'], disparity=hp['nerf.disparity'], ray_shape=hp['nerf.ray_shape'],
            min_deg_point=hp['nerf.min_deg_point'], max_deg_point=hp['nerf.max_deg_point'],
            deg_view=hp['nerf.deg_view'], density_activation=hp['nerf.density_activation'],
            density_noise=hp['nerf.density_noise'], density_bias=hp['nerf.density_bias'],
            rgb_activation=hp['nerf.rgb_activation'], rgb_padding=hp['nerf.rgb_padding'],
            disable_integration=hp['nerf.disable_integration'], append_identity=hp['nerf.append_identity'],
            mlp_net_depth=hp['nerf.mlp.net_depth'], mlp_net_width=hp['nerf.mlp.net_width'],
            mlp_net_depth_condition=hp['nerf.mlp.net_depth_condition'],
            mlp_net_width_condition=hp['nerf.mlp.net_width_condition'], mlp_skip_index=hp['nerf.mlp.skip_index'],
            mlp_num_rgb_channels=hp['nerf.mlp.num_rgb_channels'],
            mlp_num_density_channels=hp['nerf.mlp.num_density_channels'],
            mlp_net_activation=hp['nerf.mlp.net_activation'],
            # (Lightning's load_from_checkpoint(path, precision=...) merges extra keywords INTO the stored hyper-parameters instead of
            # passing them to the constructor, so the precision is also read from there)
            precision=precision or hp.get('nerf.precision') or hp.get('precision'), unbounded=bool(hp.get('nerf.unbounded', False)))

    def forward(self, batch_rays, randomized: bool, white_bkgd: bool):
        return self.mip_nerf(batch_rays, randomized, white_bkgd)     # nerf_system.py:50-54

    def setup(self, stage=None):   # nerf_system.py:56-68
        from .datasets import dataset_dict
        dataset = dataset_dict[self.hparams['dataset_name']]
        dev = next(self.mip_nerf.parameters()).device          # the datasets live where the model lives
        dev = dev if dev.type == "cuda" else None              # (None: the current HIP device, if any)
        self.train_dataset = dataset(data_dir=self.hparams['data_path'], split='train',
                                     white_bkgd=self.hparams['train.white_bkgd'],
                                     batch_type=self.hparams['train.batch_type'], device=dev)
        self.val_dataset = dataset(data_dir=self.hparams['data_path'], split='val',
                                   white_bkgd=self.hparams['val.white_bkgd'],
                                   batch_type=self.hparams['val.batch_type'], device=dev)

    def configure_optimizers(self):   # nerf_system.py:70-76
        hp = self.hparams
        sched = dict(lr_init=hp['optimizer.lr_init'], lr_final=hp['optimizer.lr_final'], max_steps=hp['optimizer.max_steps'],
                     lr_delay_steps=hp['optimizer.lr_delay_steps'], lr_delay_mult=hp['optimizer.lr_delay_mult'])
        if getattr(self, "fused_adam", False) and getattr(self, "device_lr_schedule", True):
            # same update rule + same schedule, evaluated on the device: one tiny kernel (step count, lr, bias corrections)
            # + one Adam kernel over the flat parameter buffer; nothing per-step comes from the host (graph-capturable)
            from .lr_schedule import DeviceMipLRDecay
            from .optim import FlatAdam
            optimizer = FlatAdam(self.mip_nerf.mlp, lr=hp['optimizer.lr_init'], schedule=sched)
            scheduler = DeviceMipLRDecay(optimizer, **sched)
            self._optimizer_for_log = optimizer
            return [optimizer], [{'scheduler': scheduler, 'interval': 'step'}]
        if getattr(self, "fused_adam", False):
            # same update rule as torch.optim.Adam below, one HIP kernel over the flat parameter buffer
            from .optim import FlatAdam
            optimizer = FlatAdam(self.mip_nerf.mlp, lr=self.hparams['optimizer.lr_init'])
        else:
            optimizer = torch.optim.Adam(self.mip_nerf.parameters(), lr=self.hparams['optimizer.lr_init'])
        scheduler = MipLRDecay(optimizer, self.hparams['optimizer.lr_init'], self.hparams['optimizer.lr_final'],
                               self.hparams['optimizer.max_steps'], self.hparams['optimizer.lr_delay_steps'],
                               self.hparams['optimizer.lr_delay_mult'])
        self._optimizer_for_log = optimizer
        return [optimizer], [{'scheduler': scheduler, 'interval': 'step'}]

    def lr_scheduler_step(self, scheduler, *args):
        """Lightning hook (signature differs between 1.x and 2.x: optimizer_idx / metric): the schedules returned above are step-wise
        torch LRSchedulers, so stepping them is all there is to do -- stated explicitly so that no Lightning version has to guess."""
        scheduler.step()

    def train_dataloader(self):   # nerf_system.py:78-83: shuffled batches of `train.batch_size` rays, drawn on the device
        from .datasets import RayLoader
        return RayLoader(self.train_dataset, batch_size=self.hparams['train.batch_size'], shuffle=True)

    def val_dataloader(self):     # nerf_system.py:85-93: one whole image per item
        from .datasets import RayLoader
        return RayLoader(self.val_dataset, batch_size=1, shuffle=False)

    def compute_loss(self, ret, rays, rgbs):
        """nerf_system.py:99-111."""
        mask = rays.lossmult
        if self.hparams['loss.disable_multiscale_loss']:
            mask = torch.ones_like(mask)
        losses, distlosses = [], []
        for (rgb, _, _, weights, t_samples) in ret:
            losses.append((mask * (rgb - rgbs[..., :3]) ** 2).sum() / mask.sum())
            distlosses.append(distloss(weights, t_samples))
        mse_corse, mse_fine = losses
        loss = self.hparams['loss.coarse_loss_mult'] * (mse_corse + 0.01 * distlosses[0]) \
            + mse_fine + 0.01 * distlosses[-1]
        return loss, losses, distlosses

    def training_step(self, batch, batch_nb):   # nerf_system.py:95-121
        rays, rgbs = batch
        if self._native_step_route(rays):
            # round 6: the same loss as ONE autograd node whose forward is the one-call native step (forward + loss + the 3-kernel bf16
            # backward) and whose backward hands the finished gradient to the parameters: Lightning's automatic optimisation
            # (loss.backward() -> any torch optimizer; DDP's reducer hooks fire as usual) runs the fast path without opting into anything
            loss, scalars = self.mip_nerf.loss_native(
                rays, rgbs, self.train_randomized, self.white_bkgd, coarse_loss_mult=self.hparams['loss.coarse_loss_mult'],
                disable_multiscale_loss=self.hparams['loss.disable_multiscale_loss'])
            psnr_fine = scalars[5]
        else:
            ret = self(rays, self.train_randomized, self.white_bkgd)
            loss, _, _ = self.compute_loss(ret, rays, rgbs)
            with torch.no_grad():
                psnr_fine = calc_psnr(ret[-1][0], rgbs[..., :3])
        self.log('train/loss', loss)
        self.log('train/psnr', psnr_fine, prog_bar=True)
        self._log_lr()
        return loss

    def _native_step_route(self, rays) -> bool:
        """training_step goes through MipNerf.loss_native when the one-call native step serves this configuration (bf16, stop-gradient
        resampler, an MLP shape with bf16 training kernels, rays on the GPU, gradients wanted).  `self.native_training_step = False` (or
        MIPNERF_NATIVE_TRAINING_STEP=0) keeps the per-stage autograd Functions -- same kernels underneath, the route of the fp32 parity mode."""
        import os
        if not getattr(self, "native_training_step", os.environ.get("MIPNERF_NATIVE_TRAINING_STEP", "1") != "0"):
            return False
        if not (torch.is_grad_enabled() and rays.origins.is_cuda and rays.origins.shape[0] > 0):
            return False
        if not all(p.requires_grad for p in self.mip_nerf.parameters()):
            return False
        key = (self.mip_nerf.precision, rays.origins.device)
        cache = self.__dict__.setdefault("_native_route_ok", {})
        if key not in cache:
            cache[key] = self.mip_nerf.native_step_supported(rays.origins.device)
        return cache[key]

    def _log_lr(self):
        """nerf_system.py:117 `self.log('lr', ...)`: the learning rate of the first param group of the configured optimiser
        (a host float; with the device-side schedule it is the host mirror, no synchronisation)."""
        opt = getattr(self, "_optimizer_for_log", None)
        if opt is not None:
            self.log('lr', opt.param_groups[0]['lr'])

    def training_step_native(self, batch, batch_nb=0):
        """training_step + loss.backward() in one native call (no autograd graph; bf16): the gradients are in
        `.grad` when it returns.  Returns the detached loss; logs like training_step."""
        rays, rgbs = batch
        scalars, _ = self.mip_nerf.train_step_native(
            rays, rgbs, self.train_randomized, self.white_bkgd, coarse_loss_mult=self.hparams['loss.coarse_loss_mult'],
            disable_multiscale_loss=self.hparams['loss.disable_multiscale_loss'])
        self.log('train/loss', scalars[0])
        self.log('train/psnr', scalars[5], prog_bar=True)
        self._log_lr()
        return scalars[0]

    def validation_step(self, batch, batch_nb):   # nerf_system.py:123-142 (TensorBoard images left to the caller)
        _, rgbs = batch
        rgb_gt = rgbs[..., :3]
        coarse_rgb, fine_rgb, val_mask = self.render_image(batch)
        val_mse_coarse = (val_mask * (coarse_rgb - rgb_gt) ** 2).sum() / val_mask.sum()
        val_mse_fine = (val_mask * (fine_rgb - rgb_gt) ** 2).sum() / val_mask.sum()
        val_loss = self.hparams['loss.coarse_loss_mult'] * val_mse_coarse + val_mse_fine
        return {'val/loss': val_loss, 'val/psnr': calc_psnr(fine_rgb, rgb_gt)}

    def validation_epoch_end(self, outputs):     # nerf_system.py:144-149
        self.log('val/loss', torch.stack([x['val/loss'] for x in outputs]).mean())
        self.log('val/psnr', torch.stack([x['val/psnr'] for x in outputs]).mean(), prog_bar=True)

    def enable_hip_graph(self, on: bool = True):
        """Render chunks through a captured hipGraph (one capture per chunk size; not in the reference)."""
        self.use_hip_graph = bool(on)
        self._graphed = None
        return self

    def render_image(self, batch, return_distance=False):   # nerf_system.py:151-177
        rays, rgbs = batch
        _, height, width, _ = rgbs.shape
        if getattr(self, "use_hip_graph", False) and not self.val_randomized and rays.origins.is_cuda:
            # the whole frame = one captured hipGraph over static full-frame buffers (model.GraphedFrame)
            from .model import GraphedFrame
            flat = Rays(*[getattr(rays, k).reshape(-1, getattr(rays, k).shape[-1]) for k in Rays_keys])
            val_mask = rays.lossmult
            n = flat.origins.shape[0]
            g = getattr(self, "_graphed", None)
            if (not isinstance(g, GraphedFrame)) or g.n != n or g.chunk != self.val_chunk_size or g.white_bkgd != bool(self.white_bkgd):
                g = self._graphed = GraphedFrame(self.mip_nerf, n, self.val_chunk_size, self.white_bkgd, flat.origins.device)
            c_rgb, f_rgb, dist = g(flat)
            coarse_rgb, fine_rgb = c_rgb.clone().reshape(1, height, width, -1), f_rgb.clone().reshape(1, height, width, -1)
            if return_distance:
                return coarse_rgb, fine_rgb, val_mask, dist.clone().reshape(1, height, width)
            return coarse_rgb, fine_rgb, val_mask
        single_image_rays, val_mask = rearrange_render_image(rays, self.val_chunk_size)
        coarse_rgb, fine_rgb, distances = [], [], []
        with torch.no_grad():
            for batch_rays in single_image_rays:
                (c_rgb, _, _, _, _), (f_rgb, distance, _, _, _) = self(batch_rays, self.val_randomized, self.white_bkgd)
                coarse_rgb.append(c_rgb)
                fine_rgb.append(f_rgb)
                distances.append(distance)
        coarse_rgb = torch.cat(coarse_rgb, dim=0).reshape(1, height, width, -1)
        fine_rgb = torch.cat(fine_rgb, dim=0).reshape(1, height, width, -1)
        if return_distance:
            return coarse_rgb, fine_rgb, val_mask, torch.cat(distances, dim=0).reshape(1, height, width)
        return coarse_rgb, fine_rgb, val_mask
