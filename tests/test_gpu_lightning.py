"""The Lightning contract of `MipNeRFSystem`, EXECUTED: system.py's `_HAVE_PL` branch under the strict stand-in for
pytorch_lightning 1.5.2 of tests/lightning_standin.py -- read-only `hparams` / `global_step` / `device`, `self.log` legal only inside a hook,
automatic optimisation in Lightning's call order, Lightning's checkpoint keys and its `load_from_checkpoint` constructor protocol -- driving
what the reference's train.py:34-64 and eval.py:28 drive: configure_optimizers, training_step x 3, validation_step / validation_epoch_end,
checkpoint save -> load_from_checkpoint -> render."""
import numpy as np
import pytest
import torch

import lightning_standin as pl
from oracle import mipnerf_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def G():
    import gpu_util
    assert torch.cuda.is_available()
    return gpu_util


def _system(G, g, **hp_over):
    mod = pl.system_module_under_lightning()
    hp = dict(mod.DEFAULT_HPARAMS)
    hp.update({'nerf.num_samples': 64, 'train.randomized': True, 'optimizer.lr_init': 1e-3, 'optimizer.lr_delay_steps': 0, 'exp_name': 'standin',
               'val.chunk_size': 40})
    hp.update(hp_over)
    params = orc.make_params(seed=int(g["param_seed"]), density_gain=float(g["density_gain"]))
    system = mod.MipNeRFSystem(hp)
    system.load_state_dict({"mip_nerf.mlp." + k: torch.from_numpy(v.copy()) for k, v in params.items()})
    return mod, system.to(DEV)


def test_fit_validate_checkpoint_under_the_strict_lightning_module(G, tmp_path):
    g = G.load_golden("train_64x64_trained")
    mod, system = _system(G, g)
    assert isinstance(system, pl.LightningModule) and system.device == torch.device(DEV)
    assert system.hparams['nerf.num_samples'] == 64 and system._hparams_name == "hparams"      # the constructor argument train.py passes
    rays, gt = G.to_dev(G.rays_of(g)), torch.from_numpy(g["gt"]).to(DEV)
    with pytest.raises(pl.MisconfigurationException):
        system.training_step((rays, gt), 0)                  # self.log outside the trainer's control flow is an error in Lightning
    drv = pl.LoopDriver(system)
    assert type(drv.optimizer) is torch.optim.Adam and len(drv.optimizer.param_groups[0]["params"]) == 24      # nerf_system.py:70-72
    torch.manual_seed(5)
    before = [p.detach().clone() for p in system.mip_nerf.parameters()]
    losses = [float(drv.fit_batch((rays, gt), i)) for i in range(3)]
    assert system.global_step == 3 and drv.scheduler.last_epoch == 3
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]
    assert all(not torch.equal(a, b) for a, b in zip(before, system.mip_nerf.parameters()))
    logged = drv.logged["training_step"]
    assert set(logged) == {"train/loss", "train/psnr", "lr"} and "train/psnr" in drv.progress_bar          # nerf_system.py:113-119
    assert abs(float(logged["train/loss"]) - losses[-1]) < 1e-7
    assert abs(float(logged["lr"]) - drv.optimizer.param_groups[0]["lr"]) < 1e-12 or float(logged["lr"]) > 0
    assert system._native_step_route(rays)                   # automatic optimisation ran the one-call native step (bf16 default)
    # ---- validation: one [1, H, W] image per item (nerf_system.py:123-149) ----
    H, W = 8, 8
    img_rays = type(rays)(*[x[:H * W].reshape(1, H, W, -1) for x in rays])
    img = gt[:H * W].reshape(1, H, W, 3)
    outs = drv.validate([(img_rays, img), (img_rays, img)])
    assert len(outs) == 2 and set(outs[0]) == {"val/loss", "val/psnr"}
    ve = drv.logged["validation_epoch_end"]
    assert abs(float(ve["val/psnr"]) - float(torch.stack([o["val/psnr"] for o in outs]).mean())) < 1e-6 and "val/psnr" in drv.progress_bar
    # ---- checkpoint: Lightning's keys, the reference's parameter names, then eval.py:28's load ----
    path = str(tmp_path / "last.ckpt")
    ckpt = drv.save_checkpoint(path)
    assert list(ckpt["state_dict"]) == ["mip_nerf.mlp." + k for k in orc.param_shapes()]
    assert ckpt["state_dict"]["mip_nerf.mlp.layers.0.0.weight"].shape == (256, 96) and ckpt["hparams_name"] == "hparams"
    assert ckpt["hyper_parameters"]["nerf.num_samples"] == 64 and ckpt["global_step"] == 3 and len(ckpt["optimizer_states"]) == 1
    again = mod.MipNeRFSystem.load_from_checkpoint(path).to(DEV).eval()                  # eval.py:28 / render_video.py:121
    assert again.hparams['exp_name'] == 'standin' and again.global_step == 0
    for (k1, v1), (k2, v2) in zip(system.state_dict().items(), again.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2)
    with torch.no_grad():
        a = system.eval().render_image((img_rays, img))
        b = again.render_image((img_rays, img))
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    # extra keywords of load_from_checkpoint land INSIDE the hyper-parameters under Lightning (not in the constructor's keyword): honoured
    fp32 = mod.MipNeRFSystem.load_from_checkpoint(path, precision="fp32")
    from mipnerf_pl_amd import _lib as L
    assert fp32.mip_nerf.precision == L.PREC_FP32 and fp32.hparams["precision"] == "fp32"
    # resume: optimiser / scheduler state of the checkpoint restore into freshly configured ones (trainer.fit(ckpt_path=...), train.py:64)
    drv2 = pl.LoopDriver(again)
    drv2.optimizer.load_state_dict(ckpt["optimizer_states"][0])
    drv2.scheduler.load_state_dict(ckpt["lr_schedulers"][0])
    assert drv2.scheduler.last_epoch == 3
    assert abs(drv2.optimizer.param_groups[0]["lr"] - drv.optimizer.param_groups[0]["lr"]) < 1e-12


def test_read_only_lightning_state_is_never_assigned(G):
    """`hparams`, `global_step`, `current_epoch`, `device` are properties without setters on a real LightningModule: construction, .to(),
    the hooks and a checkpoint round trip above never assign them -- and an assignment does raise here, so the stand-in would have caught one"""
    g = G.load_golden("train_64x64_trained")
    _, system = _system(G, g)
    for name in ("hparams", "global_step", "current_epoch", "device"):
        with pytest.raises(AttributeError):
            setattr(system, name, 1)
