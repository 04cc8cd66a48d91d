"""A STRICT stand-in for `pytorch_lightning` 1.5.2 (requirements.txt:1 of the reference) -- test infrastructure only.

pytorch_lightning is in no image of this project, so the `_HAVE_PL` branch of mipnerf_pl_amd/system.py (the class deriving from the
real `LightningModule`, which is what an unmodified train.py / eval.py / render_video.py gets) never ran.  This module restates the
parts of the 1.5.2 `LightningModule` contract that `MipNeRFSystem` touches, with Lightning's STRICTNESS rather than a permissive shim:

* `hparams` is a read-only property; its value is an `AttributeDict` that only `save_hyperparameters` fills (called from `__init__`,
  it inspects the caller's frame to learn the NAME of the constructor argument the dict came in through -> `_hparams_name`);
* `global_step`, `current_epoch`, `device`, `logger` are read-only properties backed by the trainer / the module's own state
  (assigning to them raises, as with the real class);
* `self.log(name, value)` is legal only inside a hook the loop is running and only for single-element values;
* checkpoints have Lightning's keys (`state_dict`, `hyper_parameters`, `hparams_name`, `optimizer_states`, `lr_schedulers`,
  `global_step`, `epoch`, `pytorch-lightning_version`); `load_from_checkpoint` rebuilds the module by passing the stored hyper-parameters
  through the constructor argument named by `hparams_name`, dropping keyword arguments the constructor does not take.

`LoopDriver` is NOT a Trainer: it is the order of calls of 1.5.2's automatic optimisation for one optimiser with a step-interval
scheduler (train.py:48-64 configures nothing else), and the validation hooks.
"""
from __future__ import annotations

import copy
import inspect
import sys
import types

import torch

__version__ = "1.5.2"


class MisconfigurationException(Exception):
    pass


class AttributeDict(dict):
    """utilities/parsing.AttributeDict: keys readable as attributes; a missing key is an AttributeError"""

    def __getattr__(self, key):
        try:
            return self[key]
        except KeyError as exp:
            raise AttributeError(f'Missing attribute "{key}"') from exp

    def __setattr__(self, key, val):
        self[key] = val


def _init_args_of(frame):
    """utilities/parsing.get_init_args: the constructor's own arguments, read from its frame (needs the implicit `__class__` cell, i.e. a
    constructor that mentions `super`)"""
    _, _, _, local_vars = inspect.getargvalues(frame)
    if "__class__" not in local_vars:
        return {}
    cls = local_vars["__class__"]
    params = inspect.signature(cls.__init__).parameters
    names = list(params)
    self_name = names[0]
    var_names = [n for n, p in params.items() if p.kind in (p.VAR_POSITIONAL, p.VAR_KEYWORD)]
    out = {k: v for k, v in local_vars.items() if k in names and k != self_name and k not in var_names}
    for n in var_names:
        if params[n].kind == params[n].VAR_KEYWORD:
            out.update(local_vars.get(n, {}))
    return out


class LightningModule(torch.nn.Module):
    CHECKPOINT_HYPER_PARAMS_KEY = "hyper_parameters"
    CHECKPOINT_HYPER_PARAMS_NAME = "hparams_name"
    CHECKPOINT_HYPER_PARAMS_TYPE = "hparams_type"

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._device = torch.device("cpu")
        self.trainer = None
        self._current_fx_name = None

    # ---- read-only state -------------------------------------------------------------------------------------------------------
    @property
    def hparams(self):
        if not hasattr(self, "_hparams"):
            self._hparams = AttributeDict()
        return self._hparams

    @property
    def hparams_initial(self):
        return copy.deepcopy(getattr(self, "_hparams_initial", AttributeDict()))

    @property
    def global_step(self) -> int:
        return self.trainer.global_step if self.trainer else 0

    @property
    def current_epoch(self) -> int:
        return self.trainer.current_epoch if self.trainer else 0

    @property
    def device(self):
        return self._device

    @property
    def logger(self):
        return self.trainer.logger if self.trainer else None

    def _apply(self, fn):                     # DeviceDtypeModuleMixin: .to() / .cuda() keep `device` current
        out = super()._apply(fn)
        for p in self.parameters():
            self._device = p.device
            break
        return out

    # ---- hyper-parameters ------------------------------------------------------------------------------------------------------
    def save_hyperparameters(self, *args, ignore=None, frame=None, logger=True):
        if not frame:
            frame = inspect.currentframe().f_back
        if len(args) == 1 and not isinstance(args, str) and not args[0]:
            return                                    # an empty container: nothing to save
        init_args = _init_args_of(frame)
        if ignore is not None:
            ignore = [ignore] if isinstance(ignore, str) else list(ignore)
            init_args = {k: v for k, v in init_args.items() if k not in ignore}
        if not args:
            hp = init_args
            self._hparams_name = "kwargs" if hp else None
        else:
            non_str = [i for i, a in enumerate(args) if not isinstance(a, str)]
            if len(non_str) == 1:
                hp = args[non_str[0]]
                cand = [k for k, v in init_args.items() if v == hp]
                self._hparams_name = cand[0] if cand else None
            else:
                hp = {a: init_args[a] for a in args if isinstance(a, str)}
                self._hparams_name = "kwargs"
        self._set_hparams(hp)
        self._hparams_initial = copy.deepcopy(self._hparams)

    def _set_hparams(self, hp):
        if isinstance(hp, types.SimpleNamespace) or hasattr(hp, "__dict__") and not isinstance(hp, dict):
            hp = vars(hp)
        if isinstance(hp, dict):
            hp = AttributeDict(hp)
        elif isinstance(hp, (bool, int, float, str)):
            raise ValueError(f"Primitives {(bool, int, float, str)} are not allowed.")
        else:
            raise ValueError(f"Unsupported config type of {type(hp)}.")
        if isinstance(hp, dict) and isinstance(self.hparams, dict):
            self.hparams.update(hp)
        else:
            self._hparams = hp

    # ---- logging ---------------------------------------------------------------------------------------------------------------
    def log(self, name, value, prog_bar=False, logger=True, on_step=None, on_epoch=None, reduce_fx="mean", **kw):
        if self.trainer is None:
            raise MisconfigurationException("You are trying to `self.log()` but the `self.trainer` reference is not registered on the model yet.")
        if self._current_fx_name is None:
            raise MisconfigurationException("You are trying to `self.log()` but it is not managed by the `Trainer` control flow")
        if isinstance(value, dict):
            raise ValueError(f"`self.log({name}, {value})` was called, but nested dictionaries cannot be logged")
        if not isinstance(value, (torch.Tensor, int, float)):
            raise ValueError(f"`self.log({name}, {value})` was called, but `{type(value).__name__}` values cannot be logged")
        value = value.detach().clone() if torch.is_tensor(value) else torch.tensor(value, device=self.device, dtype=torch.float32)
        if value.numel() != 1:
            raise ValueError(f"`self.log({name}, {value})` was called, but the tensor must have a single element.")
        self.trainer.logged.setdefault(self._current_fx_name, {})[name] = value.squeeze()
        if prog_bar:
            self.trainer.progress_bar.add(name)

    # ---- checkpoints -----------------------------------------------------------------------------------------------------------
    @classmethod
    def load_from_checkpoint(cls, checkpoint_path, map_location=None, hparams_file=None, strict=True, **kwargs):
        checkpoint = torch.load(checkpoint_path, map_location=map_location if map_location is not None else (lambda storage, loc: storage), weights_only=False)
        checkpoint.setdefault(cls.CHECKPOINT_HYPER_PARAMS_KEY, {})
        checkpoint[cls.CHECKPOINT_HYPER_PARAMS_KEY].update(kwargs)
        # core/saving._load_model_state
        spec = inspect.getfullargspec(cls.__init__)
        params = inspect.signature(cls.__init__).parameters
        init_names = [n for i, (n, p) in enumerate(params.items()) if i > 0 and p.kind not in (p.VAR_POSITIONAL, p.VAR_KEYWORD)]
        loaded = {}
        if cls.CHECKPOINT_HYPER_PARAMS_KEY in checkpoint:
            loaded.update(checkpoint.get(cls.CHECKPOINT_HYPER_PARAMS_KEY))
            args_name = checkpoint.get(cls.CHECKPOINT_HYPER_PARAMS_NAME)
            if args_name and args_name in init_names:
                loaded = {args_name: loaded}
        call = dict(loaded)
        if not spec.varkw:
            call = {k: v for k, v in call.items() if k in init_names}      # what the constructor does not name is DROPPED, silently
        model = cls(**call)
        keys = model.load_state_dict(checkpoint["state_dict"], strict=strict)
        if not strict and (keys.missing_keys or keys.unexpected_keys):
            print(f"[lightning stand-in] missing {keys.missing_keys} unexpected {keys.unexpected_keys}", file=sys.stderr)
        return model


class _Logger:
    """TensorBoardLogger's surface the hooks of nerf_system.py use: logger.experiment.add_image(s)"""

    def __init__(self):
        self.images = []
        self.experiment = self

    def add_images(self, tag, img, step):
        self.images.append((tag, tuple(img.shape), int(step)))

    def add_image(self, tag, img, step):
        self.images.append((tag, tuple(img.shape), int(step)))


class LoopDriver:
    """The call order of 1.5.2's fit loop for ONE optimiser under automatic optimisation (NOT a Trainer: no callbacks, no devices, no
    epochs).  Per batch: training_step inside the hook context -> optimizer.zero_grad() -> loss.backward() -> optimizer.step() ->
    global_step += 1 -> the step-interval scheduler's step() (through the module's lr_scheduler_step hook when it defines one, as 1.6+
    do; 1.5.2 calls scheduler.step() itself)."""

    def __init__(self, module: LightningModule):
        self.module, self.global_step, self.current_epoch = module, 0, 0
        self.logger, self.logged, self.progress_bar = _Logger(), {}, set()
        module.trainer = self
        cfg = module.configure_optimizers()
        optimizers, schedulers = cfg
        assert len(optimizers) == 1 and len(schedulers) == 1 and schedulers[0]["interval"] == "step"
        self.optimizer, self.scheduler = optimizers[0], schedulers[0]["scheduler"]

    def _hook(self, name, *args):
        m = self.module
        assert m._current_fx_name is None
        m._current_fx_name = name
        try:
            return getattr(m, name)(*args)
        finally:
            m._current_fx_name = None

    def fit_batch(self, batch, batch_idx):
        self.module.train()
        loss = self._hook("training_step", batch, batch_idx)
        if not (torch.is_tensor(loss) and loss.requires_grad and loss.numel() == 1):
            raise MisconfigurationException("In automatic optimization, `training_step` must return a Tensor (the loss, attached to the graph), "
                                            "a dict with key 'loss' or None")
        self.optimizer.zero_grad()
        loss.backward()
        self.optimizer.step()
        self.global_step += 1
        self.scheduler.step()
        return loss.detach()

    def validate(self, batches):
        self.module.eval()
        outs = []
        with torch.no_grad():
            for i, b in enumerate(batches):
                outs.append(self._hook("validation_step", b, i))
            self._hook("validation_epoch_end", outs)
        self.module.train()
        return outs

    def save_checkpoint(self, path):
        """trainer/connectors/checkpoint_connector.dump_checkpoint, the keys a 1.5.2 ModelCheckpoint file has"""
        m = self.module
        ckpt = {"epoch": self.current_epoch, "global_step": self.global_step, "pytorch-lightning_version": __version__,
                "state_dict": m.state_dict(), "callbacks": {}, "optimizer_states": [self.optimizer.state_dict()],
                "lr_schedulers": [self.scheduler.state_dict()]}
        if m.hparams:
            if hasattr(m, "_hparams_name"):
                ckpt[m.CHECKPOINT_HYPER_PARAMS_NAME] = m._hparams_name
            ckpt[m.CHECKPOINT_HYPER_PARAMS_KEY] = dict(m.hparams)
        torch.save(ckpt, path)
        return ckpt


def install():
    """sys.modules['pytorch_lightning'] = this stand-in; returns the previous entry (or None)"""
    mod = types.ModuleType("pytorch_lightning")
    mod.LightningModule = LightningModule
    mod.__version__ = __version__
    prev = sys.modules.get("pytorch_lightning")
    sys.modules["pytorch_lightning"] = mod
    return prev


def system_module_under_lightning():
    """mipnerf_pl_amd/system.py imported AS IF pytorch_lightning were installed, under its own module name (the regular
    mipnerf_pl_amd.system stays untouched): its `_HAVE_PL` branch executes and MipNeRFSystem derives from the strict LightningModule."""
    import importlib.util
    import os
    import mipnerf_pl_amd
    name = "mipnerf_pl_amd._system_under_lightning"
    if name in sys.modules:
        return sys.modules[name]
    prev = install()
    try:
        path = os.path.join(os.path.dirname(mipnerf_pl_amd.__file__), "system.py")
        spec = importlib.util.spec_from_file_location(name, path)
        mod = importlib.util.module_from_spec(spec)
        mod.__package__ = "mipnerf_pl_amd"
        sys.modules[name] = mod
        spec.loader.exec_module(mod)
    finally:
        if prev is None:
            del sys.modules["pytorch_lightning"]
        else:
            sys.modules["pytorch_lightning"] = prev
    assert mod._HAVE_PL and issubclass(mod.MipNeRFSystem, LightningModule)
    return mod
