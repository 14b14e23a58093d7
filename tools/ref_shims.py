"""sys.modules shims that let the reference's own classes be imported and executed on CPU in the BUILD container.

Only used by tools/gen_golden.py (and nothing that ships or runs on the GPU box).  The reference is never copied:
it is imported from /root/reference.  Missing third-party packages are replaced by minimal stand-ins that do not touch
the arithmetic of the hot path:
  * pytorch_lightning / hydra / omegaconf / xarray / tensordict / wandb / tensorly / tltorch : structural stubs
  * modulus.*                          : aliased to the copies the reference vendors under src/models/sfno/
  * torch_harmonics                    : the package is absent from the reference AND from this image; its
                                         RealSHT / InverseRealSHT are supplied by oracle/sht.py (restated from the
                                         published algorithm; "parity unpinned" at that boundary)
"""
import importlib
import inspect
import sys
import types

import torch
from torch import nn

REF_ROOT = "/root/reference"


class AttrDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def pop(self, k, *d):
        return dict.pop(self, k, *d)


def _wrap(obj):
    if isinstance(obj, dict) and not isinstance(obj, AttrDict):
        return AttrDict({k: _wrap(v) for k, v in obj.items()})
    return obj


def _module(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


class LightningModule(nn.Module):
    def __init__(self, *a, **k):
        super().__init__()
        self.__dict__["_hparams"] = AttrDict()
        self._trainer = None

    @property
    def hparams(self):
        return self.__dict__["_hparams"]

    def save_hyperparameters(self, *args, ignore=None, **kw):
        """Like Lightning: collect the __init__ arguments of EVERY class in the constructor chain that is on the stack
        (a base __init__ sees the arguments of the subclasses that called it)."""
        ignore = set([ignore] if isinstance(ignore, str) else (ignore or []))
        frame = inspect.currentframe().f_back
        while frame is not None:
            if frame.f_code.co_name == "__init__" and frame.f_locals.get("self") is self:
                info = inspect.getargvalues(frame)
                for name in info.args[1:]:
                    if name not in ignore and name not in self.hparams:
                        self.hparams[name] = info.locals[name]
                if info.keywords and info.keywords in info.locals:
                    for k, v in info.locals[info.keywords].items():
                        if k not in ignore and k not in self.hparams:
                            self.hparams[k] = v
            frame = frame.f_back

    @property
    def device(self):
        try:
            return next(self.parameters()).device
        except StopIteration:
            return torch.device("cpu")

    @property
    def trainer(self):
        return self._trainer

    def log(self, *a, **k):
        pass

    def log_dict(self, *a, **k):
        pass


def _instantiate(cfg, *args, _recursive_=True, **kwargs):
    cfg = dict(cfg)
    target = cfg.pop("_target_")
    cfg.pop("_recursive_", None)
    mod, _, cls = target.rpartition(".")
    klass = getattr(importlib.import_module(mod), cls)
    cfg.update(kwargs)
    return klass(*args, **cfg)


def install():
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    repo = __file__.rsplit("/tools/", 1)[0]
    if repo not in sys.path:
        sys.path.insert(0, repo)

    # ---- pytorch_lightning -------------------------------------------------------------------------------------
    class _Anything:
        def __init__(self, *a, **k):
            pass

    pl = _module("pytorch_lightning", LightningModule=LightningModule, LightningDataModule=_Anything, Trainer=_Anything,
                 seed_everything=lambda *a, **k: None)
    _module("pytorch_lightning.utilities", rank_zero_only=lambda f: f)
    _module("pytorch_lightning.utilities.types", EVAL_DATALOADERS=object, TRAIN_DATALOADERS=object)
    _module("pytorch_lightning.callbacks", Callback=_Anything, ModelCheckpoint=_Anything)
    _module("pytorch_lightning.loggers", WandbLogger=_Anything)
    _module("pytorch_lightning.loggers.wandb", WandbLogger=_Anything)
    pl.utilities = sys.modules["pytorch_lightning.utilities"]
    pl.callbacks = sys.modules["pytorch_lightning.callbacks"]
    pl.loggers = sys.modules["pytorch_lightning.loggers"]

    # ---- omegaconf / hydra ------------------------------------------------------------------------------------------
    class OmegaConf:
        @staticmethod
        def create(d=None):
            return _wrap(d or {})

        @staticmethod
        def to_container(c, **k):
            return dict(c)

        @staticmethod
        def is_config(c):
            return isinstance(c, AttrDict)

    _module("omegaconf", DictConfig=AttrDict, OmegaConf=OmegaConf, ListConfig=list, open_dict=None)
    hyd = _module("hydra")
    hyd.utils = _module("hydra.utils", instantiate=_instantiate)

    # ---- data / logging stubs -------------------------------------------------------------------------------------------
    _module("xarray", DataArray=_Anything, Dataset=_Anything)

    class TensorDict(dict):
        def __init__(self, d=None, batch_size=None, **k):
            super().__init__(d or {})

        def to(self, *a, **k):
            return TensorDict({n: (v.to(*a, **k) if hasattr(v, "to") else v) for n, v in self.items()})

    _module("tensordict", TensorDict=TensorDict, TensorDictBase=TensorDict)
    _module("wandb", run=None, Table=_Anything, Image=_Anything, Video=_Anything, Histogram=_Anything, Object3D=_Anything,
            Audio=_Anything, Html=_Anything, plot=_Anything)
    # imported (not used on the sampling path) by src/ace_inference/core/stepper_multistep.py and friends
    def _curry(f=None, *a, **k):   # toolz.curry as used by derived_variables.py: `@register()` with no arguments
        if f is None or not callable(f):
            return lambda g: _curry(g)
        import functools

        @functools.wraps(f)
        def wrapper(*args, **kwargs):
            if not args and not kwargs:
                return wrapper
            return f(*args, **kwargs)
        return wrapper

    _module("toolz", curry=_curry)
    _module("dacite", from_dict=lambda *a, **k: None, Config=_Anything)
    _module("netCDF4", Dataset=_Anything)
    _module("h5py", File=_Anything)
    _module("tensorly", set_backend=lambda *a, **k: None, ndim=lambda x: x.ndim, einsum=torch.einsum)
    _module("tltorch")
    _module("tltorch.factorized_tensors")
    _module("tltorch.factorized_tensors.core", FactorizedTensor=type("FactorizedTensor", (), {}))

    # ---- torch_harmonics: restated (oracle/sht.py) -------------------------------------------------------------------------
    from oracle import sht as osht

    class _Dummy(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

    th = _module("torch_harmonics", RealSHT=osht.RealSHT, InverseRealSHT=osht.InverseRealSHT, RealFFT2=_Dummy,
                 InverseRealFFT2=_Dummy)
    th.__all__ = ["RealSHT", "InverseRealSHT"]
    thd = _module("torch_harmonics.distributed", DistributedRealSHT=type("DistributedRealSHT", (_Dummy,), {}),
                  DistributedInverseRealSHT=type("DistributedInverseRealSHT", (_Dummy,), {}), init=lambda *a, **k: None)
    th.distributed = thd

    # ---- modulus -> vendored copies ---------------------------------------------------------------------------------------
    for n in ("modulus", "modulus.models", "modulus.models.sfno", "modulus.utils", "modulus.utils.sfno",
              "modulus.utils.sfno.distributed"):
        _module(n)
    _module("modulus.utils.sfno.logging_utils", disable_logging=lambda *a, **k: (lambda f: f))
    for n in ("activations", "contractions", "initialization"):
        sys.modules[f"modulus.models.sfno.{n}"] = importlib.import_module(f"src.models.sfno.{n}")
    sys.modules["modulus.models.sfno.factorizations"] = importlib.import_module("src.models.sfno.factorizations")
    for n in ("comm", "helpers", "mappings"):
        try:
            sys.modules[f"modulus.utils.sfno.distributed.{n}"] = importlib.import_module(f"src.models.sfno.distributed.{n}")
        except Exception:  # noqa: BLE001
            pass
