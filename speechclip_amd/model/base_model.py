"""BaseLightningModel (avssl/model/base_model.py:10-26).  Subclasses pytorch_lightning.LightningModule when Lightning is
installed (so `Trainer.fit/validate` drive the hooks exactly as in the reference); otherwise a minimal stand-in with the
members the model layer uses (`save_hyperparameters`, `log`, `log_dict`, `device`, `load_from_checkpoint`)."""
import torch
from torch import nn

from ..base import OrderedNamespace

try:  # pragma: no cover - Lightning is not in the build image
    import pytorch_lightning as pl
    _Base = pl.LightningModule
    HAVE_LIGHTNING = True
except Exception:  # noqa: BLE001
    HAVE_LIGHTNING = False

    class _Base(nn.Module):
        logger = None
        global_step = 0

        def save_hyperparameters(self, *args, **kwargs):
            self.hparams = {"config": getattr(self, "config", None)}

        def log(self, name, value, **kwargs):
            self._logged = getattr(self, "_logged", {})
            self._logged[name] = value

        def log_dict(self, d, **kwargs):
            for k, v in d.items():
                self.log(k, v)

        @property
        def device(self):
            try:
                return next(self.parameters()).device
            except StopIteration:
                return torch.device("cpu")

        @classmethod
        def load_from_checkpoint(cls, path, map_location="cpu", strict=True, **kwargs):
            # Lightning checkpoints pickle pytorch_lightning classes (`callbacks` is keyed by the ModelCheckpoint CLASS, base_task.py:176-193) next to
            # the avssl OrderedNamespace in `hyper_parameters`: read with the restricted unpickler (inert stubs for what is not importable here)
            from ..util.checkpoint_io import load_pickled_checkpoint
            ckpt, _ = load_pickled_checkpoint(path, map_location=map_location)
            config = ckpt["hyper_parameters"]["config"]
            if not isinstance(config, OrderedNamespace):
                raise TypeError(f"{path}: hyper_parameters['config'] is {type(config).__name__}, expected the avssl OrderedNamespace (base_model.py:14)")
            model = cls(config)
            load_checkpoint_state(model, ckpt["state_dict"], strict=strict)
            return model


def load_checkpoint_state(model: nn.Module, state_dict: dict, strict: bool = True):
    """Load a reference checkpoint's `state_dict` into `model`.

    Reference checkpoints carry the shared CLIP tower twice -- `clip.*` and `cascaded_branch.clip.*`, because KW_CascadedBranch registers
    the shared ClipModel as a sub-module (kwClip.py:720) -- and so does this model's own state_dict.  Both spellings are accepted: whichever
    of the two is absent from the checkpoint is aliased from the other, nothing is dropped.  Keys that are derived state rather than weights
    (the loss's identity-matrix buffers, losses.py:126) may be missing or extra; anything else missing / unexpected raises under `strict`."""
    sd = dict(state_dict)
    dup, own = "cascaded_branch.clip.", "clip."
    want = model.state_dict().keys()
    for k in want:
        if k in sd:
            continue
        if k.startswith(dup) and own + k[len(dup):] in sd:
            sd[k] = sd[own + k[len(dup):]]
        elif k.startswith(own) and dup + k[len(own):] in sd:
            sd[k] = sd[dup + k[len(own):]]
    if not any(k.startswith(dup) for k in want):          # parallel-only model fed a checkpoint that has the duplicate tower
        sd = {k: v for k, v in sd.items() if not k.startswith(dup)}
    missing, unexpected = model.load_state_dict(sd, strict=False)
    derived = ("criterion.eye_mat", "criterion.neg_eye_mat")
    bad_missing = [k for k in missing if not k.startswith(derived)]
    bad_unexpected = [k for k in unexpected if not k.startswith(derived)]
    if strict and (bad_missing or bad_unexpected):
        raise RuntimeError(f"load_checkpoint_state: missing keys {bad_missing}, unexpected keys {bad_unexpected}")
    return bad_missing, bad_unexpected


class BaseLightningModel(_Base):
    def __init__(self, config: OrderedNamespace):
        super().__init__()
        self.config = config
        self.save_hyperparameters()

    def forward(self, batch):
        raise NotImplementedError

    def training_step(self, batch, batch_idx):
        raise NotImplementedError

    def configure_optimizers(self):
        raise NotImplementedError
