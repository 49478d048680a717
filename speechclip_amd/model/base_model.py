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
            ckpt = torch.load(path, map_location=map_location, weights_only=False)
            config = ckpt["hyper_parameters"]["config"]
            model = cls(config)
            sd = {k: v for k, v in ckpt["state_dict"].items() if not k.startswith("cascaded_branch.clip.")}  # duplicate of clip.*
            model.load_state_dict(sd, strict=strict)
            return model


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
