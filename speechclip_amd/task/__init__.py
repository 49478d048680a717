"""Task entry point kept for `run_task.py <TaskName>` compatibility (avssl/task/train_KWClip.py:71-76,
avssl/task/base_task.py:17-245): `TrainKWClip_GeneralTransformer().add_args / parse_args / run`.

The reference's run() builds Flickr8k/SpokenCOCO datasets, DataLoaders, ModelCheckpoint callbacks and a Lightning Trainer --
the control plane, which is out of scope for this build (data layer / Trainer are OUT OF SCOPE; no datasets or
Lightning in the image).  run() therefore constructs the model exactly as the reference does (seed, YAML -> OrderedNamespace([args,
yaml]) or load_from_checkpoint) and drives the Lightning hook protocol over caller-supplied batches."""
import argparse

import torch
import yaml

from ..base import OrderedNamespace
from ..model import KWClip_GeneralTransformer
from ..util import add_general_arguments


class BaseTask:
    def __init__(self):
        self.args = None
        self.config = None

    def add_args(self, parser: argparse.ArgumentParser) -> argparse.ArgumentParser:
        return add_general_arguments(parser)

    def parse_args(self, parser: argparse.ArgumentParser, argv=None):
        self.args = parser.parse_args(argv)
        return self.args

    def build_model(self, model_cls):
        assert self.args is not None
        torch.manual_seed(self.args.seed)
        if self.args.resume != "":
            self.args.ckpt = self.args.resume
        if self.args.ckpt:
            model = model_cls.load_from_checkpoint(self.args.ckpt)
            cfg = model.config.to_dict()
            cfg.update(vars(self.args))
            model.config = OrderedNamespace(cfg)
        else:
            cfg = yaml.load(open(self.args.config, "r"), Loader=yaml.FullLoader)
            model = model_cls(OrderedNamespace([self.args, cfg]))
        self.config = model.config
        return model


class TrainKWClip_GeneralTransformer(BaseTask):
    @staticmethod
    def fit(model, train_batches, max_steps=None, log_every=0):
        """Lightning's training loop for the trainable tail, hook for hook: training_step -> training_step_end -> backward ->
        (gradient clipping inside the optimizer step) -> optimizer.step -> scheduler.step (interval: step).  Returns the loss history."""
        model = model.cuda().train()
        (opt,), (sch,) = model.configure_optimizers()
        hist = []
        for i, b in enumerate(train_batches):
            if max_steps is not None and i >= max_steps:
                break
            b = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in b.items()}
            opt.zero_grad()
            loss = model.training_step_end(model.training_step(b, i))["loss"]
            loss.backward()
            opt.step()
            sch["scheduler"].step()
            hist.append(float(loss))
            if log_every and i % log_every == 0:
                print(f"step {i}: loss {hist[-1]:.4f} lr {opt.param_groups[0]['lr']:.3e}")
        return hist

    def run(self, batches=None, train_batches=None):
        """Build the model; `train_batches` (iterable of collate_general-style dicts): train the tail over them (fit);
        `batches`: run validation_step / validation_step_end / validation_epoch_end over them exactly in Lightning's order and
        return the recalls."""
        model = self.build_model(KWClip_GeneralTransformer)
        if train_batches is not None:
            max_steps = self.config.trainer.get("max_steps", None) if hasattr(self.config, "trainer") else None
            self.loss_history = self.fit(model, train_batches, max_steps=max_steps)
        if batches is None:
            return model
        model = model.cuda().eval()
        outs = []
        with torch.no_grad():
            for i, b in enumerate(batches):
                outs.append(model.validation_step_end(model.validation_step({k: (v.cuda() if torch.is_tensor(v) else v) for k, v in b.items()}, i)))
            return model.validation_epoch_end(outs)
