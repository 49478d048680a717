"""LR schedulers used by configure_optimizers (avssl/optim/scheduler.py:10-47): optimizer side, plain torch."""
import torch


def get_scheduler(name: str, optimizer: torch.optim.Optimizer, **kwargs):
    if name == "linear_warmup_decay":
        warmup, max_step = kwargs["warmup"], kwargs["max_step"]
        init_lr, final_lr = kwargs.get("init_lr", 0.0), kwargs.get("final_lr", 0.0)
        base = optimizer.param_groups[0]["lr"]

        def fn(step):
            if step < warmup:
                lr = init_lr + (base - init_lr) * step / max(1, warmup)
            else:
                lr = base + (final_lr - base) * min(1.0, (step - warmup) / max(1, max_step - warmup))
            return lr / base

        return torch.optim.lr_scheduler.LambdaLR(optimizer, fn)
    if name == "noam":
        warmup = kwargs["warmup"]
        return torch.optim.lr_scheduler.LambdaLR(optimizer, lambda s: min((s + 1) ** -0.5, (s + 1) * warmup ** -1.5) * warmup ** 0.5)
    raise NotImplementedError(name)
