"""LR schedulers used by configure_optimizers (avssl/optim/scheduler.py:10-47): optimizer side, plain torch."""
import torch


def get_scheduler(name: str, optimizer: torch.optim.Optimizer, **kwargs):
    if name == "linear_warmup_decay":      # avssl/optim/scheduler.py:22-38, same multiplier step for step
        warmup, max_step = kwargs.get("warmup", 4000), kwargs.get("max_step", 1000000)
        final_lr_rate = kwargs.get("final_lr", 1e-8) / optimizer.param_groups[0]["lr"]

        def fn(step):
            if step < warmup:
                return (step + 1) / warmup
            return 1.0 - (1.0 - final_lr_rate) * (step + 1 - warmup) / (max_step - warmup)

        return torch.optim.lr_scheduler.LambdaLR(optimizer, fn)
    if name == "noam":
        warmup = kwargs.get("warmup", 4000)     # avssl/optim/scheduler.py:10-19
        return torch.optim.lr_scheduler.LambdaLR(optimizer, lambda s: (s + 1) / warmup if s < warmup else (warmup / (s + 1)) ** 0.5)
    raise NotImplementedError(name)
