"""WarmupMultiStepLR (mirror of maskrcnn_benchmark/solver/lr_scheduler.py:10-52):
lr = initial_lr * warmup(iter) * gamma ** (number of milestones <= iter)."""
from bisect import bisect_right


class WarmupMultiStepLR(object):
    def __init__(self, optimizer, milestones, gamma=0.1, warmup_factor=1.0 / 3, warmup_iters=500, warmup_method="linear",
                 last_epoch=-1):
        if list(milestones) != sorted(milestones):
            raise ValueError("Milestones should be a list of increasing integers. Got {}".format(milestones))
        if warmup_method not in ("constant", "linear"):
            raise ValueError("Only 'constant' or 'linear' warmup_method accepted, got {}".format(warmup_method))
        self.optimizer, self.milestones, self.gamma = optimizer, list(milestones), gamma
        self.warmup_factor, self.warmup_iters, self.warmup_method = warmup_factor, warmup_iters, warmup_method
        self.base_lrs = [g["initial_lr"] for g in optimizer.param_groups]
        self.last_epoch = last_epoch
        self._step_count = 0          # torch's _LRScheduler bookkeeping: carried so that state_dict() round-trips with the reference's
        self._last_lr = list(self.base_lrs)
        self.step()

    def get_lr(self):
        warmup_factor = 1
        if self.last_epoch < self.warmup_iters:
            if self.warmup_method == "constant":
                warmup_factor = self.warmup_factor
            else:
                alpha = float(self.last_epoch) / self.warmup_iters
                warmup_factor = self.warmup_factor * (1 - alpha) + alpha
        k = self.gamma ** bisect_right(self.milestones, self.last_epoch)
        return [base_lr * warmup_factor * k for base_lr in self.base_lrs]

    def step(self):
        self.last_epoch += 1
        self._step_count += 1
        self._last_lr = self.get_lr()
        for g, lr in zip(self.optimizer.param_groups, self._last_lr):
            g["lr"] = lr

    def get_last_lr(self):
        return self._last_lr

    def state_dict(self):
        """torch's _LRScheduler.state_dict(): every attribute but the optimizer (what the reference checkpoints hold)."""
        return {k: v for k, v in self.__dict__.items() if k != "optimizer"}

    def load_state_dict(self, sd):
        for k in ("milestones", "gamma", "warmup_factor", "warmup_iters", "warmup_method", "base_lrs", "last_epoch", "_step_count", "_last_lr"):
            if k in sd:
                setattr(self, k, list(sd[k]) if k in ("milestones", "base_lrs", "_last_lr") else sd[k])
