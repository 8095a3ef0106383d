"""Training helpers with the reference's names and behaviour (/root/reference/binary_seg/utils/utils.py)."""
import torch


def clip_gradient(optimizer, grad_clip):
    """Per-element clamp of every gradient to [-grad_clip, grad_clip] (reference utils/utils.py:7-17)."""
    try:          # gradients handed out by the module surface are views of ONE flat buffer: one clamp launch, and Adam.step as one kernel (pn2/optim.py)
        from pn2 import optim as _po
        if _po.clip_flat(optimizer, grad_clip):
            _po.fuse_adam(optimizer)
            return
    except ImportError:
        pass
    grads = [param.grad.data for group in optimizer.param_groups for param in group['params'] if param.grad is not None]
    dense = [g for g in grads if g.is_cuda and not g.is_sparse]
    if len(dense) == len(grads) and grads:
        # the same in-place clamp of every tensor, as two multi-tensor launches instead of one launch per parameter (~480 for PraNet-V2)
        torch._foreach_clamp_min_(grads, -grad_clip)
        torch._foreach_clamp_max_(grads, grad_clip)
        return
    for g in grads:
        g.clamp_(-grad_clip, grad_clip)


def adjust_lr(optimizer, init_lr, epoch, decay_rate=0.1, decay_epoch=30):
    """Multiplies the CURRENT lr (so decay compounds), exactly as reference utils/utils.py:20-23."""
    decay = decay_rate ** (epoch // decay_epoch)
    for param_group in optimizer.param_groups:
        param_group['lr'] *= decay


class AvgMeter(object):
    """Loss meter of the training loop (same surface as reference utils/utils.py:26-46: update(val, n), show(), reset(); val / avg / sum /
    count attributes).  show() is the mean of the most recent `num` values, as a tensor."""

    def __init__(self, num=40):
        self.num = num
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0
        self.losses = []

    def update(self, val, n=1):
        self.val, self.sum, self.count = val, self.sum + val * n, self.count + n
        self.avg = self.sum / self.count
        self.losses.append(val)

    def show(self):
        recent = self.losses[-self.num:] if self.num > 0 else self.losses[len(self.losses):]
        return torch.stack(list(recent)).mean()
