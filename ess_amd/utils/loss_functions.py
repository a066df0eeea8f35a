"""Losses of the ESS trainers as fused HIP kernels (reference: utils/loss_functions.py).

Each loss is one forward kernel pair that also emits the gradient w.r.t. its first argument; the class
surface (TaskLoss, symJSDivLoss, DiceLoss, L1) is the reference's."""
import torch

from .. import functional as Fn


class TaskLoss(torch.nn.Module):
    """Dice + cross-entropy (reference :6-24); gamma / alpha / weight / reduction are accepted and, as in the
    reference, unused."""

    def __init__(self, losses=['cross_entropy'], gamma=2.0, num_classes=13, alpha=None, weight=None, ignore_index=None,
                 reduction='mean'):
        super().__init__()
        self.losses = losses
        self.weight, self.gamma, self.alpha = weight, gamma, alpha
        self.num_classes = num_classes
        self.ignore_index = ignore_index

    def forward(self, predict, target, weight=1.0):
        """weight: the trainer's loss weight, folded into the kernel (loss and gradient come out scaled)."""
        assert predict.shape[1] == self.num_classes, 'predict & target shape do not match'
        ign = -1 if self.ignore_index is None else self.ignore_index
        return Fn.task_loss(predict, target, ign, 'dice' in self.losses, 'cross_entropy' in self.losses, weight)


class DiceLoss(torch.nn.Module):
    """Multi-class dice over the whole batch, smooth=1, p=2, mean over classes (reference :96-135)."""

    def __init__(self, weight=None, num_classes=13, ignore_index=None, **kwargs):
        super().__init__()
        if weight is not None or kwargs:
            raise NotImplementedError('DiceLoss class weights / non-default smooth,p are not used by ESS')
        self.num_classes, self.ignore_index = num_classes, ignore_index

    def forward(self, predict, target):
        ign = -1 if self.ignore_index is None else self.ignore_index
        return Fn.task_loss(predict, target, ign, True, False)


class symJSDivLoss(torch.nn.Module):
    """0.5*KL(p||q) + 0.5*KL(q||p) with element-mean reduction and 1e-10 clamps (reference :27-37).
    Gradient flows to `predict`; `target` is a no-grad prediction in every call site."""

    def forward(self, predict, target, weight=1.0):
        return Fn.sym_js_div(predict, target.detach(), weight)


class L1Loss(torch.nn.Module):
    """torch.nn.L1Loss() as used for the cycle losses (training/ess_trainer.py:28,217-253)."""

    def forward(self, predict, target, weight=1.0):
        return Fn.l1_loss(predict, Fn.detach_keep_c8(target), weight)
