"""Semantic-segmentation metrics with an on-device confusion matrix (reference: evaluation/metrics.py)."""
import torch

from .. import hip


def semseg_compute_confusion(y_hat_lbl, y_lbl, num_classes, ignore_label):
    """conf[label, prediction] as int64 (reference :4-24), via one histogram kernel over the two label maps instead of a masked
    bincount (no one-hot tensor: `ess_label_confusion`)."""
    assert torch.is_tensor(y_hat_lbl) and torch.is_tensor(y_lbl), 'Inputs must be torch tensors'
    assert y_lbl.device == y_hat_lbl.device, 'Input tensors have different device placement'
    if y_hat_lbl.dim() == 4:
        y_hat_lbl = y_hat_lbl.squeeze(1)
    if y_lbl.dim() == 4:
        y_lbl = y_lbl.squeeze(1)
    conf = torch.zeros(num_classes, num_classes, dtype=torch.int64, device=y_lbl.device)
    hip.label_confusion(y_hat_lbl.long().contiguous(), y_lbl.long().contiguous(), conf, ignore_label)
    return conf


def logits_to_confusion(logits, y_lbl, num_classes, ignore_label, conf=None):
    """argmax(dim=1) + confusion accumulation in one pass (training/ess_trainer.py:485-492 fused)."""
    if conf is None:
        conf = torch.zeros(num_classes, num_classes, dtype=torch.int64, device=logits.device)
    pred = hip.argmax_confusion(logits.contiguous(), y_lbl.long().contiguous(), conf, ignore_label)
    return pred, conf


def semseg_accum_confusion_to_iou(confusion_accum):
    conf = confusion_accum.double()
    diag = conf.diag()
    iou_per_class = 100 * diag / (conf.sum(dim=1) + conf.sum(dim=0) - diag).clamp(min=1e-12)
    return iou_per_class.mean(), iou_per_class


def semseg_accum_confusion_to_acc(confusion_accum):
    conf = confusion_accum.double()
    return 100 * conf.diag().sum() / conf.sum().clamp(min=1e-12)


class MetricsSemseg:
    def __init__(self, num_classes, ignore_label, class_names):
        self.num_classes, self.ignore_label, self.class_names = num_classes, ignore_label, class_names
        self.metrics_acc = None

    def reset(self):
        self.metrics_acc = None

    def update_batch(self, y_hat_lbl, y_lbl):
        with torch.no_grad():
            batch = semseg_compute_confusion(y_hat_lbl, y_lbl, self.num_classes, self.ignore_label).cpu()
            self.metrics_acc = batch if self.metrics_acc is None else self.metrics_acc.cpu() + batch

    def update_batch_logits(self, logits, y_lbl):
        """argmax + confusion in one kernel, accumulated into a device-resident matrix: no host read per batch (the
        reference's update_batch moves every batch's matrix to the CPU, evaluation/metrics.py:50-57)."""
        with torch.no_grad():
            if self.metrics_acc is None or self.metrics_acc.device != logits.device:
                prev = self.metrics_acc
                self.metrics_acc = torch.zeros(self.num_classes, self.num_classes, dtype=torch.int64, device=logits.device)
                if prev is not None:
                    self.metrics_acc += prev.to(logits.device)
            pred, _ = logits_to_confusion(logits, y_lbl, self.num_classes, self.ignore_label, conf=self.metrics_acc)
        return pred

    def get_metrics_summary(self):
        from ..training import distributed as D
        D.all_reduce_sum_([self.metrics_acc])  # data parallel: every rank validated its own shard; the matrix is the sum
        self.metrics_acc = self.metrics_acc.cpu()  # the one host read; the reference keeps the accumulator on the CPU
        iou_mean, iou_per_class = semseg_accum_confusion_to_iou(self.metrics_acc)
        out = {self.class_names[i]: iou for i, iou in enumerate(iou_per_class)}
        out['mean_iou'] = iou_mean
        out['acc'] = semseg_accum_confusion_to_acc(self.metrics_acc)
        out['cm'] = self.metrics_acc
        return out
