"""``YOGOLoss`` -- the reference's loss surface (yogo/yogo_loss.py:8-129) on one fused HIP kernel.

``forward(pred, label) -> (loss, {"iou_loss", "objectness_loss", "classification_loss"})``: the scalar carries the
autograd edge (its backward is the gradient the same kernel already produced), the components are read back with a
single 16-byte device->host copy on first access instead of the reference's three ``.item()`` syncs.
"""
from __future__ import annotations

from typing import Dict, Iterator, Tuple

import torch

from yogo_amd import _hip


class LossComponents(dict):
    """dict of floats materialised lazily from the 4-float device result (one sync, only if somebody looks)."""

    _KEYS = ("iou_loss", "objectness_loss", "classification_loss")

    def __init__(self, dev: torch.Tensor):
        super().__init__()
        self._dev = dev
        self._done = False

    def _fill(self) -> None:
        if not self._done:
            vals = self._dev.detach().cpu().tolist()
            for k, v in zip(self._KEYS, vals[1:]):
                dict.__setitem__(self, k, float(v))
            self._done = True

    def __getitem__(self, k):
        self._fill()
        return dict.__getitem__(self, k)

    def __iter__(self) -> Iterator:
        self._fill()
        return dict.__iter__(self)

    def __len__(self) -> int:
        return 3

    def __contains__(self, k) -> bool:
        return k in self._KEYS

    def keys(self):
        self._fill()
        return dict.keys(self)

    def items(self):
        self._fill()
        return dict.items(self)

    def values(self):
        self._fill()
        return dict.values(self)

    def get(self, k, default=None):
        self._fill()
        return dict.get(self, k, default)

    def __repr__(self) -> str:
        self._fill()
        return dict.__repr__(self)


class _LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, label, now, iw, cw, ls):  # type: ignore[override]
        B, P, Sy, Sx = pred.shape
        pred_c = pred.detach().contiguous().float()
        label_c = label.detach().contiguous().float()
        grad = torch.empty_like(pred_c)
        out = torch.empty(4, dtype=torch.float32, device=pred.device)
        with torch.cuda.device(pred.device):
            ws = torch.empty(_hip.query_size("yogo_loss_workspace_bytes", B, Sy, Sx) // 4, dtype=torch.float32, device=pred.device)
            _hip.call("yogo_loss_fwd_bwd", pred_c, label_c, grad, out, ws, B, P, Sy, Sx, now, iw, cw, ls, _hip.stream_ptr())
        ctx.save_for_backward(grad)
        ctx.mark_non_differentiable(out)
        return out[0], out

    @staticmethod
    def backward(ctx, gloss, _gout):  # type: ignore[override]
        (grad,) = ctx.saved_tensors
        return grad * gloss, None, None, None, None, None


class YOGOLoss(torch.nn.modules.loss._Loss):
    __constants__ = ["no_obj_weight", "iou_weight", "classify_weight"]

    def __init__(self, no_obj_weight: float = 0.5, iou_weight: float = 5.0, classify_weight: float = 1.0,
                 label_smoothing: float = 0.01) -> None:
        super().__init__()
        self.no_obj_weight = no_obj_weight
        self.iou_weight = iou_weight
        self.classify_weight = classify_weight
        self.label_smoothing = label_smoothing
        self.device = "cpu"

    def to(self, device):
        self.device = device
        super().to(device, non_blocking=True, dtype=torch.float32)
        return self

    def forward(self, pred_batch: torch.Tensor, label_batch: torch.Tensor) -> Tuple[torch.Tensor, Dict[str, float]]:
        """pred_batch [B, 5+C, Sy, Sx] (decoded boxes, objectness, raw class logits); label_batch [B, 6, Sy, Sx]
        (mask, x1, y1, x2, y2, class).  Returns (loss, components)."""
        _hip.require_cuda(pred_batch, "pred_batch")
        _hip.require_cuda(label_batch, "label_batch")
        if pred_batch.ndim != 4 or label_batch.ndim != 4 or label_batch.shape[1] != 6 or pred_batch.shape[1] <= 5:
            raise ValueError(f"expected pred [B,5+C,Sy,Sx] and label [B,6,Sy,Sx], got {tuple(pred_batch.shape)}, {tuple(label_batch.shape)}")
        if pred_batch.shape[0] != label_batch.shape[0] or pred_batch.shape[2:] != label_batch.shape[2:]:
            raise ValueError("pred and label batch/grid sizes differ")
        loss, out = _LossFn.apply(pred_batch, label_batch, float(self.no_obj_weight), float(self.iou_weight),
                                  float(self.classify_weight), float(self.label_smoothing))
        return loss, LossComponents(out)
