"""CPU oracle for the YOGO hot path -- TEST INFRASTRUCTURE ONLY.

This file is the checker, never the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it.  Nothing under ``yogo_amd/`` imports it and the product path has no CPU
fallback.

It restates, with plain ``torch`` CPU ops / numpy, the algorithm of the
reference (czbiohub-sf/yogo @ 2024_08_07, paths relative to the reference
root):

* backbone + head                      yogo/model_defns.py:30-529, called at yogo/model.py:275
* box decode                           yogo/model.py:277-313
* Kaiming init + grad clamp            yogo/model.py:76-87
* loss                                 yogo/yogo_loss.py:38-129
* threshold + NMS post-process         yogo/utils/prediction_formatting.py:23-93
* inference output rows / arrays       yogo/infer.py:39-124, yogo/utils/prediction_formatting.py:96-156
* label rasteriser (synthetic labels)  yogo/data/yogo_dataset.py:24-46
* AdamW + cosine LR (trainer glue)     yogo/train.py:206-223, 324-325

Third-party arithmetic that is NOT in the reference tree (``torchvision.ops``,
pinned by the reference as ``torchvision>=0.14.1``, pyproject.toml:17) is
restated here from torchvision's published algorithm (0.15/0.16):
``box_convert``, ``complete_box_iou_loss``, ``nms`` (CPU kernel), ``box_iou``.

Pinning status (see DESIGN.md "Oracle"):
* backbone / decode / init / grid / checkpoint keys: pinned against the real
  reference, imported in the build container (tests/golden/make_golden.py).
* loss and format_preds control flow: pinned against the reference's own
  ``yogo_loss.py`` / ``prediction_formatting.py`` run on top of this file's
  torchvision restatement, and against the reference's 4 + 4 known-answer
  tests (tests/test_utils_tensor_formatting.py, tests/test_count_predictions.py).
* the torchvision arithmetic itself (CIoU value, NMS with real suppression):
  PARITY UNPINNED -- no torchvision is installable here and the reference's
  tests hold no vector for it.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

# ---------------------------------------------------------------------------
# architecture table (data read off yogo/model_defns.py:30-529 by introspection)
# entry = (cout, ksize, stride, has_bias, has_bn, act, dropout_p); 3x3 -> pad 1, 1x1 -> pad 0
# the last entry's cout is 5 + num_classes (written here as None)
# ---------------------------------------------------------------------------
def _std(w: Sequence[int], act: str) -> list:
    a, b, c, d = w
    return [
        (a, 3, 2, 0, 1, act, 0.0),
        (b, 3, 1, 1, 0, act, 0.05),
        (c, 3, 2, 1, 0, act, 0.1),
        (d, 3, 1, 1, 0, act, 0.15),
        (d, 3, 2, 0, 1, act, 0.0),
        (d, 3, 1, 1, 1, act, 0.0),
        (d, 3, 1, 1, 0, act, 0.0),
        (None, 1, 1, 1, 0, None, 0.0),
    ]


ARCH: Dict[str, list] = {
    "base_model": _std((16, 32, 64, 128), "leaky"),
    "silu_model": _std((16, 32, 64, 128), "silu"),
    "double_filters": _std((32, 64, 128, 256), "leaky"),
    "triple_filters": _std((48, 96, 192, 384), "leaky"),
    "half_filters": _std((8, 16, 32, 64), "leaky"),
    "quarter_filters": _std((4, 8, 16, 32), "leaky"),
    "depth_ver_0": [
        (32, 3, 2, 0, 1, "leaky", 0.0), (128, 3, 2, 1, 0, "leaky", 0.1),
        (128, 3, 2, 0, 1, "leaky", 0.0), (None, 1, 1, 1, 0, None, 0.0)],
    "depth_ver_1": [
        (16, 3, 2, 0, 1, "leaky", 0.0), (64, 3, 2, 1, 0, "leaky", 0.1),
        (128, 3, 1, 1, 0, "leaky", 0.15), (128, 3, 2, 0, 1, "leaky", 0.0),
        (128, 3, 1, 1, 0, "leaky", 0.0), (None, 1, 1, 1, 0, None, 0.0)],
    "depth_ver_2": _std((16, 32, 64, 128), "leaky"),
    "depth_ver_3": [
        (16, 3, 2, 0, 1, "leaky", 0.0), (32, 3, 1, 1, 0, "leaky", 0.05),
        (32, 3, 1, 1, 0, "leaky", 0.05), (64, 3, 2, 1, 0, "leaky", 0.1),
        (128, 3, 1, 1, 0, "leaky", 0.15), (128, 3, 1, 1, 1, "leaky", 0.0),
        (128, 3, 2, 0, 0, "leaky", 0.0), (128, 3, 1, 1, 1, "leaky", 0.0),
        (128, 3, 1, 1, 0, "leaky", 0.0), (None, 1, 1, 1, 0, None, 0.0)],
    "depth_ver_4": [
        (16, 3, 2, 0, 1, "leaky", 0.0), (16, 3, 1, 1, 0, "leaky", 0.0),
        (32, 3, 1, 1, 0, "leaky", 0.05), (32, 3, 1, 1, 0, "leaky", 0.05),
        (64, 3, 2, 1, 0, "leaky", 0.1), (64, 3, 1, 1, 0, "leaky", 0.0),
        (128, 3, 1, 1, 0, "leaky", 0.15), (128, 3, 1, 1, 1, "leaky", 0.0),
        (128, 3, 2, 1, 0, "leaky", 0.0), (128, 3, 1, 1, 1, "leaky", 0.0),
        (128, 3, 1, 1, 0, "leaky", 0.0), (None, 1, 1, 1, 0, None, 0.0)],
}

BN_EPS = 1e-5
BN_MOMENTUM = 0.1
LEAKY_SLOPE = 0.01


def arch(name: str, num_classes: int) -> list:
    return [((5 + num_classes) if e[0] is None else e[0],) + tuple(e[1:]) for e in ARCH[name]]


# ---------------------------------------------------------------------------
# grid maths -- yogo/model.py:189-234 (floor formula per conv) and :48-61 (linspace grids)
# ---------------------------------------------------------------------------
def grid_size(spec: list, h: int, w: int) -> Tuple[int, int]:
    """returns (Sx, Sy)"""
    for (_, k, s, *_r) in spec:
        p = 1 if k == 3 else 0
        h = (h + 2 * p - (k - 1) - 1) // s + 1
        w = (w + 2 * p - (k - 1) - 1) // s + 1
    return int(w), int(h)


def make_grids(Sx: int, Sy: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """_Cxs, _Cys exactly as yogo/model.py:48-55 (linspace, NOT k/S)."""
    cxs = torch.linspace(0, 1 - 1 / Sx, Sx).expand(Sy, -1).clone()
    cys = torch.linspace(0, 1 - 1 / Sy, Sy).expand(1, -1).transpose(0, 1).expand(Sy, Sx).clone()
    return cxs, cys


# ---------------------------------------------------------------------------
# parameter init -- yogo/model.py:79-87 (kaiming normal, a=0.01, fan_out, leaky_relu; zero bias)
# state-dict naming follows nn.Sequential of blocks: model.{i}.{j}.* ; bare head conv: model.{n-1}.*
# ---------------------------------------------------------------------------
def init_state(spec: list, in_ch: int = 1, seed: int = 0) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}
    cin = in_ch
    n = len(spec)
    for i, (co, k, s, hb, hbn, act, dp) in enumerate(spec):
        pre = f"model.{i}." if i == n - 1 and act is None and not hbn else f"model.{i}.0."
        std = math.sqrt(2.0 / (1 + 0.01 ** 2)) / math.sqrt(co * k * k)
        sd[pre + "weight"] = torch.randn(co, cin, k, k, generator=g) * std
        if hb:
            sd[pre + "bias"] = torch.zeros(co)
        if hbn:
            b = f"model.{i}.1."
            sd[b + "weight"] = torch.ones(co)
            sd[b + "bias"] = torch.zeros(co)
            sd[b + "running_mean"] = torch.zeros(co)
            sd[b + "running_var"] = torch.ones(co)
            sd[b + "num_batches_tracked"] = torch.tensor(0, dtype=torch.long)
        cin = co
    return sd


def conv_prefix(spec: list, i: int) -> str:
    co, k, s, hb, hbn, act, dp = spec[i]
    bare = (i == len(spec) - 1) and act is None and not hbn
    return f"model.{i}." if bare else f"model.{i}.0."


# ---------------------------------------------------------------------------
# backbone -- yogo/model_defns.py block grammar: Conv2d, [BatchNorm2d], [LeakyReLU|SiLU], [Dropout2d]
# ---------------------------------------------------------------------------
def _act(x: torch.Tensor, act: Optional[str]) -> torch.Tensor:
    if act == "leaky":
        return F.leaky_relu(x, LEAKY_SLOPE)
    if act == "silu":
        return F.silu(x)
    return x


def backbone_forward(
    x: torch.Tensor,
    sd: Dict[str, torch.Tensor],
    spec: list,
    train: bool = False,
    drop_masks: Optional[Dict[int, torch.Tensor]] = None,
    new_stats: Optional[Dict[str, torch.Tensor]] = None,
    taps: Optional[Dict[int, torch.Tensor]] = None,
) -> torch.Tensor:
    """x: float [B,Cin,H,W] (the uint8->float cast of yogo/model.py:272-273 is the caller's).

    train=True: BatchNorm uses biased batch statistics and (if ``new_stats`` is a dict) reports the
    momentum-0.1 running-stat update with the unbiased variance, as torch.nn.BatchNorm2d does.
    Dropout2d cannot be bit-matched to torch's RNG: ``drop_masks[i]`` is an injected, already
    scaled [B,C] channel mask (1/(1-p) or 0) for block i, identity when absent.
    ``taps[i]`` receives the output of block i (for per-layer checks).
    """
    for i, (co, k, s, hb, hbn, act, dp) in enumerate(spec):
        pre = conv_prefix(spec, i)
        x = F.conv2d(x, sd[pre + "weight"], sd.get(pre + "bias") if hb else None, stride=s, padding=1 if k == 3 else 0)
        if hbn:
            b = f"model.{i}.1."
            # torch.nn.BatchNorm2d semantics (biased batch var to normalise, unbiased var into the
            # running stats, momentum 0.1, eps 1e-5); F.batch_norm is the same ATen op the reference runs
            rm, rv = sd[b + "running_mean"].clone(), sd[b + "running_var"].clone()
            x = F.batch_norm(x, rm, rv, sd[b + "weight"], sd[b + "bias"], training=train, momentum=BN_MOMENTUM, eps=BN_EPS)
            if train and new_stats is not None:
                new_stats[b + "running_mean"] = rm
                new_stats[b + "running_var"] = rv
                new_stats[b + "num_batches_tracked"] = sd[b + "num_batches_tracked"] + 1
        x = _act(x, act)
        if train and dp > 0 and drop_masks is not None and i in drop_masks:
            x = x * drop_masks[i][:, :, None, None]
        if taps is not None:
            taps[i] = x
    return x


# ---------------------------------------------------------------------------
# decode -- yogo/model.py:277-313
# ---------------------------------------------------------------------------
def decode(
    raw: torch.Tensor, cxs: torch.Tensor, cys: torch.Tensor, anchor_w: float, anchor_h: float,
    width_multiplier: float = 1.0, height_multiplier: float = 1.0, inference: bool = False,
) -> torch.Tensor:
    _, _, Sy, Sx = raw.shape
    cls = torch.softmax(raw[:, 5:], dim=1) if inference else raw[:, 5:]
    wh = torch.clamp(raw[:, 2:4], max=80)
    aw = torch.tensor(anchor_w, dtype=torch.float32)
    ah = torch.tensor(anchor_h, dtype=torch.float32)
    wm = torch.tensor(width_multiplier, dtype=torch.float32)
    hm = torch.tensor(height_multiplier, dtype=torch.float32)
    return torch.cat(
        (
            ((1 / Sx) * torch.sigmoid(raw[:, 0]) + cxs)[:, None],
            ((1 / Sy) * torch.sigmoid(raw[:, 1]) + cys)[:, None],
            aw * torch.exp(wh[:, 0:1]) * wm,
            ah * torch.exp(wh[:, 1:2]) * hm,
            torch.sigmoid(raw[:, 4])[:, None],
            cls,
        ),
        dim=1,
    )


def yogo_forward(x, sd, spec, anchor_w, anchor_h, inference=False, train=False, drop_masks=None,
                 new_stats=None, width_multiplier=1.0, height_multiplier=1.0):
    """YOGO.forward, yogo/model.py:267-313 (grids taken from the state dict when present)."""
    if x.ndim == 3:
        x = x[None]
    if not x.is_floating_point():
        x = x.float()
    raw = backbone_forward(x, sd, spec, train=train, drop_masks=drop_masks, new_stats=new_stats)
    Sy, Sx = raw.shape[2:]
    if "_Cxs" in sd:
        cxs, cys = sd["_Cxs"], sd["_Cys"]
    else:
        cxs, cys = make_grids(Sx, Sy)
    return decode(raw, cxs, cys, anchor_w, anchor_h, width_multiplier, height_multiplier, inference)


def clamp_grads(grads: Dict[str, torch.Tensor], clip: float = 1.0) -> Dict[str, torch.Tensor]:
    """per-parameter hook of yogo/model.py:76-77."""
    return {k: torch.clamp(v, -clip, clip) for k, v in grads.items()}


# ---------------------------------------------------------------------------
# torchvision.ops restatement (third-party; torchvision>=0.14.1, not in the reference tree)
# ---------------------------------------------------------------------------
def box_convert_cxcywh_to_xyxy(b: torch.Tensor) -> torch.Tensor:
    """torchvision.ops._box_convert._box_cxcywh_to_xyxy: separate mul, then sub/add."""
    cx, cy, w, h = b.unbind(-1)
    x1 = cx - 0.5 * w
    y1 = cy - 0.5 * h
    x2 = cx + 0.5 * w
    y2 = cy + 0.5 * h
    return torch.stack((x1, y1, x2, y2), dim=-1)


def box_convert(boxes: torch.Tensor, in_fmt: str, out_fmt: str) -> torch.Tensor:
    if in_fmt == out_fmt:
        return boxes.clone()
    if (in_fmt, out_fmt) == ("cxcywh", "xyxy"):
        return box_convert_cxcywh_to_xyxy(boxes)
    if (in_fmt, out_fmt) == ("xyxy", "cxcywh"):
        x1, y1, x2, y2 = boxes.unbind(-1)
        return torch.stack(((x1 + x2) / 2, (y1 + y2) / 2, x2 - x1, y2 - y1), dim=-1)
    raise ValueError(f"unsupported conversion {in_fmt}->{out_fmt}")


def _loss_inter_union(b1, b2):
    x1, y1, x2, y2 = b1.unbind(dim=-1)
    x1g, y1g, x2g, y2g = b2.unbind(dim=-1)
    xkis1 = torch.max(x1, x1g)
    ykis1 = torch.max(y1, y1g)
    xkis2 = torch.min(x2, x2g)
    ykis2 = torch.min(y2, y2g)
    intsctk = torch.zeros_like(x1)
    mask = (ykis2 > ykis1) & (xkis2 > xkis1)
    intsctk[mask] = (xkis2[mask] - xkis1[mask]) * (ykis2[mask] - ykis1[mask])
    unionk = (x2 - x1) * (y2 - y1) + (x2g - x1g) * (y2g - y1g) - intsctk
    return intsctk, unionk


def complete_box_iou_loss(b1: torch.Tensor, b2: torch.Tensor, reduction: str = "none", eps: float = 1e-7) -> torch.Tensor:
    """torchvision.ops.complete_box_iou_loss (ciou_loss.py + diou_loss.py), reduction='none'|'sum'|'mean'."""
    b1 = b1.float() if not b1.is_floating_point() else b1
    b2 = b2.float() if not b2.is_floating_point() else b2
    intsct, union = _loss_inter_union(b1, b2)
    iou = intsct / (union + eps)
    x1, y1, x2, y2 = b1.unbind(dim=-1)
    x1g, y1g, x2g, y2g = b2.unbind(dim=-1)
    xc1 = torch.min(x1, x1g)
    yc1 = torch.min(y1, y1g)
    xc2 = torch.max(x2, x2g)
    yc2 = torch.max(y2, y2g)
    diag = ((xc2 - xc1) ** 2) + ((yc2 - yc1) ** 2) + eps
    x_p = (x2 + x1) / 2
    y_p = (y2 + y1) / 2
    x_g = (x1g + x2g) / 2
    y_g = (y1g + y2g) / 2
    dist = ((x_p - x_g) ** 2) + ((y_p - y_g) ** 2)
    diou = 1 - iou + (dist / diag)
    w_pred = x2 - x1
    h_pred = y2 - y1
    w_gt = x2g - x1g
    h_gt = y2g - y1g
    v = (4 / (torch.pi ** 2)) * torch.pow((torch.atan(w_gt / h_gt) - torch.atan(w_pred / h_pred)), 2)
    with torch.no_grad():
        alpha = v / (1 - iou + v + eps)
    loss = diou + alpha * v
    if reduction == "mean":
        loss = loss.mean() if loss.numel() > 0 else 0.0 * loss.sum()
    elif reduction == "sum":
        loss = loss.sum()
    return loss


def nms_numpy(boxes: np.ndarray, scores: np.ndarray, iou_threshold: float) -> np.ndarray:
    """torchvision CPU kernel nms_kernel_impl<float>: stable descending sort, greedy O(n^2),
    fp32 arithmetic, ``ovr > iou_threshold`` compared in double, no eps (0/0 = NaN: not suppressed).
    Returns kept ORIGINAL indices (int64) in descending-score order."""
    boxes = np.ascontiguousarray(boxes, dtype=np.float32)
    scores = np.ascontiguousarray(scores, dtype=np.float32)
    n = boxes.shape[0]
    if n == 0:
        return np.zeros((0,), dtype=np.int64)
    x1, y1, x2, y2 = boxes[:, 0], boxes[:, 1], boxes[:, 2], boxes[:, 3]
    areas = (x2 - x1) * (y2 - y1)
    order = np.argsort(-scores, kind="stable")
    nan = np.isnan(scores)
    if nan.any():  # torch sorts NaN as the largest value; np.argsort(-x) would put it last
        rest_idx = np.nonzero(~nan)[0]
        order = np.concatenate([np.nonzero(nan)[0], rest_idx[np.argsort(-scores[rest_idx], kind="stable")]])
    suppressed = np.zeros(n, dtype=bool)
    keep: List[int] = []
    thr = float(iou_threshold)
    with np.errstate(invalid="ignore", divide="ignore"):
        for _i in range(n):
            i = order[_i]
            if suppressed[i]:
                continue
            keep.append(int(i))
            rest = order[_i + 1:]
            if rest.size == 0:
                break
            xx1 = np.maximum(x1[i], x1[rest])
            yy1 = np.maximum(y1[i], y1[rest])
            xx2 = np.minimum(x2[i], x2[rest])
            yy2 = np.minimum(y2[i], y2[rest])
            w = np.maximum(np.float32(0), xx2 - xx1)
            h = np.maximum(np.float32(0), yy2 - yy1)
            inter = w * h
            ovr = inter / (areas[i] + areas[rest] - inter)
            suppressed[rest[ovr.astype(np.float64) > thr]] = True
    return np.asarray(keep, dtype=np.int64)


def nms(boxes: torch.Tensor, scores: torch.Tensor, iou_threshold: float) -> torch.Tensor:
    return torch.from_numpy(nms_numpy(boxes.detach().cpu().numpy(), scores.detach().cpu().numpy(), iou_threshold))


def box_iou(b1: torch.Tensor, b2: torch.Tensor) -> torch.Tensor:
    a1 = (b1[:, 2] - b1[:, 0]) * (b1[:, 3] - b1[:, 1])
    a2 = (b2[:, 2] - b2[:, 0]) * (b2[:, 3] - b2[:, 1])
    lt = torch.max(b1[:, None, :2], b2[:, :2])
    rb = torch.min(b1[:, None, 2:], b2[:, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[:, :, 0] * wh[:, :, 1]
    return inter / (a1[:, None] + a2 - inter)


# ---------------------------------------------------------------------------
# loss -- yogo/yogo_loss.py:38-129
# ---------------------------------------------------------------------------
def yogo_loss(
    pred: torch.Tensor, label: torch.Tensor, no_obj_weight: float = 0.5, iou_weight: float = 5.0,
    classify_weight: float = 1.0, label_smoothing: float = 0.01,
) -> Tuple[torch.Tensor, Dict[str, float]]:
    B, _, Sy, Sx = pred.shape
    fp = pred[:, :4].permute(1, 0, 2, 3).reshape(4, B * Sx * Sy)
    fl = label[:, 1:5].permute(1, 0, 2, 3).reshape(4, B * Sx * Sy)
    mask = label[:, 0:1].permute(1, 0, 2, 3).reshape(B * Sx * Sy).bool()
    fpm = fp[:, mask].permute(1, 0)
    flm = fl[:, mask].permute(1, 0)
    xyxy = box_convert_cxcywh_to_xyxy(fpm)
    valid = torch.logical_and(xyxy[:, 0] != xyxy[:, 2], xyxy[:, 1] != xyxy[:, 3])
    xyxy = xyxy[valid]
    flm = flm[valid]
    iou_loss = iou_weight * complete_box_iou_loss(torch.clamp(xyxy, min=0, max=1), flm).sum() / B
    cel = F.cross_entropy(pred[:, 5:], label[:, 5].long(), reduction="none", label_smoothing=label_smoothing)
    cls_loss = classify_weight * (label[:, 0] * cel).sum() / B
    obj_loss = (
        F.mse_loss(pred[:, 4], label[:, 0], reduction="none") * (label[:, 0] * (1 - no_obj_weight) + no_obj_weight)
    ).sum() / B
    loss = obj_loss + iou_loss + cls_loss
    return loss, {
        "iou_loss": float(iou_loss.item()),
        "objectness_loss": float(obj_loss.item()),
        "classification_loss": float(cls_loss.item()),
    }


# ---------------------------------------------------------------------------
# post-process -- yogo/utils/prediction_formatting.py:23-93
# ---------------------------------------------------------------------------
def format_preds(
    pred: torch.Tensor, obj_thresh: float = 0.5, iou_thresh: float = 0.5, box_format: str = "cxcywh",
    min_class_confidence_threshold: float = 0.0, return_cells: bool = False,
):
    if pred.ndim != 3:
        raise ValueError(f"argument to format_pred should be unbatched result - shape should be (pred_shape, Sy, Sx), got {pred.shape}")
    if box_format not in ("xyxy", "cxcywh"):
        raise ValueError(f"invalid box format {box_format}; valid box formats are ('xyxy', 'cxcywh')")
    P, Sy, Sx = pred.shape
    ref = pred.reshape(P, Sx * Sy).T
    m = ref[:, 4] > obj_thresh          # float32 compare (python scalar is cast to the tensor dtype)
    cells = torch.nonzero(m).flatten()
    preds = ref[m]
    if box_format == "xyxy":
        preds[:, :4] = box_convert_cxcywh_to_xyxy(preds[:, :4])
        nms_boxes = preds[:, :4]
    else:
        nms_boxes = box_convert_cxcywh_to_xyxy(preds[:, :4])
    if iou_thresh > 0:
        if preds.shape[0] > 0:
            keep = nms(nms_boxes, torch.max(preds[:, 5:], dim=1).values * preds[:, 4], iou_thresh)
        else:
            keep = torch.zeros(0, dtype=torch.long)
        preds = preds[keep]
        cells = cells[keep]
    if min_class_confidence_threshold > 0:
        k = preds[:, 5:].max(dim=1).values > min_class_confidence_threshold
        preds = preds[k]
        cells = cells[k]
    return (preds, cells) if return_cells else preds


def count_cells_for_formatted_preds(cls: torch.Tensor, min_confidence_threshold: Optional[float] = None) -> torch.Tensor:
    """yogo/infer.py:90-124"""
    if cls.ndim != 2:
        raise ValueError("expected formatted_class_predictions to be shape (N, num_classes)")
    if min_confidence_threshold is not None:
        if min_confidence_threshold < 0 or min_confidence_threshold > 1:
            raise ValueError(f"min_confidence_threshold should be between 0 and 1; is {min_confidence_threshold}")
    else:
        min_confidence_threshold = 0
    values, indices = cls.max(dim=1)
    return F.one_hot(indices[values > min_confidence_threshold], num_classes=cls.shape[1]).sum(dim=0)


def get_prediction_class_counts(batch_preds, obj_thresh=0.5, iou_thresh=0.5, min_class_confidence_threshold=0.0):
    """yogo/infer.py:60-87"""
    C = batch_preds.shape[1] - 5
    tot = torch.zeros(C, dtype=torch.long)
    for p in batch_preds:
        r = format_preds(p, obj_thresh, iou_thresh, min_class_confidence_threshold=min_class_confidence_threshold)
        if r.numel() == 0:
            continue
        tot += count_cells_for_formatted_preds(r[:, 5:])
    return tot


def format_preds_and_labels_v2(pred: torch.Tensor, label: torch.Tensor, objectness_thresh: float = 0.5,
                               min_class_confidence_threshold: float = 0.0):
    """prediction <-> label matching for the metrics (yogo/utils/prediction_formatting.py:254-330): threshold + NMS
    (iou 0.5, xyxy), pairwise IoU of labels x predictions, Hungarian assignment on 1 - IoU (scipy), then the matched rows
    plus the unmatched labels / predictions.  Returns (preds, labels, missed_labels, extra_predictions)."""
    from scipy.optimize import linear_sum_assignment

    pred = pred.squeeze()
    label = label.squeeze()
    if pred.ndim != 3:
        raise ValueError(f"argument to format_pred should be unbatched result - shape should be (pred_shape, Sy, Sx), got {pred.shape}")
    fp = format_preds(pred, obj_thresh=objectness_thresh, iou_thresh=0.5, box_format="xyxy",
                      min_class_confidence_threshold=min_class_confidence_threshold)
    L, Sy, Sx = label.shape
    labels = label.reshape(L, Sx * Sy).T
    fl = labels[labels[:, 0].bool()]
    M, N = fp.shape[0], fl.shape[0]
    cost = 1 - box_iou(fl[:, 1:5], fp[:, :4]).numpy()
    rows, cols = linear_sum_assignment(cost)
    rows_t, cols_t = torch.tensor(rows, dtype=torch.long), torch.tensor(cols, dtype=torch.long)
    un_p = torch.tensor([i for i in range(M) if i not in set(cols.tolist())], dtype=torch.long)
    un_l = torch.tensor([i for i in range(N) if i not in set(rows.tolist())], dtype=torch.long)
    return fp[cols_t], fl[rows_t], fl[un_l], fp[un_p]


def save_predictions_text(rows: torch.Tensor) -> str:
    """text of one image's prediction file: one "class xc yc w h" line per kept row, class = first argmax over the class
    columns, numbers printed as Python floats of the float32 values (yogo/infer.py:39-57, argmax helper :35-36)"""
    lines = []
    for pred in rows:
        cls = pred[5:].tolist()
        am = max(range(len(cls)), key=cls.__getitem__)
        lines.append(f"{am} {pred[0].item()} {pred[1].item()} {pred[2].item()} {pred[3].item()}")
    return "\n".join(lines)


def format_to_numpy(img_id: int, prediction_tensor: np.ndarray, img_h: int, img_w: int, np_dtype=np.float32) -> np.ndarray:
    """(8 + C) x N array: img id, x1*W, y1*H, x2*W, y2*H, objectness, argmax class, its probability, all class probabilities
    (yogo/utils/prediction_formatting.py:96-156; default thresholds, box_format="xyxy")"""
    fp = format_preds(torch.from_numpy(prediction_tensor), box_format="xyxy").numpy().T
    n = fp.shape[1]
    img_ids = np.ones(n).astype(np_dtype) * img_id
    tlx, tly, brx, bry = fp[0, :] * img_w, fp[1, :] * img_h, fp[2, :] * img_w, fp[3, :] * img_h
    objectness = fp[4, :].astype(np_dtype)
    all_confs = fp[5:, :].astype(np_dtype)
    pred_labels = np.argmax(all_confs, axis=0).astype(np.uint8)
    pred_probs = fp[5:,][pred_labels, np.arange(n)]
    return np.vstack((img_ids, tlx, tly, brx, bry, objectness, pred_labels.astype(np_dtype), pred_probs.astype(np_dtype), all_confs))


# ---------------------------------------------------------------------------
# label rasteriser -- yogo/data/yogo_dataset.py:24-46 ; synthetic inputs of SURVEY.md section 8(d)
# ---------------------------------------------------------------------------
def format_labels_tensor(labels: torch.Tensor, Sx: int, Sy: int) -> torch.Tensor:
    out = torch.zeros(6, Sy, Sx)
    iis = (labels[:, 1] + labels[:, 3]) * Sx // 2
    jjs = (labels[:, 2] + labels[:, 4]) * Sy // 2
    for i, j, lab in zip(iis.int(), jjs.int(), labels):
        out[0, j, i] = 1
        out[1:5, j, i] = lab[1:]
        out[5, j, i] = lab[0]
    return out


def label_rows_to_tensor(rows: torch.Tensor, Sx: int, Sy: int) -> torch.Tensor:
    """yogo/data/yogo_dataset.py:113-133 (label_file_to_tensor after parsing): rows (class, xc, yc, w, h) -> label tensor."""
    rows = rows.clone().float().reshape(-1, 5)
    if rows.nelement() == 0:
        return torch.zeros(6, Sy, Sx)
    rows[:, 1:] = box_convert(rows[:, 1:], "cxcywh", "xyxy")
    return format_labels_tensor(rows, Sx, Sy)


def hflip_with_bbs(img: torch.Tensor, lab: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """yogo/data/data_transforms.py:51-74, the flipped branch: x1, x2 <- 1 - x2, 1 - x1 in every cell, then mirror along W."""
    lab = lab.clone()
    a, b = 1 - lab[:, 3, :, :], 1 - lab[:, 1, :, :]
    lab[:, 1, :, :], lab[:, 3, :, :] = a, b
    return torch.flip(img, dims=(3,)), torch.flip(lab, dims=(3,))


def vflip_with_bbs(img: torch.Tensor, lab: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """yogo/data/data_transforms.py:77-98, the flipped branch."""
    lab = lab.clone()
    a, b = 1 - lab[:, 4, :, :], 1 - lab[:, 2, :, :]
    lab[:, 2, :, :], lab[:, 4, :, :] = a, b
    return torch.flip(img, dims=(2,)), torch.flip(lab, dims=(2,))


def random_flips_with_bbs(img: torch.Tensor, lab: torch.Tensor, p_h: float = 0.5, p_v: float = 0.5) -> Tuple[torch.Tensor, torch.Tensor]:
    """the training augmentation of yogo/data/yogo_dataloader.py:203-210: one torch.rand(1) draw per transform and batch."""
    if torch.rand(1) < p_h:
        img, lab = hflip_with_bbs(img, lab)
    if torch.rand(1) < p_v:
        img, lab = vflip_with_bbs(img, lab)
    return img, lab


def synthetic_images(B: int, H: int = 772, W: int = 1032, seed: int = 0) -> torch.Tensor:
    g = torch.Generator().manual_seed(seed)
    return torch.randint(0, 256, (B, 1, H, W), dtype=torch.uint8, generator=g)


def synthetic_labels(B: int, Sx: int, Sy: int, K: int = 64, num_classes: int = 7, seed: int = 1,
                     anchor_w: float = 0.0425, anchor_h: float = 0.0555) -> torch.Tensor:
    g = torch.Generator().manual_seed(seed)
    out = torch.zeros(B, 6, Sy, Sx)
    for b in range(B):
        c = torch.rand(K, 2, generator=g) * 0.9 + 0.05
        w = anchor_w * torch.exp(torch.randn(K, generator=g) * 0.2)
        h = anchor_h * torch.exp(torch.randn(K, generator=g) * 0.2)
        cls = torch.randint(0, num_classes, (K,), generator=g).float()
        lab = torch.stack((cls, c[:, 0] - w / 2, c[:, 1] - h / 2, c[:, 0] + w / 2, c[:, 1] + h / 2), dim=1)
        out[b] = format_labels_tensor(lab, Sx, Sy)
    return out


def synthetic_predictions(B: int, Sx: int, Sy: int, num_classes: int = 7, K: int = 100, seed: int = 2) -> torch.Tensor:
    """'realistic' NMS input of SURVEY.md 8(d): K objects/img, each predicted by 1-4 neighbouring
    cells with jittered boxes, obj U(0.5,1), softmaxed random class logits; other cells obj U(0,0.4)."""
    g = torch.Generator().manual_seed(seed)
    P = 5 + num_classes
    out = torch.zeros(B, P, Sy, Sx)
    out[:, 4] = torch.rand(B, Sy, Sx, generator=g) * 0.4
    out[:, 0] = (torch.arange(Sx).float()[None, None, :] + 0.5) / Sx
    out[:, 1] = (torch.arange(Sy).float()[None, :, None] + 0.5) / Sy
    out[:, 2] = 0.0425
    out[:, 3] = 0.0555
    out[:, 5:] = torch.softmax(torch.randn(B, num_classes, Sy, Sx, generator=g), dim=1)
    for b in range(B):
        cx = torch.rand(K, generator=g) * 0.9 + 0.05
        cy = torch.rand(K, generator=g) * 0.9 + 0.05
        w = 0.0425 * torch.exp(torch.randn(K, generator=g) * 0.2)
        h = 0.0555 * torch.exp(torch.randn(K, generator=g) * 0.2)
        ncell = torch.randint(1, 5, (K,), generator=g)
        for k in range(K):
            i0 = int(cx[k] * Sx)
            j0 = int(cy[k] * Sy)
            logits = torch.randn(num_classes, generator=g) * 2
            for (di, dj) in [(0, 0), (1, 0), (0, 1), (1, 1)][: int(ncell[k])]:
                i, j = min(i0 + di, Sx - 1), min(j0 + dj, Sy - 1)
                jit = 0.05 * torch.randn(4, generator=g)
                out[b, 0, j, i] = cx[k] + w[k] * jit[0]
                out[b, 1, j, i] = cy[k] + h[k] * jit[1]
                out[b, 2, j, i] = w[k] * (1 + jit[2])
                out[b, 3, j, i] = h[k] * (1 + jit[3])
                out[b, 4, j, i] = 0.5 + 0.5 * torch.rand(1, generator=g).item()
                out[b, 5:, j, i] = torch.softmax(logits + 0.3 * torch.randn(num_classes, generator=g), dim=0)
    return out


# ---------------------------------------------------------------------------
# optimiser glue -- yogo/train.py:213-223 (AdamW over ALL parameters, cosine LR per step)
# ---------------------------------------------------------------------------
def adamw_step(p, g, m, v, step: int, lr: float, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=5e-2):
    """torch.optim.AdamW single-tensor update (decoupled decay, bias correction); step is 1-based."""
    p = p * (1 - lr * weight_decay)
    m = beta1 * m + (1 - beta1) * g
    v = beta2 * v + (1 - beta2) * g * g
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = v.sqrt() / math.sqrt(bc2) + eps
    p = p - (lr / bc1) * m / denom
    return p, m, v


def cosine_lr(step: int, base_lr: float, t_max: int, eta_min: float) -> float:
    """closed form of torch CosineAnnealingLR after ``step`` scheduler steps."""
    return eta_min + (base_lr - eta_min) * (1 + math.cos(math.pi * step / t_max)) / 2


# ---------------------------------------------------------------------------
# bf16-STORAGE emulation of the training step (test infrastructure, like everything in this file)
#
# The reference trains under autocast (yogo/train.py:309-322); the HIP path stores activations and activation gradients
# as bf16 and feeds bf16 weights to the matrix cores, with fp32 accumulation, fp32/fp64 statistics and fp32 parameter
# gradients.  A plain fp32 step can only bound such a step loosely (bf16 rounding of eight layers of activations moves
# every gradient by ~1 %).  This function restates the SAME arithmetic as `yogo_forward` + `yogo_loss` + autograd, but
# rounds to bf16 exactly where the HIP path stores bf16 -- so the whole step can be held to ~1e-3, and a kernel error of
# a few per cent is no longer hidden behind "bf16 noise".  Rounding points (file:kernel of yogo_amd/csrc):
#   * weights multiplied as bf16 (conv_bf16.hip:conv_bf16_pack_kernel; layer 0: conv_first_mfma.hip `wa`)
#   * block without BatchNorm: y = bf16(act(conv + bias) * mask); LeakyReLU sign taken from that value (sign map / stored y)
#   * block with BatchNorm: z = bf16(conv + bias); batch statistics of the STORED z (layer 0 on the matrix cores: of the
#     unrounded convolution, from the exact patch Gram matrix); y = bf16(act(z * sc + sh)), sc = invstd * gamma,
#     sh = fma(-mean, sc, beta)   (bn.hip:bn_apply_act_8c_kernel)
#   * head: fp32 output; loss in fp32; d loss / d raw rounded to bf16 (decode_loss.hip:decode_loss_bwd_bf16_kernel)
#   * BatchNorm backward on bf16 g and z, dz = bf16(...) (bn.hip:bn_bwd_*_8c_kernel); data gradients dx = bf16(convT(g, bf16 w)
#     * act'(previous block) * mask); weight / bias gradients fp32 from the bf16 tensors
#   * layer 0 backward: the fused sums of conv_first.hip:conv_first_bn_wgrad_kernel (+ the Gram form of sum xhat * patch)
# ---------------------------------------------------------------------------
def _rb(t: torch.Tensor) -> torch.Tensor:
    """round to bf16 (nearest even) and widen again"""
    return t.to(torch.bfloat16).to(torch.float32)


def _act_bwd_factor(ref: torch.Tensor, act: Optional[str]) -> torch.Tensor:
    """csrc/common.h:act_bwd_factor -- leaky: by the sign of the OUTPUT, silu: from the PRE-activation"""
    if act == "leaky":
        return torch.where(ref > 0, torch.ones_like(ref), torch.full_like(ref, LEAKY_SLOPE))
    if act == "silu":
        s = torch.sigmoid(ref)
        return s * (1 + ref * (1 - s))
    return torch.ones_like(ref)


def l0_on_matrix_cores(spec: list, x: torch.Tensor) -> bool:
    """the shapes yogo_conv_first_mfma_supported takes (uint8, 1 -> <=16 channels, stride 2, even sizes, BatchNorm, no dropout)"""
    co, k, s, hb, hbn, act, dp = spec[0]
    H, W = x.shape[-2:]
    return bool(x.dtype == torch.uint8 and x.shape[1] == 1 and co <= 16 and s == 2 and k == 3 and H % 2 == 0 and W % 2 == 0
                and H >= 4 and W >= 4 and hbn)


def bf16_block_forward(i: int, x_in: torch.Tensor, sd: Dict[str, torch.Tensor], spec: list, l0_mfma: bool,
                       mask: Optional[torch.Tensor] = None, z_given: Optional[torch.Tensor] = None,
                       stats_given: Optional[Tuple[torch.Tensor, torch.Tensor]] = None) -> Dict[str, torch.Tensor]:
    """block i of the bf16-storage forward from the (bf16-valued, fp32-typed) input ``x_in``: returns a dict with ``y`` and, for
    BatchNorm blocks, ``z``, ``mean``, ``invstd``, ``mean64``, ``var64`` (``pre`` for SiLU blocks without BatchNorm).
    ``z_given`` / ``stats_given`` = (mean, invstd) substitute the stored conv output / the batch statistics (teacher-forced
    checks of the BatchNorm kernels alone: statistics from a given z, apply from given z and statistics)."""
    n = len(spec)
    co, k, s, hb, hbn, act, dp = spec[i]
    f64 = torch.float64
    pre = conv_prefix(spec, i)
    w = sd[pre + "weight"].float()
    wq = w if (i == 0 and not l0_mfma) else _rb(w)   # the direct layer-0 kernels multiply fp32 weights
    bias = sd[pre + "bias"].float() if hb else None
    a = F.conv2d(x_in, wq, bias, stride=s, padding=1 if k == 3 else 0)
    S: Dict[str, torch.Tensor] = {"x": x_in, "wq": wq, "act": act, "bn": bool(hbn), "k": k, "s": s, "mask": mask}
    if hbn:
        bpre = f"model.{i}.1."
        z = _rb(a) if z_given is None else z_given
        src = a if (i == 0 and z_given is None) else z   # layer 0: statistics of the unrounded convolution (Gram form / fp32 epilogue sums)
        cnt = src.numel() // src.shape[1]
        mean64 = src.to(f64).mean(dim=(0, 2, 3))
        var64 = ((src.to(f64) ** 2).mean(dim=(0, 2, 3)) - mean64 ** 2).clamp_min(0)
        if stats_given is None:
            mean, invstd = mean64.float(), (1.0 / torch.sqrt(var64 + BN_EPS)).float()
        else:
            mean, invstd = stats_given
        gamma, beta = sd[bpre + "weight"].float(), sd[bpre + "bias"].float()
        sc = invstd * gamma
        sh = torch.addcmul(beta, -mean, sc)
        y = _rb(_act(z * sc[None, :, None, None] + sh[None, :, None, None], act))
        S.update(z=z, mean=mean, invstd=invstd, gamma=gamma, beta=beta, y=y, mean64=mean64, var64=var64, count=cnt)
        if i == 0:
            S["a"] = a   # the unrounded convolution (layer 0 of the engine may keep no z: bf16_block_backward(l0_no_z=True))
    elif i == n - 1:
        S.update(y=a)   # fp32 head
    else:
        v = _act(a, act)
        if mask is not None:
            v = v * mask[:, :, None, None]
        S.update(y=_rb(v), pre=_rb(a) if act == "silu" else None)
    return S


def bf16_head_gradient(raw: torch.Tensor, sd: Dict[str, torch.Tensor], label: torch.Tensor, anchor_w: float, anchor_h: float,
                       no_obj_weight: float = 0.5, iou_weight: float = 5.0, classify_weight: float = 1.0,
                       label_smoothing: float = 0.01):
    """decode + loss in fp32 on the fp32 head output, d loss / d raw rounded to bf16 (decode_loss.hip:decode_loss_bwd_bf16_kernel).
    Returns (loss tensor, components, bf16-rounded gradient, fp32 gradient)."""
    raw_l = raw.detach().clone().requires_grad_(True)
    Sy, Sx = raw.shape[2:]
    if "_Cxs" in sd:
        cxs, cys = sd["_Cxs"], sd["_Cys"]
    else:
        cxs, cys = make_grids(Sx, Sy)
    pred = decode(raw_l, cxs, cys, anchor_w, anchor_h)
    loss, comps = yogo_loss(pred, label.float(), no_obj_weight, iou_weight, classify_weight, label_smoothing)
    (g,) = torch.autograd.grad(loss, raw_l)
    return loss.detach(), comps, _rb(g), g


# Layer 0 of the engine keeps no conv output when its fused backward sweep can do without (engine._L0_NO_Z: one-channel uint8 image
# on the matrix cores, 8 or 16 output channels, BatchNorm, no conv bias, no activation or LeakyReLU): bf16_train_step follows.
L0_NO_Z = True


def l0_keeps_no_z(spec: list, l0_mfma: bool) -> bool:
    co, k, s, hb, hbn, act, dp = spec[0]
    return bool(L0_NO_Z and l0_mfma and hbn and not hb and act in (None, "leaky") and co in (8, 16))


def l0_sign_map(S: Dict[str, torch.Tensor]) -> torch.Tensor:
    """what yogo_conv_first_mfma_signs writes beside y: uint8 [B][OH*OW*2]; byte h of a pixel: bit i = (BatchNorm output of channel
    4h + i > 0), bit 4 + i = (... of channel 8 + 4h + i > 0); the BatchNorm output = z * sc + sh of the ROUNDED z (what y is made of)."""
    z, mean, invstd, gamma, beta = S["z"], S["mean"], S["invstd"], S["gamma"], S["beta"]
    sc = invstd * gamma
    sh = torch.addcmul(beta, -mean, sc)
    pos = (z * sc[None, :, None, None] + sh[None, :, None, None]) > 0
    B, co, OH, OW = pos.shape
    full = torch.zeros(B, 16, OH * OW, dtype=torch.int32)
    full[:, :co] = pos.reshape(B, co, -1).to(torch.int32)
    out = torch.zeros(B, OH * OW, 2, dtype=torch.int32)
    for h in range(2):
        for i in range(4):
            out[:, :, h] |= full[:, 4 * h + i] << i
            out[:, :, h] |= full[:, 8 + 4 * h + i] << (4 + i)
    return out.to(torch.uint8).reshape(B, OH * OW * 2)


def bf16_block_backward(i: int, g: torch.Tensor, S: Dict[str, torch.Tensor], Sp: Optional[Dict[str, torch.Tensor]], spec: list,
                        l0_mfma: bool, dz_given: Optional[torch.Tensor] = None, l0_no_z: bool = False) -> Dict[str, torch.Tensor]:
    """block i of the bf16-storage backward from ``g`` = gradient w.r.t. the block's output (bf16-valued): returns a dict with
    ``dW``, optionally ``db``, ``dgamma``, ``dbeta``, ``dz`` (the bf16 BatchNorm-backward output) and ``dx`` (the bf16 gradient
    w.r.t. the previous block's output, with that block's activation derivative and dropout mask applied when it has no
    BatchNorm).  ``S`` / ``Sp``: the forward records of this / the previous block (bf16_block_forward)."""
    co, k, s, hb, hbn, act, dp = spec[i]
    f64 = torch.float64
    B = g.shape[0]
    out: Dict[str, torch.Tensor] = {}
    N = g.numel() // g.shape[1]
    if hbn and i == 0 and not hb and act in (None, "leaky"):
        # conv_first_bn_wgrad_kernel + finalize (what the engine runs for a bias-free first conv + BatchNorm with no /
        # LeakyReLU activation): dz never exists; dW = c1 (A1 - S1/N P - S2/N A2)
        z, mean, invstd, gamma, beta = S["z"], S["mean"], S["invstd"], S["gamma"], S["beta"]
        xh = (z - mean[None, :, None, None]) * invstd[None, :, None, None]
        if l0_no_z:
            # the engine kept the SIGN MAP of the BatchNorm output instead of z (conv_first_mfma_kernel: r = z * sc + sh of the rounded
            # z, the value y is made of) and derives sum gb * xhat from the weight-gradient sums, i.e. from the UNROUNDED convolution
            # (conv_first_bn_wgrad_pk_kernel<true> + finalize, derive_s2): xhat of `a`, not of bf16(a)
            sc = invstd * gamma
            sh = torch.addcmul(beta, -mean, sc)
            yb = z * sc[None, :, None, None] + sh[None, :, None, None]
            xh = (S["a"].to(f64) - mean.to(f64)[None, :, None, None]) * invstd.to(f64)[None, :, None, None]
        else:
            yb = torch.addcmul(beta[None, :, None, None], gamma[None, :, None, None], xh)
        gb = g * _act_bwd_factor(yb, act)
        S1 = gb.to(f64).sum(dim=(0, 2, 3))
        S2 = (gb.to(f64) * xh.to(f64)).sum(dim=(0, 2, 3))
        cin = S["x"].shape[1]
        patches = F.unfold(S["x"], kernel_size=3, padding=1, stride=s).to(f64)          # [B, cin*9, L]
        A1 = torch.einsum("bcl,bjl->cj", gb.reshape(B, co, -1).to(f64), patches)           # [co, cin*9]
        P = patches.sum(dim=(0, 2))                                                         # [cin*9]
        if l0_mfma:   # Gram form: sum xhat * patch_j from the UNROUNDED convolution, with the weights the forward used
            G = torch.einsum("bjl,bkl->jk", patches, patches)
            Wm = S["wq"].reshape(co, -1).to(f64)
            A2 = invstd.to(f64)[:, None] * (Wm @ G - mean.to(f64)[:, None] * P[None, :])
        else:
            A2 = torch.einsum("bcl,bjl->cj", xh.reshape(B, co, -1).to(f64), patches)
        c1 = (gamma * invstd).to(f64)
        dW = c1[:, None] * (A1 - (S1 / N)[:, None] * P[None, :] - (S2 / N)[:, None] * A2)
        out.update(dW=dW.float().reshape(co, cin, 3, 3), dgamma=S2.float(), dbeta=S1.float())
        return out
    if hbn:
        z, mean, invstd, gamma, beta = S["z"], S["mean"], S["invstd"], S["gamma"], S["beta"]
        xh = (z - mean[None, :, None, None]) * invstd[None, :, None, None]
        ge = g * _act_bwd_factor(torch.addcmul(beta[None, :, None, None], xh, gamma[None, :, None, None]), act)
        S1 = ge.to(f64).sum(dim=(0, 2, 3))
        S2 = (ge.to(f64) * xh.to(f64)).sum(dim=(0, 2, 3))
        out.update(dgamma=S2.float(), dbeta=S1.float())
        mg = (S1.float() * (1.0 / float(N)))[None, :, None, None]
        mgx = (S2.float() * (1.0 / float(N)))[None, :, None, None]
        g = _rb((invstd * gamma)[None, :, None, None] * (ge - mg - xh * mgx))
        out["dz"] = g
        if dz_given is not None:
            g = dz_given
    # ---- weight / bias gradient: fp32 sums over the bf16 tensors ---------------------------------------------------------
    xl = S["x"].detach().clone().requires_grad_(True)
    wl = S["wq"].detach().clone().requires_grad_(True)
    o = F.conv2d(xl, wl, None, stride=s, padding=1 if k == 3 else 0)
    need_dx = i > 0
    gs = torch.autograd.grad(o, (xl, wl) if need_dx else (wl,), g)
    out["dW"] = gs[-1]
    if hb:
        out["db"] = g.to(f64).sum(dim=(0, 2, 3)).float()
        out["db_abs"] = g.to(f64).abs().sum(dim=(0, 2, 3)).float()   # (scale of the sum: a conv bias in front of BatchNorm sums to ~0)
    if need_dx:
        dx = gs[0]
        if not Sp["bn"]:   # the data gradient's epilogue applies the previous block's activation derivative and dropout mask
            if Sp["act"] == "leaky":
                dx = dx * _act_bwd_factor(Sp["y"], "leaky")
            elif Sp["act"] == "silu":
                dx = dx * _act_bwd_factor(Sp["pre"], "silu")
            if Sp.get("mask") is not None:
                dx = dx * Sp["mask"][:, :, None, None]
        out["dx"] = _rb(dx)
    return out


def bf16_train_step(
    x: torch.Tensor, sd: Dict[str, torch.Tensor], spec: list, label: torch.Tensor, anchor_w: float, anchor_h: float,
    no_obj_weight: float = 0.5, iou_weight: float = 5.0, classify_weight: float = 1.0, label_smoothing: float = 0.01,
    clip: float = 0.0, drop_masks: Optional[Dict[int, torch.Tensor]] = None, taps: Optional[Dict[str, torch.Tensor]] = None,
) -> Tuple[float, Dict[str, float], Dict[str, torch.Tensor], Dict[str, torch.Tensor]]:
    """One bf16-storage training step's loss and parameter gradients (train mode, batch statistics).

    Returns (loss, components, grads by state-dict name, new running statistics).  ``clip`` > 0 clamps every gradient
    tensor to +-clip (yogo/model.py:76-77).  ``taps`` (optional dict) receives intermediate tensors: "y{i}", "z{i}", "g{i}"
    (gradient w.r.t. block i's output), "dz{i}" for layer-wise bisection.
    """
    n = len(spec)
    l0_mfma = l0_on_matrix_cores(spec, x)
    cur = x.float()   # (uint8 is exact in bf16; a float image is used as it is by the direct layer-0 kernels)
    saved = []
    new_stats: Dict[str, torch.Tensor] = {}
    for i, (co, k, s, hb, hbn, act, dp) in enumerate(spec):
        mask = drop_masks.get(i) if (drop_masks is not None and dp > 0) else None
        S = bf16_block_forward(i, cur, sd, spec, l0_mfma, mask)
        if hbn:
            bpre = f"model.{i}.1."
            cnt = S["count"]
            unbiased = S["var64"] * cnt / max(cnt - 1, 1)
            new_stats[bpre + "running_mean"] = ((1 - BN_MOMENTUM) * sd[bpre + "running_mean"].double() + BN_MOMENTUM * S["mean64"]).float()
            new_stats[bpre + "running_var"] = ((1 - BN_MOMENTUM) * sd[bpre + "running_var"].double() + BN_MOMENTUM * unbiased).float()
            new_stats[bpre + "num_batches_tracked"] = sd[bpre + "num_batches_tracked"] + 1
        if taps is not None:
            taps[f"y{i}"] = S["y"]
            if hbn:
                taps[f"z{i}"] = S["z"]
        saved.append(S)
        cur = S["y"]
    loss, comps, g, g32 = bf16_head_gradient(cur, sd, label, anchor_w, anchor_h, no_obj_weight, iou_weight, classify_weight, label_smoothing)
    if taps is not None:
        taps["graw_f32"] = g32
        taps["saved"] = saved
    grads: Dict[str, torch.Tensor] = {}

    def fin(t: torch.Tensor) -> torch.Tensor:
        return torch.clamp(t, -clip, clip) if clip > 0 else t

    for i in range(n - 1, -1, -1):
        co, k, s, hb, hbn, act, dp = spec[i]
        pre = conv_prefix(spec, i)
        if taps is not None:
            taps[f"g{i}"] = g
        r = bf16_block_backward(i, g, saved[i], saved[i - 1] if i > 0 else None, spec, l0_mfma, l0_no_z=(i == 0 and l0_keeps_no_z(spec, l0_mfma)))
        grads[pre + "weight"] = fin(r["dW"])
        if "db" in r:
            grads[pre + "bias"] = fin(r["db"])
        if "dgamma" in r:
            grads[f"model.{i}.1.weight"] = fin(r["dgamma"])
            grads[f"model.{i}.1.bias"] = fin(r["dbeta"])
        if taps is not None and "dz" in r:
            taps[f"dz{i}"] = r["dz"]
        if i > 0:
            g = r["dx"]
    return float(loss), comps, grads, new_stats
