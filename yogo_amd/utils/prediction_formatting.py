"""``format_preds`` and its batched form on the HIP decode/threshold/NMS kernel.

Reference: yogo/utils/prediction_formatting.py:23-93 (``format_preds``), yogo/infer.py:60-124
(``get_prediction_class_counts``, ``count_cells_for_formatted_preds``).  The reference runs ``format_preds`` once per
image from a Python loop; here the whole batch is one kernel launch and the per-image results are slices of it.
"""
from __future__ import annotations

from typing import List, Literal, Optional, Tuple, Union, get_args

import torch

from yogo_amd import _hip

BoxFormat = Literal["xyxy", "cxcywh"]


class RawPredictions:
    """The head's output of a batch BEFORE the box decode, with the decode's operands (``YOGO.forward_raw``).  Handed to
    ``format_preds_batched`` (and everything built on it) in place of ``model(x)``, the decode of yogo/model.py:277-313 runs inside
    the threshold + NMS kernel's loads (``yogo_decode_format_preds_batched``): the decoded tensor never goes through memory and the
    rows / cells / counts are bit-identical to ``format_preds_batched(model(x))``.  ``decoded()`` gives that tensor when it is
    wanted after all."""

    def __init__(self, raw: torch.Tensor, cxs: torch.Tensor, cys: torch.Tensor, anchor_w: float, anchor_h: float,
                 width_multiplier: float, height_multiplier: float, inference: bool):
        if raw.ndim != 4:
            raise ValueError(f"RawPredictions expects (B, pred_shape, Sy, Sx), got {tuple(raw.shape)}")
        _hip.require_cuda(raw, "the raw prediction")
        Sy, Sx = raw.shape[2:]
        if tuple(cxs.shape) != (Sy, Sx) or tuple(cys.shape) != (Sy, Sx):
            raise RuntimeError(f"yogo_amd: grid buffers {tuple(cxs.shape)} do not match the network output grid ({Sy}, {Sx})")
        self.raw = raw.detach().contiguous().float()
        self.cxs, self.cys = cxs.contiguous(), cys.contiguous()
        self.scalars = (float(anchor_w), float(anchor_h), float(width_multiplier), float(height_multiplier))
        self.inference = bool(inference)

    @property
    def shape(self) -> torch.Size:
        return self.raw.shape

    @property
    def device(self) -> torch.device:
        return self.raw.device

    def decoded(self) -> torch.Tensor:
        B, P, Sy, Sx = self.raw.shape
        out = torch.empty_like(self.raw)
        if B:
            with torch.cuda.device(self.raw.device):
                _hip.call("yogo_decode_fwd", self.raw, out, self.cxs, self.cys, B, P, Sy, Sx, *self.scalars, int(self.inference),
                          _hip.stream_ptr())
        return out


def format_preds_batched(
    pred: Union[torch.Tensor, RawPredictions],
    obj_thresh: float = 0.5,
    iou_thresh: float = 0.5,
    box_format: BoxFormat = "cxcywh",
    min_class_confidence_threshold: float = 0.0,
) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """pred [B, 5+C, Sy, Sx] on the GPU -> (rows [B, cap, 5+C], cells int64 [B, cap], counts int32 [B]).

    Image b's result is ``rows[b, :counts[b]]`` -- the same rows in the same order ``format_preds(pred[b])`` returns;
    ``cells`` are the flat grid-cell indices (y*Sx + x) the rows came from.  No host synchronisation happens here.
    A ``RawPredictions`` (``YOGO.forward_raw``) is decoded inside the kernel, with the same result as its decoded tensor.
    """
    raw = pred if isinstance(pred, RawPredictions) else None
    if len(pred.shape) != 4:
        raise ValueError(f"format_preds_batched expects (B, pred_shape, Sy, Sx), got {tuple(pred.shape)}")
    if box_format not in get_args(BoxFormat):
        raise ValueError(f"invalid box format {box_format}; valid box formats are {get_args(BoxFormat)}")
    if raw is None:
        _hip.require_cuda(pred, "pred")
    B, P, Sy, Sx = pred.shape
    p = raw.raw if raw is not None else pred.detach().contiguous().float()
    cap = Sy * Sx
    dev = pred.device
    rows = torch.empty(B, cap, P, dtype=torch.float32, device=dev)
    cells = torch.empty(B, cap, dtype=torch.int64, device=dev)
    counts = torch.empty(B, dtype=torch.int32, device=dev)
    if B == 0:
        return rows, cells, counts
    with torch.cuda.device(dev):
        ws = torch.empty(_hip.query_size("yogo_format_preds_workspace_bytes", B, Sy, Sx), dtype=torch.uint8, device=dev)
        tail = (float(obj_thresh), float(iou_thresh), 0 if box_format == "cxcywh" else 1, float(min_class_confidence_threshold),
                _hip.stream_ptr())
        if raw is not None:
            _hip.call("yogo_decode_format_preds_batched", p, raw.cxs, raw.cys, rows, cells, counts, ws, B, P, Sy, Sx, cap, *raw.scalars,
                      int(raw.inference), *tail)
        else:
            _hip.call("yogo_format_preds_batched", p, rows, cells, counts, ws, B, P, Sy, Sx, cap, *tail)
    return rows, cells, counts


def format_preds(
    pred: torch.Tensor,
    obj_thresh: float = 0.5,
    iou_thresh: float = 0.5,
    box_format: BoxFormat = "cxcywh",
    min_class_confidence_threshold: float = 0.0,
) -> torch.Tensor:
    """formats pred, prediction tensor straight from YOGO (unbatched, [5+C, Sy, Sx]), into [N, 5+C] after objectness
    thresholding, NMS and the class-confidence filter.  For all thresholds, set to 0 to disable."""
    if len(pred.shape) != 3:
        raise ValueError(
            "argument to format_pred should be unbatched result - " f"shape should be (pred_shape, Sy, Sx), got {pred.shape}"
        )
    elif box_format not in get_args(BoxFormat):
        raise ValueError(f"invalid box format {box_format}; valid box formats are {get_args(BoxFormat)}")
    rows, _, counts = format_preds_batched(pred[None], obj_thresh, iou_thresh, box_format, min_class_confidence_threshold)
    n = int(counts[0].item())
    return rows[0, :n]


def split_batched(rows: torch.Tensor, counts: torch.Tensor) -> List[torch.Tensor]:
    """one device->host copy of the counts, then views"""
    c = counts.cpu().tolist()
    return [rows[b, : c[b]] for b in range(len(c))]


def count_cells_for_formatted_preds(formatted_class_predictions: torch.Tensor,
                                    min_confidence_threshold: Optional[float] = None) -> torch.Tensor:
    """class histogram by argmax over rows whose max confidence exceeds the threshold (yogo/infer.py:90-124)"""
    if not len(formatted_class_predictions.shape) == 2:
        raise ValueError(f"expected formatted_class_predictions to be shape (N, num_classes); got {formatted_class_predictions.shape}")
    if min_confidence_threshold is not None:
        if min_confidence_threshold < 0 or min_confidence_threshold > 1:
            raise ValueError(f"min_confidence_threshold should be between 0 and 1; is {min_confidence_threshold}")
    else:
        min_confidence_threshold = 0
    _, n_classes = formatted_class_predictions.shape
    values, indices = formatted_class_predictions.max(dim=1)
    mask = values > min_confidence_threshold
    return torch.nn.functional.one_hot(indices[mask], num_classes=n_classes).sum(dim=0)


def get_prediction_class_counts(batch_preds: torch.Tensor, obj_thresh=0.5, iou_thresh=0.5,
                                min_class_confidence_threshold: float = 0) -> torch.Tensor:
    """yogo/infer.py:60-87 over one batched launch"""
    bs, pred_dim, Sy, Sx = batch_preds.shape
    num_classes = pred_dim - 5
    rows, _, counts = format_preds_batched(batch_preds, obj_thresh, iou_thresh, "cxcywh", min_class_confidence_threshold)
    tot = torch.zeros(num_classes, dtype=torch.long)
    for r in split_batched(rows, counts):
        if r.numel() == 0:
            continue
        tot += count_cells_for_formatted_preds(r[:, 5:]).cpu()
    return tot


def _argmax_first(vals) -> int:
    return max(range(len(vals)), key=vals.__getitem__)


def prediction_rows_to_text(rows: torch.Tensor) -> str:
    """one image's prediction file: "class xc yc w h" per kept row, numbers as Python floats of the float32 values
    (the f-string of yogo/infer.py:52-55; `rows` on the host)"""
    out = []
    for r in rows.tolist():
        out.append(f"{_argmax_first(r[5:])} {r[0]} {r[1]} {r[2]} {r[3]}")
    return "\n".join(out)


def save_predictions(fnames, batch_preds: torch.Tensor, obj_thresh=0.5, iou_thresh=0.5) -> None:
    """yogo/infer.py:39-57 with ONE batched threshold + NMS launch and one device->host copy for the whole batch"""
    rows, _, counts = format_preds_batched(batch_preds, obj_thresh, iou_thresh)
    host = rows.cpu()
    for fname, n, r in zip(fnames, counts.cpu().tolist(), host):
        with open(fname, "w") as f:
            f.write(prediction_rows_to_text(r[:n]))


def _rows_xyxy_to_numpy(img_id: int, rows_xyxy, img_h: int, img_w: int, np_dtype):
    import numpy as np

    fp = rows_xyxy.numpy().T
    n = fp.shape[1]
    img_ids = np.ones(n).astype(np_dtype) * img_id
    tlx, tly, brx, bry = fp[0, :] * img_w, fp[1, :] * img_h, fp[2, :] * img_w, fp[3, :] * img_h
    objectness = fp[4, :].astype(np_dtype)
    all_confs = fp[5:, :].astype(np_dtype)
    pred_labels = np.argmax(all_confs, axis=0).astype(np.uint8)
    pred_probs = fp[5:,][pred_labels, np.arange(n)]
    return np.vstack((img_ids, tlx, tly, brx, bry, objectness, pred_labels.astype(np_dtype), pred_probs.astype(np_dtype), all_confs))


def format_to_numpy_batched(img_ids, batch_preds: torch.Tensor, img_h: int, img_w: int, np_dtype=None):
    """`format_to_numpy` of every image of a device batch: one threshold + NMS launch, one copy; returns a list of
    (8 + C) x N_b arrays (yogo/utils/prediction_formatting.py:96-156, call site yogo/infer.py:360-380)"""
    import numpy as np

    np_dtype = np.float32 if np_dtype is None else np_dtype
    rows, _, counts = format_preds_batched(batch_preds, box_format="xyxy")
    host = rows.cpu()
    return [_rows_xyxy_to_numpy(i, host[b, :n], img_h, img_w, np_dtype)
            for b, (i, n) in enumerate(zip(img_ids, counts.cpu().tolist()))]


def format_to_numpy(img_id: int, prediction_tensor, img_h: int, img_w: int, np_dtype=None):
    """reference signature (one image, numpy in / numpy out); the threshold + NMS runs on the MI355X"""
    t = torch.from_numpy(prediction_tensor) if not isinstance(prediction_tensor, torch.Tensor) else prediction_tensor
    if t.ndim != 3:
        raise ValueError(f"argument to format_pred should be unbatched result - shape should be (pred_shape, Sy, Sx), got {t.shape}")
    return format_to_numpy_batched([img_id], t.unsqueeze(0).cuda(), img_h, img_w, np_dtype)[0]


# ---------------------------------------------------------------------------------------------------------------------
# prediction <-> label matching for the metrics (yogo/utils/prediction_formatting.py:165-330)
# ---------------------------------------------------------------------------------------------------------------------
from dataclasses import dataclass  # noqa: E402


def _one_hot(idx: int, num_classes: int) -> torch.Tensor:
    return torch.nn.functional.one_hot(torch.tensor(idx, dtype=torch.long), num_classes=num_classes)


def _box_iou(b1: torch.Tensor, b2: torch.Tensor) -> torch.Tensor:
    """pairwise IoU of xyxy boxes (torchvision.ops.box_iou's published algorithm)"""
    a1 = (b1[:, 2] - b1[:, 0]) * (b1[:, 3] - b1[:, 1])
    a2 = (b2[:, 2] - b2[:, 0]) * (b2[:, 3] - b2[:, 1])
    lt = torch.max(b1[:, None, :2], b2[:, :2])
    rb = torch.min(b1[:, None, 2:], b2[:, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[:, :, 0] * wh[:, :, 1]
    return inter / (a1[:, None] + a2 - inter)


@dataclass
class PredictionLabelMatch:
    """one-to-one matches plus the missed labels and the extra (background) predictions -- the reference's dataclass
    (prediction_formatting.py:165-252), same fields and methods"""

    preds: torch.Tensor
    labels: torch.Tensor
    missed_labels: Optional[torch.Tensor]
    extra_predictions: Optional[torch.Tensor]

    @staticmethod
    def concat(preds_and_labels: List["PredictionLabelMatch"]) -> "PredictionLabelMatch":
        missed = [p.missed_labels for p in preds_and_labels if p.missed_labels is not None]
        extra = [p.extra_predictions for p in preds_and_labels if p.extra_predictions is not None]
        return PredictionLabelMatch(
            preds=torch.cat([p.preds for p in preds_and_labels]),
            labels=torch.cat([p.labels for p in preds_and_labels]),
            missed_labels=torch.cat(missed, dim=0) if missed else None,
            extra_predictions=torch.cat(extra, dim=0) if extra else None,
        )

    def convert_background_errors(self, num_classes: int) -> "PredictionLabelMatch":
        """missed labels become predictions of the (last) background class, extra predictions get a background label"""
        new_preds, new_labels = [], []
        for ml in ([] if self.missed_labels is None else self.missed_labels.tolist()):
            new_preds.append(torch.tensor([*ml[1:5], 1, *_one_hot(num_classes - 1, num_classes).float()]))
            new_labels.append(torch.tensor(ml))
        for ep in ([] if self.extra_predictions is None else self.extra_predictions.tolist()):
            new_preds.append(torch.tensor([*ep, 0]))
            new_labels.append(torch.tensor([1, *ep[:4], num_classes - 1]))
        new_preds_ten = torch.stack(new_preds).to(self.preds.device)
        new_labels_ten = torch.stack(new_labels).to(self.labels.device)
        self.preds = torch.cat([self.preds, torch.zeros(self.preds.shape[0], 1, device=self.preds.device)], dim=1)
        return PredictionLabelMatch(preds=torch.cat([self.preds, new_preds_ten]), labels=torch.cat([self.labels, new_labels_ten]),
                                    missed_labels=None, extra_predictions=None)


def _match_rows_to_labels(rows_xyxy: torch.Tensor, label: torch.Tensor) -> PredictionLabelMatch:
    """host side of format_preds_and_labels_v2: Hungarian assignment on 1 - IoU between the label boxes and the kept rows"""
    from scipy.optimize import linear_sum_assignment

    L, Sy, Sx = label.shape
    labels = label.reshape(L, Sx * Sy).T
    fl = labels[labels[:, 0].bool()]
    M, N = rows_xyxy.shape[0], fl.shape[0]
    cost = 1 - _box_iou(fl[:, 1:5], rows_xyxy[:, :4]).cpu().numpy()
    r, c = linear_sum_assignment(cost)
    rs, cs = set(r.tolist()), set(c.tolist())
    un_p = torch.tensor([i for i in range(M) if i not in cs], dtype=torch.long, device=rows_xyxy.device)
    un_l = torch.tensor([i for i in range(N) if i not in rs], dtype=torch.long, device=fl.device)
    return PredictionLabelMatch(preds=rows_xyxy[torch.tensor(c, dtype=torch.long)], labels=fl[torch.tensor(r, dtype=torch.long)],
                                missed_labels=fl[un_l], extra_predictions=rows_xyxy[un_p])


def format_preds_and_labels_v2_batched(preds: torch.Tensor, labels: torch.Tensor, objectness_thresh: float = 0.5,
                                       min_class_confidence_threshold: float = 0.0) -> List[PredictionLabelMatch]:
    """the matching of every image of a batch: ONE threshold + NMS launch and one device->host copy, then the assignment per
    image on the host (scipy, as the reference).  preds [B, 5+C, Sy, Sx] on the MI355X, labels [B, 6, Sy, Sx] anywhere."""
    rows, _, counts = format_preds_batched(preds, objectness_thresh, 0.5, "xyxy", min_class_confidence_threshold)
    host, lab = rows.cpu(), labels.detach().cpu()
    return [_match_rows_to_labels(host[b, :n], lab[b]) for b, n in enumerate(counts.cpu().tolist())]


def format_preds_and_labels_v2(pred: torch.Tensor, label: torch.Tensor, objectness_thresh: float = 0.5,
                               min_class_confidence_threshold: float = 0.0) -> PredictionLabelMatch:
    """reference signature (one image; prediction_formatting.py:254-330).  Results live on the host."""
    pred = pred.squeeze()
    label = label.squeeze()
    if len(pred.shape) != 3:
        raise ValueError(f"argument to format_pred should be unbatched result - shape should be (pred_shape, Sy, Sx), got {pred.shape}")
    return format_preds_and_labels_v2_batched(pred.unsqueeze(0), label.unsqueeze(0), objectness_thresh, min_class_confidence_threshold)[0]
