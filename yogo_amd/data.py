"""The step in front of the hot path: label files -> label tensors, and the training augmentations.

Host-side mirror of the reference's interface for this step -- same names, arguments and error behaviour:

* ``load_labels`` / ``correct_label_idx`` / ``label_file_to_tensor`` / ``format_labels_tensor`` -- yogo/data/yogo_dataset.py:24-133
* ``RandomHorizontalFlipWithBBs`` / ``RandomVerticalFlipWithBBs`` / ``MultiArgSequential`` / ``DualInputId`` /
  ``ImageTransformLabelIdentity`` -- yogo/data/data_transforms.py:17-98
* ``collate_batch_robust`` -- yogo/data/utils.py:49-63

The reference rasterises one image at a time in Python inside DataLoader workers and flips batches with ATen on the CPU.
Here the parsing stays on the host (file I/O), while the rasteriser and the flips run as HIP kernels on device tensors
(``yogo_labels_rasterize``, ``yogo_flip_batch``): ``format_labels_batch`` rasterises a whole batch with one launch, and a
``MultiArgSequential`` holding the two flips fuses them into one pass.  There is no CPU fallback: tensors must be on the GPU.
"""
from __future__ import annotations

import csv
from pathlib import Path
from typing import Any, Dict, List, Optional, Sequence, Tuple, Union

import torch

from . import _hip

LABEL_TENSOR_PRED_DIM_SIZE = 1 + 4 + 1
# yogo/data/yogo_dataset.py:21 (hard-coded for 772 x 1032 images in the reference)
AREA_FILTER_THRESHOLD = 200 / (772 * 1032)


def correct_label_idx(label: str, classes: List[str], notes_data: Optional[Dict[str, Any]] = None) -> int:
    """yogo/data/yogo_dataset.py:49-69: class index of a label-file entry (numeric id, LabelStudio id via notes.json, or name)."""
    if notes_data is None:
        return int(label)
    if label.isnumeric():
        label_name: Optional[str] = None
        for row in notes_data["categories"]:
            if int(label) == int(row["id"]):
                label_name = row["name"]
                break
        if label_name is None:
            raise ValueError(f"label index {label} not found in notes.json file")
        return classes.index(label_name)
    return classes.index(label)


def load_labels(label_path: Union[str, Path], classes: List[str], notes_data: Optional[Dict[str, Any]] = None) -> List[List[float]]:
    """yogo/data/yogo_dataset.py:72-110: rows [class, xc, yc, w, h] of a (csv-sniffed) YOLO label file; boxes smaller than
    AREA_FILTER_THRESHOLD are dropped; an empty file gives []."""
    labels: List[List[float]] = []
    with open(label_path, "r") as f:
        file_chunk = f.read(1024)
        f.seek(0)
        try:
            dialect = csv.Sniffer().sniff(file_chunk)
            has_header = csv.Sniffer().has_header(file_chunk)
            reader = csv.reader(f, dialect)
        except csv.Error:
            return []
        if has_header:
            next(reader, None)
        for row in reader:
            assert len(row) == 5, f"should have [class,xc,yc,w,h] - got length {len(row)} {row}"
            xc, yc, w, h = map(float, row[1:])
            if w * h < AREA_FILTER_THRESHOLD:
                continue
            labels.append([float(correct_label_idx(row[0], classes, notes_data)), xc, yc, w, h])
    return labels


def _device_of(device) -> torch.device:
    dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    if dev.type != "cuda":
        raise RuntimeError(f"yogo_amd: the label rasteriser runs on an MI355X device (got {dev}); there is no CPU fallback")
    return dev


def format_labels_batch(labels: Sequence[torch.Tensor], Sx: int, Sy: int, box_format: str = "xyxy", device=None) -> torch.Tensor:
    """Rasterise the label rows of a whole batch with one launch: labels[b] is [N_b, 5] (class, x1, y1, x2, y2) -- or
    (class, xc, yc, w, h) with box_format="cxcywh" -- host or device; returns [B, 6, Sy, Sx] fp32 on the device.
    Raises IndexError (like the reference's indexing) when a box centre falls outside the grid."""
    if box_format not in ("xyxy", "cxcywh"):
        raise ValueError(f"box_format must be 'xyxy' or 'cxcywh', got {box_format}")
    B = len(labels)
    dev = _device_of(device if device is not None else next((t.device for t in labels if t.is_cuda), None))
    counts = [int(t.shape[0]) if t.numel() else 0 for t in labels]
    for t in labels:
        if t.numel() and (t.ndim != 2 or t.shape[1] != 5):
            raise ValueError(f"labels must have shape (N, 5), got {tuple(t.shape)}")
    offsets = torch.zeros(B + 1, dtype=torch.int32)
    if B:
        offsets[1:] = torch.tensor(counts, dtype=torch.int32).cumsum(0)
    rows = [t.reshape(-1, 5).to(dtype=torch.float32) for t in labels if t.numel()]
    with torch.cuda.device(dev):
        flat = torch.cat([r.to(dev, non_blocking=True) for r in rows]) if rows else torch.zeros(0, 5, device=dev)
        out = torch.empty(B, LABEL_TENSOR_PRED_DIM_SIZE, Sy, Sx, dtype=torch.float32, device=dev)
        if B == 0:
            return out
        status = torch.zeros(1, dtype=torch.int32, device=dev)
        _hip.call("yogo_labels_rasterize", flat.contiguous(), offsets.to(dev), out, status, B, Sx, Sy,
                  1 if box_format == "cxcywh" else 0, _hip.stream_ptr())
        bad = int(status.item())
    if bad:
        raise IndexError(f"label row {bad - 1}: box centre outside the {Sx} x {Sy} grid")
    return out


def format_labels_tensor(labels: torch.Tensor, Sx: int, Sy: int) -> torch.Tensor:
    """yogo/data/yogo_dataset.py:24-46: (N, 5) rows (class, x1, y1, x2, y2) -> (6, Sy, Sx) = (mask, x1, y1, x2, y2, class)."""
    return format_labels_batch([labels], Sx, Sy, device=labels.device if labels.is_cuda else None)[0]


def label_file_to_tensor(label_path: Union[str, Path], Sx: int, Sy: int, classes: List[str],
                         notes_data: Optional[Dict[str, Any]] = None, device=None) -> torch.Tensor:
    """yogo/data/yogo_dataset.py:113-133."""
    try:
        labels = load_labels(label_path, classes=classes, notes_data=notes_data)
    except Exception as e:
        raise RuntimeError(f"exception from {label_path}") from e
    return format_labels_batch([torch.tensor(labels, dtype=torch.float32).reshape(-1, 5)], Sx, Sy, "cxcywh", device)[0]


def label_files_to_batch(label_paths: Sequence[Union[str, Path]], Sx: int, Sy: int, classes: List[str],
                         notes_data: Optional[Dict[str, Any]] = None, device=None) -> torch.Tensor:
    """label_file_to_tensor for a list of files: parse on the host, rasterise the batch with one launch."""
    rows = []
    for path in label_paths:
        try:
            rows.append(torch.tensor(load_labels(path, classes=classes, notes_data=notes_data), dtype=torch.float32).reshape(-1, 5))
        except Exception as e:
            raise RuntimeError(f"exception from {path}") from e
    return format_labels_batch(rows, Sx, Sy, "cxcywh", device)


# ---------------------------------------------------------------------------------------------------------------------------
# augmentations -- yogo/data/data_transforms.py
# ---------------------------------------------------------------------------------------------------------------------------
def flip_batch(img_batch: torch.Tensor, label_batch: torch.Tensor, hflip: bool, vflip: bool) -> Tuple[torch.Tensor, torch.Tensor]:
    """Both flips of a batch (images [B,C,H,W] uint8 / float32, labels [B,6,Sy,Sx] float32) in one pass over each tensor."""
    assert img_batch.ndim == 4 and label_batch.ndim == 4
    _hip.require_cuda(img_batch, "the image batch")
    _hip.require_cuda(label_batch, "the label batch")
    if not (hflip or vflip):
        return img_batch, label_batch
    if img_batch.dtype not in (torch.uint8, torch.float32) or label_batch.dtype != torch.float32 or label_batch.shape[1] != 6:
        raise ValueError("flip_batch: images must be uint8 or float32, labels float32 of shape (B, 6, Sy, Sx)")
    B, C, H, W = img_batch.shape
    if label_batch.shape[0] != B:
        raise ValueError("flip_batch: image and label batch sizes differ")
    img, lab = img_batch.contiguous(), label_batch.contiguous()
    img_out, lab_out = torch.empty_like(img), torch.empty_like(lab)
    with torch.cuda.device(img.device):
        _hip.call("yogo_flip_batch", img, img_out, img.element_size(), lab, lab_out, B, C, H, W, lab.shape[2], lab.shape[3],
                  1 if hflip else 0, 1 if vflip else 0, _hip.stream_ptr())
    return img_out, lab_out


class DualInputModule(torch.nn.Module):
    def forward(self, inpt_a, inpt_b): ...


class DualInputId(DualInputModule):
    def forward(self, img_batch, labels):
        return img_batch, labels


class ImageTransformLabelIdentity(DualInputModule):
    """data_transforms.py:40-48: a transform of the images that leaves the (normalised) labels alone."""

    def __init__(self, transform):
        super().__init__()
        self.transform = transform

    def forward(self, img_batch, labels):
        return self.transform(img_batch), labels


class RandomHorizontalFlipWithBBs(DualInputModule):
    """data_transforms.py:51-74: with probability p flip the whole batch left-right, labels included."""

    def __init__(self, p=0.5):
        super().__init__()
        self.p = p

    def draw(self) -> bool:
        return bool(torch.rand(1) < self.p)   # one host draw per batch, like the reference

    def forward(self, img_batch: torch.Tensor, label_batch: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        return flip_batch(img_batch, label_batch, self.draw(), False)


class RandomVerticalFlipWithBBs(DualInputModule):
    """data_transforms.py:77-98."""

    def __init__(self, p=0.5):
        super().__init__()
        self.p = p

    def draw(self) -> bool:
        return bool(torch.rand(1) < self.p)

    def forward(self, img_batch: torch.Tensor, label_batch: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        return flip_batch(img_batch, label_batch, False, self.draw())


class MultiArgSequential(torch.nn.Sequential):
    """data_transforms.py:27-37.  A horizontal flip directly followed by a vertical one (the reference's training
    augmentation, yogo/data/yogo_dataloader.py:203-210) runs as ONE pass; the random draws happen in the reference's order."""

    def __init__(self, *args: DualInputModule, **kwargs):
        super().__init__(*[t for t in args if not isinstance(t, DualInputId)], **kwargs)

    def forward(self, *input):
        mods = list(self)
        k = 0
        while k < len(mods):
            m = mods[k]
            if (isinstance(m, RandomHorizontalFlipWithBBs) and k + 1 < len(mods) and isinstance(mods[k + 1], RandomVerticalFlipWithBBs)
                    and input[0].ndim == 4 and input[1].ndim == 4):
                h = m.draw()
                v = mods[k + 1].draw()
                input = flip_batch(input[0], input[1], h, v)
                k += 2
                continue
            input = m(*input)
            k += 1
        return input


def collate_batch_robust(batch: List[Optional[Tuple[torch.Tensor, torch.Tensor]]],
                         transforms: MultiArgSequential = MultiArgSequential(), device=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """yogo/data/utils.py:49-63: drop None items, stack, move to the device, apply the (device-side) transforms."""
    inputs, labels = zip(*[pair for pair in batch if pair is not None])
    dev = _device_of(device if device is not None else (inputs[0].device if inputs[0].is_cuda else None))
    batched_inputs = torch.stack(inputs).to(dev, non_blocking=True)
    batched_labels = torch.stack(labels).to(dev, non_blocking=True)
    return transforms(batched_inputs, batched_labels)
