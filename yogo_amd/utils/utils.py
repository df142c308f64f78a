"""Small helpers behind the drivers (yogo/utils/utils.py): box drawing for `yogo infer --draw-boxes` / the validation image,
free-port lookup, device choice.  Drawing is host work on top of ONE batched threshold + NMS launch."""
from __future__ import annotations

import colorsys
import socket
import time
from typing import List, Optional, Tuple

import torch

from yogo_amd.utils.prediction_formatting import format_preds


class Timer:
    """yogo/utils/utils.py:29-47"""

    def __init__(self, name: str = "", precision: int = 5, post_print: bool = False):
        self.name, self.precision, self.post_print = name, precision, post_print

    def __enter__(self):
        self.start = time.perf_counter()
        return self

    def __exit__(self, *args):
        self.elapsed = time.perf_counter() - self.start
        if self.post_print:
            print(f"{self.name}: {self.elapsed:.{self.precision}f} s")


def get_free_port() -> int:
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    return port


def choose_device() -> torch.device:
    """the hot path is HIP-only: an MI355X or nothing (the reference falls back to mps / cpu, yogo/utils/utils.py:258-264)"""
    if torch.cuda.is_available():
        return torch.device("cuda")
    raise RuntimeError("yogo_amd needs an MI355X (torch.cuda.is_available() is False); there is no CPU compute path")


def _format_tensor_for_rects(rects: torch.Tensor, img_h: int, img_w: int, obj_thresh: float = 0.5, iou_thresh: float = 0.5,
                             min_class_confidence_threshold: float = 0.0) -> torch.Tensor:
    """[N, 6] = (x1, y1, x2, y2 in pixels, class, objectness) of the kept predictions (yogo/utils/utils.py:143-167)"""
    fp = format_preds(rects, obj_thresh=obj_thresh, iou_thresh=iou_thresh, box_format="xyxy",
                      min_class_confidence_threshold=min_class_confidence_threshold)
    out = torch.zeros((fp.shape[0], 6), device=fp.device)
    out[:, (0, 2)] = img_w * fp[:, (0, 2)]
    out[:, (1, 3)] = img_h * fp[:, (1, 3)]
    if fp.shape[0]:
        out[:, 4] = torch.argmax(fp[:, 5:], dim=1)
    out[:, 5] = fp[:, 4]
    return out


def bbox_colour(label_index: int, num_classes: int) -> Tuple[int, int, int, int]:
    hue = (label_index / num_classes * (5 / 3)) % 1
    r, g, b = colorsys.hls_to_rgb(hue, 0.5, 1.0)
    return int(r * 255), int(g * 255), int(b * 255), 255


def draw_yogo_prediction(img: torch.Tensor, prediction: torch.Tensor, obj_thresh: float = 0.5, iou_thresh: float = 0.5,
                         min_class_confidence_threshold: float = 0.0, labels: Optional[List[str]] = None,
                         images_are_normalized: bool = False):
    """PIL RGBA image with the predicted boxes drawn on it (yogo/utils/utils.py:183-255)"""
    import PIL.Image
    import PIL.ImageDraw

    img, prediction = img.clone().squeeze(), prediction.clone().squeeze()
    if images_are_normalized:
        img = img * 255
    img = img.to(torch.uint8)
    if img.ndim not in (2, 3) or (img.ndim == 3 and img.shape[0] not in (1, 3)):
        raise ValueError("img must be 2-dimensional (i.e. grayscale), or 3-dimensional (1 or three input channels) "
                         f"but has {img.ndim} dimensions")
    if img.ndim == 2:
        img = img[None, ...]
    if prediction.ndim != 3:
        raise ValueError("prediction must be 'unbatched' (i.e. shape (pred_dim, Sy, Sx) or (1, pred_dim, Sy, Sx)) - "
                         f"got shape {prediction.shape} ")
    num_channels, img_h, img_w = img.shape
    rects = _format_tensor_for_rects(prediction, img_h=img_h, img_w=img_w, obj_thresh=obj_thresh, iou_thresh=iou_thresh,
                                     min_class_confidence_threshold=min_class_confidence_threshold).cpu()
    arr = img.cpu().numpy()
    pil_img = PIL.Image.fromarray(arr[0], mode="L") if num_channels == 1 else PIL.Image.fromarray(arr.transpose(1, 2, 0), mode="RGB")
    rgb = PIL.Image.new("RGBA", pil_img.size)
    rgb.paste(pil_img)
    draw = PIL.ImageDraw.Draw(rgb)
    for r in rects.tolist():
        label_idx = int(r[4])
        label = labels[label_idx] if labels is not None else str(label_idx)
        # (the reference passes the image's channel count - 5 as the class count here, utils.py:246; kept)
        draw.rectangle(r[:4], outline=bbox_colour(label_idx, num_classes=num_channels - 5))
        draw.text((r[0], r[1]), label, (0, 0, 0, 255))
    return rgb
