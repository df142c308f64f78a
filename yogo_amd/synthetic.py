"""Synthetic inputs of the shape the reference trains on: uint8 772x1032 grayscale images
(``torch.randint(0, 256)``, the distribution the reference itself uses for dummy inputs: yogo/infer.py:233) and label
tensors [B, 6, Sy, Sx] = (mask, x1, y1, x2, y2, class) rasterised the way yogo/data/yogo_dataset.py:24-46 does:
cell (i, j) = (floor((x1+x2)*Sx/2), floor((y1+y2)*Sy/2)).  Generated directly on the device."""
from __future__ import annotations

import torch


def synthetic_images(B: int, H: int = 772, W: int = 1032, channels: int = 1, device="cuda", seed: int = 0) -> torch.Tensor:
    g = torch.Generator(device=device).manual_seed(seed)
    return torch.randint(0, 256, (B, channels, H, W), dtype=torch.uint8, device=device, generator=g)


def synthetic_labels(B: int, Sx: int, Sy: int, K: int = 64, num_classes: int = 7, anchor_w: float = 0.0425,
                     anchor_h: float = 0.0555, device="cuda", seed: int = 1) -> torch.Tensor:
    g = torch.Generator(device=device).manual_seed(seed)
    c = torch.rand(B, K, 2, device=device, generator=g) * 0.9 + 0.05
    w = anchor_w * torch.exp(torch.randn(B, K, device=device, generator=g) * 0.2)
    h = anchor_h * torch.exp(torch.randn(B, K, device=device, generator=g) * 0.2)
    cls = torch.randint(0, num_classes, (B, K), device=device, generator=g).float()
    x1, y1, x2, y2 = c[..., 0] - w / 2, c[..., 1] - h / 2, c[..., 0] + w / 2, c[..., 1] + h / 2
    ii = torch.div((x1 + x2) * Sx, 2, rounding_mode="floor").long().clamp(0, Sx - 1)
    jj = torch.div((y1 + y2) * Sy, 2, rounding_mode="floor").long().clamp(0, Sy - 1)
    out = torch.zeros(B, 6, Sy * Sx, device=device)
    cell = jj * Sx + ii                                    # later boxes overwrite earlier ones, as the reference loop does
    vals = torch.stack((torch.ones_like(x1), x1, y1, x2, y2, cls), dim=1)   # [B, 6, K]
    for k in range(K):
        out.scatter_(2, cell[:, None, k : k + 1].expand(B, 6, 1), vals[:, :, k : k + 1])
    return out.view(B, 6, Sy, Sx)
