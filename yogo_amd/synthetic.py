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


def synthetic_predictions(B: int, Sx: int, Sy: int, num_classes: int = 7, K: int = 100, device="cuda", seed: int = 2) -> torch.Tensor:
    """'realistic' post-process input: K objects per image, each predicted by 1-4 neighbouring cells
    with 5 %-jittered boxes, objectness U(0.5, 1), soft-maxed class logits; every other cell has objectness U(0, 0.4)."""
    g = torch.Generator(device=device).manual_seed(seed)
    P = 5 + num_classes
    out = torch.zeros(B, P, Sy, Sx, device=device)
    out[:, 4] = torch.rand(B, Sy, Sx, device=device, generator=g) * 0.4
    out[:, 0] = (torch.arange(Sx, device=device).float()[None, None, :] + 0.5) / Sx
    out[:, 1] = (torch.arange(Sy, device=device).float()[None, :, None] + 0.5) / Sy
    out[:, 2] = 0.0425
    out[:, 3] = 0.0555
    out[:, 5:] = torch.softmax(torch.randn(B, num_classes, Sy, Sx, device=device, generator=g), dim=1)
    cx = torch.rand(B, K, device=device, generator=g) * 0.9 + 0.05
    cy = torch.rand(B, K, device=device, generator=g) * 0.9 + 0.05
    w = 0.0425 * torch.exp(torch.randn(B, K, device=device, generator=g) * 0.2)
    h = 0.0555 * torch.exp(torch.randn(B, K, device=device, generator=g) * 0.2)
    ncell = torch.randint(1, 5, (B, K), device=device, generator=g)
    logits = torch.randn(B, K, num_classes, device=device, generator=g) * 2
    flat = out.view(B, P, Sy * Sx)
    i0 = (cx * Sx).long().clamp(0, Sx - 1)
    j0 = (cy * Sy).long().clamp(0, Sy - 1)
    for n, (di, dj) in enumerate([(0, 0), (1, 0), (0, 1), (1, 1)]):
        use = (ncell > n)
        i = (i0 + di).clamp(max=Sx - 1)
        j = (j0 + dj).clamp(max=Sy - 1)
        cell = j * Sx + i
        jit = 0.05 * torch.randn(B, K, 4, device=device, generator=g)
        vals = torch.stack((cx + w * jit[..., 0], cy + h * jit[..., 1], w * (1 + jit[..., 2]), h * (1 + jit[..., 3]),
                            0.5 + 0.5 * torch.rand(B, K, device=device, generator=g)), dim=1)
        cls = torch.softmax(logits + 0.3 * torch.randn(B, K, num_classes, device=device, generator=g), dim=2).permute(0, 2, 1)
        vals = torch.cat((vals, cls), dim=1)                       # [B, P, K]
        cur = torch.gather(flat, 2, cell[:, None, :].expand(B, P, K))
        vals = torch.where(use[:, None, :], vals, cur)
        flat.scatter_(2, cell[:, None, :].expand(B, P, K), vals)
    return out


def synthetic_dense_predictions(B: int, Sx: int, Sy: int, num_classes: int = 7, frac: float = 0.93, device="cuda", seed: int = 3) -> torch.Tensor:
    """'dense' post-process input (what a random-init network produces -- ~93 % of the cells pass the
    objectness threshold, the NMS worst case): every cell predicts a box near its own centre with anchor-sized extents, so each
    box overlaps dozens of neighbours; objectness U(0.5, 1) on `frac` of the cells, U(0, 0.5) on the rest."""
    g = torch.Generator(device=device).manual_seed(seed)
    P = 5 + num_classes
    out = torch.empty(B, P, Sy, Sx, device=device)
    out[:, 0] = (torch.arange(Sx, device=device).float()[None, None, :] + torch.rand(B, Sy, Sx, device=device, generator=g)) / Sx
    out[:, 1] = (torch.arange(Sy, device=device).float()[None, :, None] + torch.rand(B, Sy, Sx, device=device, generator=g)) / Sy
    out[:, 2] = 0.0425 * torch.exp(torch.randn(B, Sy, Sx, device=device, generator=g) * 0.2)
    out[:, 3] = 0.0555 * torch.exp(torch.randn(B, Sy, Sx, device=device, generator=g) * 0.2)
    fire = torch.rand(B, Sy, Sx, device=device, generator=g) < frac
    u = torch.rand(B, Sy, Sx, device=device, generator=g) * 0.5
    out[:, 4] = torch.where(fire, 0.5 + u, u)
    out[:, 5:] = torch.softmax(torch.randn(B, num_classes, Sy, Sx, device=device, generator=g) * 2, dim=1)
    return out


def raw_from_predictions(pred: torch.Tensor, cxs: torch.Tensor, cys: torch.Tensor, anchor_w: float, anchor_h: float) -> torch.Tensor:
    """a head output whose box decode (yogo/model.py:277-313, inference mode) gives `pred` back up to rounding: the input of the
    fused decode + threshold + NMS kernel for the 'realistic' / 'dense' post-process workloads above"""
    B, P, Sy, Sx = pred.shape
    eps = 1e-4
    raw = torch.empty_like(pred)
    raw[:, 0] = torch.logit(((pred[:, 0] - cxs) * Sx).clamp(eps, 1 - eps))
    raw[:, 1] = torch.logit(((pred[:, 1] - cys) * Sy).clamp(eps, 1 - eps))
    raw[:, 2] = torch.log(pred[:, 2] / anchor_w)
    raw[:, 3] = torch.log(pred[:, 3] / anchor_h)
    raw[:, 4] = torch.logit(pred[:, 4].clamp(eps, 1 - eps))
    raw[:, 5:] = torch.log(pred[:, 5:].clamp_min(1e-12))
    return raw
