#!/usr/bin/env python3
"""Golden vectors for the step in front of the hot path, from the REAL reference functions:

* ``format_labels_tensor`` / ``load_labels`` / ``label_file_to_tensor``  (yogo/data/yogo_dataset.py:24-133)
* ``RandomHorizontalFlipWithBBs`` / ``RandomVerticalFlipWithBBs``        (yogo/data/data_transforms.py:51-98)

Run only in the build container (``/root/reference`` does not exist on the GPU box):  python tests/golden/make_golden_data.py

The two reference modules are pure torch apart from their imports of ``torchvision.{ops,datasets,transforms}`` and
``yogo.data.utils`` (ruamel / zarr readers), none of which is installable here.  They are imported on top of build-owned stubs:
``torchvision.ops`` = the oracle's restatement (as make_golden.py), ``torchvision.datasets`` / ``transforms.Resize`` = empty
placeholders (the functions exercised never touch them), ``torchvision.transforms.functional.hflip / vflip`` = ``tensor.flip(-1)`` /
``tensor.flip(-2)`` (what torchvision's tensor backend does), ``yogo.data.utils.read_image_robust`` = a placeholder.  So the
control flow pinned here is the reference's own: the cell index arithmetic ((x1 + x2) * Sx // 2 in float32, ``.int()``), "a later
row overwrites the cell", Python's negative-index wrap, the IndexError beyond the grid, the csv sniffing / header skip / area
filter of the label files, the order of the coordinate swap and the mirror in the flips.

Only data is written: inputs and expected outputs.  No reference source is copied.
"""
import importlib
import json
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import yogo_oracle as O  # noqa: E402


def import_reference_data():
    for sub in ("yogo", "yogo.data", "yogo.utils"):
        m = types.ModuleType(sub)
        m.__path__ = [os.path.join(REF, sub.replace(".", "/"))]
        sys.modules[sub] = m
    tv = types.ModuleType("torchvision")
    ops = types.ModuleType("torchvision.ops")
    ops.box_convert, ops.complete_box_iou_loss, ops.nms, ops.box_iou = O.box_convert, O.complete_box_iou_loss, O.nms, O.box_iou
    ds = types.ModuleType("torchvision.datasets")
    ds.VisionDataset = type("VisionDataset", (torch.utils.data.Dataset,), {"__init__": lambda self, *a, **k: None})
    tr = types.ModuleType("torchvision.transforms")
    tr.Resize = type("Resize", (torch.nn.Module,), {"__init__": lambda self, *a, **k: torch.nn.Module.__init__(self)})
    fn = types.ModuleType("torchvision.transforms.functional")
    fn.hflip = lambda t: t.flip(-1)
    fn.vflip = lambda t: t.flip(-2)
    tr.functional = fn
    tv.ops, tv.datasets, tv.transforms = ops, ds, tr
    for name, mod in (("torchvision", tv), ("torchvision.ops", ops), ("torchvision.datasets", ds), ("torchvision.transforms", tr),
                      ("torchvision.transforms.functional", fn)):
        sys.modules[name] = mod
    du = types.ModuleType("yogo.data.utils")
    du.read_image_robust = lambda *a, **k: None
    sys.modules["yogo.data.utils"] = du
    yd = importlib.import_module("yogo.data.yogo_dataset")
    dt = importlib.import_module("yogo.data.data_transforms")
    return yd, dt


def main():
    yd, dt = import_reference_data()
    out = {}
    meta = {"rast": [], "files": [], "flips": []}
    g = torch.Generator().manual_seed(2024)

    def rows(n, C=7):
        c = torch.rand(n, 2, generator=g) * 0.96 + 0.02
        w = 0.0425 * torch.exp(torch.randn(n, generator=g) * 0.2)
        h = 0.0555 * torch.exp(torch.randn(n, generator=g) * 0.2)
        cls = torch.randint(0, C, (n,), generator=g).float()
        return torch.stack((cls, c[:, 0] - w / 2, c[:, 1] - h / 2, c[:, 0] + w / 2, c[:, 1] + h / 2), dim=1)

    # ---- format_labels_tensor: plain, many rows per cell (the later row wins), negative-index wrap, out of the grid ------------
    cases = [("plain", 129, 97, rows(64)), ("dense", 9, 7, rows(400)), ("one", 33, 25, rows(1)),
             ("two_in_one_cell", 5, 3, torch.tensor([[1.0, 0.41, 0.40, 0.45, 0.46], [4.0, 0.43, 0.41, 0.47, 0.49]])),
             ("negative_wrap", 33, 25, torch.tensor([[2.0, -0.2 / 33, 0.4, 0.1 / 33, 0.6], [3.0, 0.5, -0.3 / 25, 0.6, 0.1 / 25]])),
             ("edge_exact", 4, 4, torch.tensor([[0.0, 0.25, 0.25, 0.25, 0.25], [1.0, 0.4999999, 0.5, 0.5, 0.5000001]]))]
    for name, Sx, Sy, r in cases:
        out[f"rast/{name}/rows"] = r
        out[f"rast/{name}/out"] = yd.format_labels_tensor(r.clone(), Sx, Sy)
        meta["rast"].append({"name": name, "Sx": Sx, "Sy": Sy})
    errs = []
    for name, Sx, Sy, r in [("beyond_right", 33, 25, torch.tensor([[0.0, 0.9, 0.4, 1.1, 0.6]])),
                            ("beyond_bottom", 33, 25, torch.tensor([[0.0, 0.4, 0.95, 0.6, 1.2]])),
                            ("far_negative", 33, 25, torch.tensor([[0.0, -1.2, 0.4, -0.9, 0.6]]))]:
        try:
            yd.format_labels_tensor(r.clone(), Sx, Sy)
            kind = "none"
        except IndexError:
            kind = "IndexError"
        out[f"rast_err/{name}/rows"] = r
        errs.append({"name": name, "Sx": Sx, "Sy": Sy, "raises": kind})
    meta["rast_err"] = errs

    # ---- label files: csv sniffing, header row, the area filter (200 px^2 of a 772 x 1032 image), empty file ----------------------
    classes = ["healthy", "ring", "troph", "schizont", "gametocyte", "wbc", "misc"]
    files = {
        "plain.txt": "0 0.5 0.5 0.05 0.06\n3 0.25 0.75 0.04 0.05\n6 0.9 0.1 0.03 0.07\n",
        "comma.csv": "1,0.31,0.42,0.05,0.05\n2,0.61,0.22,0.06,0.04\n5,0.11,0.92,0.05,0.05\n",
        "header.csv": "class,xc,yc,w,h\n1,0.31,0.42,0.05,0.05\n2,0.61,0.22,0.06,0.04\n4,0.71,0.32,0.06,0.04\n",
        "tiny_boxes.txt": "0 0.5 0.5 0.05 0.06\n1 0.3 0.3 0.01 0.02\n2 0.7 0.7 0.0158 0.0158\n3 0.2 0.8 0.0159 0.0159\n",
        "empty.txt": "",
        "same_cell.txt": "0 0.500 0.500 0.05 0.06\n5 0.501 0.501 0.04 0.05\n",
    }
    with tempfile.TemporaryDirectory() as td:
        for fname, text in files.items():
            path = os.path.join(td, fname)
            open(path, "w").write(text)
            for (Sx, Sy) in ((129, 97), (33, 25)):
                t = yd.label_file_to_tensor(path, Sx, Sy, classes)
                out[f"file/{fname}/{Sx}x{Sy}"] = t
            lab = yd.load_labels(path, classes)
            out[f"file/{fname}/rows"] = np.asarray(lab, dtype=np.float64).reshape(-1, 5)
            meta["files"].append({"name": fname, "text": text})
    meta["classes"] = classes

    # ---- flips: forced (p = 1) and skipped (p = 0), images uint8 and float, odd sizes, empty cells included --------------------------
    for name, B, Cc, H, W, Sx, Sy, dtype in [("u8", 3, 1, 12, 20, 5, 3, torch.uint8), ("f32_rgb", 2, 3, 7, 9, 4, 3, torch.float32),
                                             ("odd", 1, 1, 5, 17, 7, 2, torch.uint8)]:
        img = torch.randint(0, 256, (B, Cc, H, W), generator=g).to(dtype)
        lab = torch.zeros(B, 6, Sy, Sx)
        for b in range(B):
            lab[b] = yd.format_labels_tensor(rows(6), Sx, Sy)
        out[f"flip/{name}/img"], out[f"flip/{name}/lab"] = img, lab
        for tag, mod in (("h", dt.RandomHorizontalFlipWithBBs(1.0)), ("v", dt.RandomVerticalFlipWithBBs(1.0)),
                         ("h0", dt.RandomHorizontalFlipWithBBs(0.0)), ("v0", dt.RandomVerticalFlipWithBBs(0.0))):
            oi, ol = mod(img.clone(), lab.clone())
            out[f"flip/{name}/{tag}/img"], out[f"flip/{name}/{tag}/lab"] = oi, ol
        hv = dt.MultiArgSequential(dt.RandomHorizontalFlipWithBBs(1.0), dt.RandomVerticalFlipWithBBs(1.0))
        oi, ol = hv(img.clone(), lab.clone())
        out[f"flip/{name}/hv/img"], out[f"flip/{name}/hv/lab"] = oi, ol
        meta["flips"].append({"name": name})
    payload = {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else v) for k, v in out.items()}
    payload["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(HERE, "data_step.npz"), **payload)
    print("wrote data_step.npz", len(payload), "arrays")


if __name__ == "__main__":
    main()
