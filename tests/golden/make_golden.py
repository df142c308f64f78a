#!/usr/bin/env python3
"""Generate the golden fixtures in this directory from the REAL reference.

Run only in the build container (``/root/reference`` does not exist on the GPU box):

    python tests/golden/make_golden.py

What is real and what is restated:
* ``yogo/model.py`` + ``yogo/model_defns.py`` import as they are (torch only) -> files ``net_*.npz``,
  ``grid.npz``, ``ckpt_keys.json`` come from the genuine reference code.
* ``yogo/yogo_loss.py`` and ``yogo/utils/prediction_formatting.py`` need ``torchvision.ops``, which is
  not installable here.  They are imported on top of a build-owned ``torchvision.ops`` module whose four
  functions are the restatement in ``oracle/yogo_oracle.py`` -> files ``loss_*.npz``, ``fmt_*.npz`` pin the
  reference's control flow (masking, weights, ordering, thresholds); the torchvision arithmetic inside
  stays a flagged restatement ("parity unpinned" for CIoU / NMS arithmetic).

Only data is written: inputs and expected outputs.  No reference source is copied.
"""
import importlib
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import yogo_oracle as O  # noqa: E402


def import_reference():
    for sub in ("yogo", "yogo.data", "yogo.utils"):
        m = types.ModuleType(sub)
        m.__path__ = [os.path.join(REF, sub.replace(".", "/"))]
        sys.modules[sub] = m
    tv = types.ModuleType("torchvision")
    ops = types.ModuleType("torchvision.ops")
    ops.box_convert = O.box_convert
    ops.complete_box_iou_loss = O.complete_box_iou_loss
    ops.nms = O.nms
    ops.box_iou = O.box_iou
    tv.ops = ops
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.ops"] = ops
    model = importlib.import_module("yogo.model")
    defns = importlib.import_module("yogo.model_defns")
    loss = importlib.import_module("yogo.yogo_loss")
    pf = importlib.import_module("yogo.utils.prediction_formatting")
    return model, defns, loss, pf


def npsave(name, **kw):
    out = {}
    for k, v in kw.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = v
    np.savez_compressed(os.path.join(HERE, name), **out)
    print("wrote", name, len(out), "arrays")


def trained_like_bn(net, seed=3):
    """random-init nets in eval() saturate the decode; give BN plausible stats."""
    g = torch.Generator().manual_seed(seed)
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.weight.data = 1 + 0.1 * torch.randn(m.num_features, generator=g)
            m.bias.data = 0.1 * torch.randn(m.num_features, generator=g)


def calibrate_bn(net, x):
    """one train-mode pass with momentum 1 so running stats == batch stats of a real input."""
    mods = [m for m in net.modules() if isinstance(m, torch.nn.BatchNorm2d)]
    for m in mods:
        m.momentum = 1.0
    drops = [m for m in net.modules() if isinstance(m, torch.nn.Dropout2d)]
    ps = [d.p for d in drops]
    for d in drops:
        d.p = 0.0
    net.train()
    with torch.no_grad():
        net(x)
    for m in mods:
        m.momentum = 0.1
        m.num_batches_tracked.zero_()
    for d, p in zip(drops, ps):
        d.p = p


def net_fixture(model, defns, name, model_name, img_hw, B, num_classes=7, is_rgb=False, save_state=True):
    torch.manual_seed(0)
    H, W = img_hw
    net = model.YOGO((H, W), 0.0425, 0.0555, num_classes, is_rgb=is_rgb, model_func=defns.get_model_func(model_name))
    trained_like_bn(net)
    g = torch.Generator().manual_seed(10)
    x = torch.randint(0, 256, (B, 3 if is_rgb else 1, H, W), dtype=torch.uint8, generator=g)
    calibrate_bn(net, x)
    sd0 = {k: v.clone() for k, v in net.state_dict().items()}

    # eval forward, inference False / True
    net.eval()
    net.inference = False
    with torch.no_grad():
        out_eval = net(x.clone())
        raw_eval = net.model(x.float())
    net.inference = True
    with torch.no_grad():
        out_inf = net(x.clone())
    net.inference = False

    # train-mode forward + backward with dropout forced to p=0 (RNG cannot be matched), clamp hooks live
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout2d):
            m.p = 0.0
    net.train()
    out_tr = net(x.clone())
    gu = torch.Generator().manual_seed(11)
    upstream = torch.randn(out_tr.shape, generator=gu) * 3e-3
    net.zero_grad()
    out_tr.backward(upstream)
    grads = {k: p.grad.clone() for k, p in net.named_parameters()}
    sd1 = {k: v.clone() for k, v in net.state_dict().items()}

    payload = dict(x=x, out_eval=out_eval, raw_eval=raw_eval, out_inf=out_inf, out_train=out_tr, upstream=upstream)
    for k, v in sd0.items():
        # save_state=False: conv weights are those of net_base_64x96.npz (same manual_seed(0) init draw)
        if save_state or v.ndim != 4:
            payload["sd/" + k] = v
    for k, v in grads.items():
        payload["grad/" + k] = v
    for k, v in sd1.items():
        if "running" in k or "num_batches" in k:
            payload["sd_after/" + k] = v
    payload["meta"] = np.array(json.dumps(dict(model=model_name, H=H, W=W, B=B, num_classes=num_classes, is_rgb=is_rgb,
                                                Sx=net.Sx, Sy=net.Sy, anchor_w=0.0425, anchor_h=0.0555)))
    npsave(name, **payload)
    return sd0


def main():
    model, defns, loss_mod, pf = import_reference()
    torch.set_num_threads(8)

    # ---- (ii) grid + checkpoint key set / dtypes ---------------------------------------------------
    torch.manual_seed(0)
    y = model.YOGO((772, 1032), 0.0425, 0.0555, 7)
    sizes = {}
    for hw in [(772, 1032), (193, 1032), (64, 96), (100, 131)]:
        sizes[f"{hw[0]}x{hw[1]}"] = list(y.get_grid_size(hw))
    npsave("grid.npz", Cxs=y._Cxs, Cys=y._Cys, sizes=np.array(json.dumps(sizes)))
    keys = {}
    for name in defns.MODELS:
        if name == "convnext_small":
            continue
        torch.manual_seed(0)
        n = model.YOGO((772, 1032), 0.0425, 0.0555, 7, model_func=defns.get_model_func(name))
        keys[name] = dict(
            keys=[[k, list(v.shape), str(v.dtype)] for k, v in n.state_dict().items()],
            num_params=n.num_params(), Sx=n.Sx, Sy=n.Sy,
        )
    with open(os.path.join(HERE, "ckpt_keys.json"), "w") as f:
        json.dump(keys, f)
    print("wrote ckpt_keys.json")

    # kaiming-init statistics of the reference init (model.py:79-87)
    stats = {}
    for k, v in y.state_dict().items():
        if k.endswith("weight") and v.ndim == 4:
            stats[k] = [float(v.mean()), float(v.std()), list(v.shape)]
    with open(os.path.join(HERE, "init_stats.json"), "w") as f:
        json.dump(stats, f)

    # ---- (i) backbone / decode / grads -------------------------------------------------------------
    net_fixture(model, defns, "net_base_64x96.npz", "base_model", (64, 96), 2)
    net_fixture(model, defns, "net_silu_64x96.npz", "silu_model", (64, 96), 2, save_state=False)
    net_fixture(model, defns, "net_quarter_rgb_50x70.npz", "quarter_filters", (50, 70), 3, num_classes=4, is_rgb=True)
    net_fixture(model, defns, "net_depth0_40x56.npz", "depth_ver_0", (40, 56), 2, num_classes=3)

    # full-size single image, eval only, weights = those of net_base_64x96 (same seed => same init draw?)
    # the init draw does not depend on the image size, but BN calibration does: store nothing but outputs
    torch.manual_seed(0)
    net = model.YOGO((772, 1032), 0.0425, 0.0555, 7)
    trained_like_bn(net)
    g = torch.Generator().manual_seed(10)
    x = torch.randint(0, 256, (1, 1, 772, 1032), dtype=torch.uint8, generator=g)
    calibrate_bn(net, x)
    net.eval()
    net.inference = True
    with torch.no_grad():
        out = net(x.clone())
    bn = {("sd/" + k): v for k, v in net.state_dict().items() if ".1." in k and v.ndim == 1 or "num_batches" in k}
    # conv weights equal net_base_64x96's (same manual_seed(0) draw order); only BN tensors differ
    npsave("net_base_full_eval.npz", out_inf=out, **bn,
           meta=np.array(json.dumps(dict(model="base_model", H=772, W=1032, B=1, num_classes=7, x_seed=10))))

    # ---- (v) loss from the reference's yogo_loss.py (on the torchvision restatement) -----------------
    def loss_case(name, B, C, Sy, Sx, seed, special):
        g = torch.Generator().manual_seed(seed)
        P = 5 + C
        pred = torch.zeros(B, P, Sy, Sx)
        cxs, cys = O.make_grids(Sx, Sy)
        raw = torch.randn(B, P, Sy, Sx, generator=g)
        pred = O.decode(raw, cxs, cys, 0.0425 * 4, 0.0555 * 4, inference=False)
        label = torch.zeros(B, 6, Sy, Sx)
        for b in range(B):
            K = 9
            c = torch.rand(K, 2, generator=g) * 0.9 + 0.05
            w = 0.17 * torch.exp(torch.randn(K, generator=g) * 0.2)
            h = 0.22 * torch.exp(torch.randn(K, generator=g) * 0.2)
            cls = torch.randint(0, C, (K,), generator=g).float()
            lab = torch.stack((cls, c[:, 0] - w / 2, c[:, 1] - h / 2, c[:, 0] + w / 2, c[:, 1] + h / 2), dim=1)
            label[b] = O.format_labels_tensor(lab, Sx, Sy)
        if special:
            # a zero-area prediction in a masked cell (dropped by the valid-box filter, yogo_loss.py:84-90)
            idx = torch.nonzero(label[0, 0])
            j, i = idx[0].tolist()
            pred[0, 2, j, i] = 0.0
            j, i = idx[1].tolist()
            pred[0, 3, j, i] = 0.0
            # a prediction reaching outside [0,1] (clamped, yogo_loss.py:96-100)
            j, i = idx[2].tolist()
            pred[0, 0, j, i] = 0.02
            pred[0, 2, j, i] = 0.3
            j, i = idx[3].tolist()
            pred[0, 1, j, i] = 0.99
            pred[0, 3, j, i] = 0.4
        pred = pred.clone().requires_grad_(True)
        L = loss_mod.YOGOLoss()
        lv, comps = L(pred, label)
        lv.backward()
        npsave(name, pred=pred, label=label, loss=lv.detach(), grad=pred.grad,
               comps=np.array([comps["iou_loss"], comps["objectness_loss"], comps["classification_loss"]], dtype=np.float64),
               weights=np.array([0.5, 5.0, 1.0, 0.01]))
        # non-default weights
        pred2 = pred.detach().clone().requires_grad_(True)
        L2 = loss_mod.YOGOLoss(no_obj_weight=0.25, iou_weight=2.0, classify_weight=3.0, label_smoothing=0.1)
        lv2, comps2 = L2(pred2, label)
        lv2.backward()
        npsave(name.replace(".npz", "_w2.npz"), loss=lv2.detach(), grad=pred2.grad,
               comps=np.array([comps2["iou_loss"], comps2["objectness_loss"], comps2["classification_loss"]], dtype=np.float64),
               weights=np.array([0.25, 2.0, 3.0, 0.1]))

    loss_case("loss_2x12x13x17.npz", 2, 7, 13, 17, 20, special=True)
    loss_case("loss_3x9x24x33.npz", 3, 4, 24, 33, 21, special=False)

    # ---- (vi) format_preds from the reference's prediction_formatting.py -------------------------------
    def fmt_case(name, pred, variants):
        payload = dict(pred=pred)
        for vi, kw in enumerate(variants):
            out = pf.format_preds(pred.clone(), **kw)
            payload[f"out{vi}"] = out
            payload[f"kw{vi}"] = np.array(json.dumps(kw))
        npsave(name, **payload)

    variants = [
        dict(), dict(box_format="xyxy"), dict(iou_thresh=0.0), dict(min_class_confidence_threshold=0.9),
        dict(obj_thresh=0.6, iou_thresh=0.3, box_format="xyxy", min_class_confidence_threshold=0.35),
    ]
    sparse = O.synthetic_predictions(1, 33, 24, num_classes=7, K=40, seed=30)[0]
    fmt_case("fmt_sparse_12x24x33.npz", sparse, variants)
    g = torch.Generator().manual_seed(31)
    dense = O.decode(torch.randn(1, 12, 24, 33, generator=g) * 1.5, *O.make_grids(33, 24), 0.17, 0.22, inference=True)[0]
    dense[4] = torch.rand(24, 33, generator=g) * 0.6 + 0.4
    fmt_case("fmt_dense_12x24x33.npz", dense, variants)
    # exact ties in score, identical boxes, zero-area boxes (0/0 -> NaN: not suppressed), raw-logit classes
    tie = sparse.clone()
    tie[:, 3, 4] = tie[:, 10, 20]
    tie[:, 5, 6] = tie[:, 10, 20]
    tie[4, 3, 4] = tie[4, 5, 6] = tie[4, 10, 20] = 0.9
    tie[2, 7, 7] = 0.0
    tie[3, 7, 7] = 0.0
    tie[4, 7, 7] = 0.95
    tie[2:4, 7, 8] = 0.0
    tie[0:2, 7, 8] = tie[0:2, 7, 7]
    tie[4, 7, 8] = 0.96
    fmt_case("fmt_ties_12x24x33.npz", tie, variants)
    logits = sparse.clone()
    logits[5:] = torch.randn(7, 24, 33, generator=g) * 3     # Metrics path feeds raw logits (inference=False)
    fmt_case("fmt_logits_12x24x33.npz", logits, variants)

    # ---- (viii) prediction <-> label matching from the reference's format_preds_and_labels_v2 (:254-330) ---------------
    lab = O.synthetic_labels(1, 33, 24, K=25, num_classes=7, seed=33)[0]
    for nm, pr in (("sparse", sparse), ("dense", dense)):
        for thr in (0.0, 0.9):
            m = pf.format_preds_and_labels_v2(pr.clone(), lab.clone(), objectness_thresh=0.5, min_class_confidence_threshold=thr)
            npsave(f"match_{nm}_{int(thr * 10)}.npz", pred=pr, label=lab, preds=m.preds, labels=m.labels,
                   missed=m.missed_labels, extra=m.extra_predictions)

    # ---- (vii) inference output arrays from the reference's format_to_numpy (prediction_formatting.py:96-156) --------
    npsave("fnp_12x24x33.npz", sparse=sparse, dense=dense,
           out_sparse=pf.format_to_numpy(3, sparse.numpy().copy(), 772, 1032),
           out_dense=pf.format_to_numpy(7, dense.numpy().copy(), 193, 1032))


if __name__ == "__main__":
    main()
