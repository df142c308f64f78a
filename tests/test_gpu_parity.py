"""Path-level parity on a real MI355X: the yogo_amd drop-in surface (YOGO / YOGOLoss / format_preds), which calls
the HIP kernels through the C ABI, against the golden vectors from the reference and against the CPU oracle."""
import json

import numpy as np
import pytest
import torch

import yogo_oracle as O
from _util import as_t, load_net_fixture, load_npz, rel_err

pytestmark = pytest.mark.gpu

NETS = ["net_base_64x96.npz", "net_quarter_rgb_50x70.npz", "net_depth0_40x56.npz", "net_silu_64x96.npz"]


def build_model(meta, sd, inference=False):
    from yogo_amd.model import YOGO
    from yogo_amd.model_defns import get_model_func

    m = YOGO((meta["H"], meta["W"]), meta["anchor_w"], meta["anchor_h"], meta["num_classes"], is_rgb=meta.get("is_rgb", False),
             inference=inference, model_func=get_model_func(meta["model"]))
    m.load_state_dict(sd)
    return m.to("cuda")


@pytest.mark.parametrize("fix", NETS)
def test_forward_eval_matches_reference(fix):
    meta, x, sd, grads, after, outs = load_net_fixture(fix)
    m = build_model(meta, sd)
    m.eval()
    with torch.no_grad():
        raw = m.model(x.cuda().float())
        out = m(x.cuda())
        m.inference = True
        out_inf = m(x.cuda())
    assert rel_err(raw.cpu(), outs["raw_eval"]) < 1e-4
    torch.testing.assert_close(out.cpu(), outs["out_eval"], rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(out_inf.cpu(), outs["out_inf"], rtol=1e-4, atol=1e-5)
    # the running stats must be untouched in eval
    for k, v in m.state_dict().items():
        if "running" in k:
            assert torch.equal(v.cpu(), sd[k])


@pytest.mark.parametrize("fix", NETS)
def test_train_forward_backward_matches_reference(fix):
    meta, x, sd, grads, after, outs = load_net_fixture(fix)
    m = build_model(meta, sd)
    m.train()
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout2d):
            mod.p = 0.0   # the fixture was generated with dropout disabled (RNG cannot be matched)
    out = m(x.cuda())
    torch.testing.assert_close(out.detach().cpu(), outs["out_train"], rtol=2e-4, atol=2e-4)
    out.backward(outs["upstream"].cuda())
    got = {k: p.grad.cpu() for k, p in m.named_parameters()}
    assert set(got) == set(grads)
    gmax = max(float(v.abs().max()) for v in grads.values())
    for k in grads:
        # a conv bias in front of BatchNorm has a mathematically zero gradient (rounding noise only): absolute floor
        err = float((got[k] - grads[k]).abs().max())
        assert err < 2e-3 * float(grads[k].abs().max()) + 1e-6 * gmax, (k, err)
    sd_now = m.state_dict()
    for k, v in after.items():
        torch.testing.assert_close(sd_now[k].cpu(), v, rtol=2e-4, atol=1e-4)


def test_full_size_eval_matches_reference():
    z = load_npz("net_base_full_eval.npz")
    meta = json.loads(str(z["meta"]))
    _, _, sd, *_ = load_net_fixture("net_base_64x96.npz")
    for k, v in z.items():
        if k.startswith("sd/"):
            sd[k[3:]] = as_t(v)
    cxs, cys = O.make_grids(129, 97)
    sd["_Cxs"], sd["_Cys"] = cxs, cys
    sd["img_size"] = torch.tensor([772, 1032])
    m = build_model(dict(model="base_model", H=772, W=1032, num_classes=7, anchor_w=0.0425, anchor_h=0.0555), sd, inference=True)
    m.eval()
    g = torch.Generator().manual_seed(meta["x_seed"])
    x = torch.randint(0, 256, (1, 1, 772, 1032), dtype=torch.uint8, generator=g)
    with torch.no_grad():
        out = m(x.cuda())
    torch.testing.assert_close(out.cpu(), as_t(z["out_inf"]), rtol=2e-4, atol=2e-5)


def test_dropout_mask_statistics_and_scaling():
    # Dropout2d zeroes whole channels with probability p and scales the rest by 1/(1-p)
    from yogo_amd.engine import get_engine

    meta, x, sd, *_ = load_net_fixture("net_base_64x96.npz")
    m = build_model(meta, sd)
    m.train()
    torch.manual_seed(0)
    xs = x.cuda().repeat(16, 1, 1, 1)
    raw, saved = get_engine(m.model).forward(xs, need_grad=True)
    for i, L in enumerate(get_engine(m.model).layers):
        if L.drop is not None:
            mask = saved[i].mask.cpu()
            p = L.drop.p
            for v in mask.unique().tolist():
                assert v == 0.0 or abs(v - 1 / (1 - p)) < 1e-6
            y = saved[i].y.cpu()
            dead = (mask == 0)
            assert torch.all(y[dead] == 0)
    assert torch.isfinite(raw).all()


@pytest.mark.parametrize("fix", ["loss_2x12x13x17", "loss_3x9x24x33"])
def test_loss_matches_reference(fix):
    from yogo_amd.yogo_loss import YOGOLoss

    z = load_npz(fix + ".npz")
    for suffix in ("", "_w2"):
        zz = load_npz(fix + suffix + ".npz")
        now, iw, cw, ls = (float(v) for v in zz["weights"])
        pred = as_t(z["pred"]).cuda().requires_grad_(True)
        L = YOGOLoss(now, iw, cw, ls).to("cuda")
        loss, comps = L(pred, as_t(z["label"]).cuda())
        loss.backward()
        assert abs(float(loss.detach()) - float(zz["loss"])) <= 2e-5 * abs(float(zz["loss"]))
        np.testing.assert_allclose([comps["iou_loss"], comps["objectness_loss"], comps["classification_loss"]], zz["comps"], rtol=2e-5)
        torch.testing.assert_close(pred.grad.cpu(), as_t(zz["grad"]), rtol=2e-4, atol=1e-6)


def test_loss_full_size_vs_oracle():
    from yogo_amd.yogo_loss import YOGOLoss

    B, C, Sy, Sx = 4, 7, 97, 129
    g = torch.Generator().manual_seed(40)
    raw = torch.randn(B, 5 + C, Sy, Sx, generator=g)
    pred = O.decode(raw, *O.make_grids(Sx, Sy), 0.0425, 0.0555).requires_grad_(True)
    label = O.synthetic_labels(B, Sx, Sy, K=64, num_classes=C, seed=41)
    ref, comps_ref = O.yogo_loss(pred, label)
    ref.backward()
    pd = pred.detach().cuda().requires_grad_(True)
    loss, comps = YOGOLoss().to("cuda")(pd, label.cuda())
    (loss * 2.0).backward()   # upstream gradient is honoured
    assert abs(float(loss.detach()) - float(ref.detach())) < 2e-5 * abs(float(ref.detach()))
    for k in comps_ref:
        assert abs(comps[k] - comps_ref[k]) <= 2e-5 * abs(comps_ref[k]) + 1e-7
    torch.testing.assert_close(pd.grad.cpu(), 2.0 * pred.grad, rtol=2e-4, atol=1e-6)


FMT = ["fmt_sparse_12x24x33", "fmt_dense_12x24x33", "fmt_ties_12x24x33", "fmt_logits_12x24x33"]


@pytest.mark.parametrize("fix", FMT)
def test_format_preds_bit_exact_vs_reference(fix):
    from yogo_amd.utils import format_preds

    z = load_npz(fix + ".npz")
    pred = as_t(z["pred"]).cuda()
    i = 0
    while f"out{i}" in z:
        kw = json.loads(str(z[f"kw{i}"]))
        out = format_preds(pred, **kw).cpu()
        want = as_t(z[f"out{i}"])
        assert out.shape == want.shape, (fix, kw, out.shape, want.shape)
        assert torch.equal(out, want), (fix, kw)
        i += 1
    assert i == 5


def test_format_preds_reference_known_answers():
    # tests/test_utils_tensor_formatting.py:8-68 of the reference, on the HIP path
    from yogo_amd.utils import format_preds

    none = torch.zeros(12, 4, 4)
    single = torch.zeros(12, 4, 4)
    single[4, 0, 0] = 1.0
    single[5] = 1.0
    box = torch.zeros(12, 4, 4)
    box[5] = 1.0
    box[4, 1, 1] = 1.0
    box[0, 1, 1] = 0.5
    box[1, 1, 1] = 0.5
    box[2, 1, 1] = 0.1
    box[3, 1, 1] = 0.1
    torch.testing.assert_close(format_preds(none.cuda()).cpu(), torch.empty(0, 12))
    torch.testing.assert_close(format_preds(single.cuda()).cpu(), single[:, 0, 0].unsqueeze(0))
    torch.testing.assert_close(format_preds(box.cuda()).cpu(), box[:, 1, 1].unsqueeze(0))
    actual = box[:, 1, 1].unsqueeze(0).clone()
    actual[:, 0] = actual[:, 0] - actual[:, 2] / 2
    actual[:, 1] = actual[:, 1] - actual[:, 3] / 2
    actual[:, 2] = actual[:, 0] + actual[:, 2]
    actual[:, 3] = actual[:, 1] + actual[:, 3]
    torch.testing.assert_close(format_preds(box.cuda(), box_format="xyxy").cpu(), actual)
    with pytest.raises(ValueError):
        format_preds(torch.zeros(1, 12, 4, 4).cuda())
    with pytest.raises(ValueError):
        format_preds(none.cuda(), box_format="xywh")


@pytest.mark.parametrize("kind", ["realistic", "dense"])
def test_format_preds_batched_full_size_bit_exact_vs_oracle(kind):
    from yogo_amd.utils import format_preds_batched

    Sx, Sy, B = 129, 97, 6
    if kind == "realistic":
        pred = O.synthetic_predictions(B, Sx, Sy, num_classes=7, K=100, seed=50)
    else:
        g = torch.Generator().manual_seed(51)
        pred = O.decode(torch.randn(B, 12, Sy, Sx, generator=g) * 1.5, *O.make_grids(Sx, Sy), 0.0425, 0.0555, inference=True)
        pred[:, 4] = torch.rand(B, Sy, Sx, generator=g) * 0.55 + 0.45   # ~90% of the cells pass 0.5
    for kw in (dict(), dict(box_format="xyxy", min_class_confidence_threshold=0.3), dict(iou_thresh=0.0)):
        rows, cells, counts = format_preds_batched(pred.cuda(), **kw)
        rows, cells, counts = rows.cpu(), cells.cpu(), counts.cpu()
        for b in range(B):
            want, wcells = O.format_preds(pred[b], return_cells=True, **kw)
            n = int(counts[b])
            assert n == want.shape[0], (kind, kw, b, n, want.shape[0])
            assert torch.equal(cells[b, :n], wcells), (kind, kw, b)
            assert torch.equal(rows[b, :n], want), (kind, kw, b)


def test_class_counts_match_oracle():
    from yogo_amd.utils import get_prediction_class_counts

    pred = O.synthetic_predictions(3, 129, 97, num_classes=7, K=80, seed=52)
    got = get_prediction_class_counts(pred.cuda(), min_class_confidence_threshold=0.25)
    want = O.get_prediction_class_counts(pred, min_class_confidence_threshold=0.25)
    assert torch.equal(got, want)


def test_inference_writers_match_oracle(tmp_path):
    """save_predictions / format_to_numpy (yogo/infer.py:39-57, prediction_formatting.py:96-156) over the batched kernel"""
    from yogo_amd.utils import format_to_numpy, format_to_numpy_batched, save_predictions

    preds = O.synthetic_predictions(3, 33, 24, num_classes=7, K=30, seed=77)
    names = [str(tmp_path / f"img{i}.txt") for i in range(3)]
    save_predictions(names, preds.cuda(), obj_thresh=0.5, iou_thresh=0.5)
    for b, n in enumerate(names):
        assert open(n).read() == O.save_predictions_text(O.format_preds(preds[b]))
    arrs = format_to_numpy_batched([10, 11, 12], preds.cuda(), 772, 1032)
    for b in range(3):
        want = O.format_to_numpy(10 + b, preds[b].numpy(), 772, 1032)
        assert arrs[b].dtype == want.dtype and np.array_equal(arrs[b], want)
    assert np.array_equal(format_to_numpy(5, preds[1].numpy(), 193, 1032), O.format_to_numpy(5, preds[1].numpy(), 193, 1032))


def test_format_preds_and_labels_v2_matches_oracle():
    """metrics-side matching over the batched kernel (prediction_formatting.py:254-330)"""
    from yogo_amd.utils import PredictionLabelMatch, format_preds_and_labels_v2, format_preds_and_labels_v2_batched

    preds = O.synthetic_predictions(3, 33, 24, num_classes=7, K=30, seed=78)
    labels = O.synthetic_labels(3, 33, 24, K=25, num_classes=7, seed=79)
    ms = format_preds_and_labels_v2_batched(preds.cuda(), labels, 0.5, 0.0)
    for b, m in enumerate(ms):
        want = O.format_preds_and_labels_v2(preds[b], labels[b], 0.5, 0.0)
        for g, w in zip((m.preds, m.labels, m.missed_labels, m.extra_predictions), want):
            assert torch.equal(g, w)
    one = format_preds_and_labels_v2(preds[1].cuda(), labels[1], 0.5, 0.9)
    want = O.format_preds_and_labels_v2(preds[1], labels[1], 0.5, 0.9)
    assert torch.equal(one.preds, want[0]) and torch.equal(one.extra_predictions, want[3])
    cat = PredictionLabelMatch.concat(ms)
    assert cat.preds.shape[0] == sum(m.preds.shape[0] for m in ms)
    conv = cat.convert_background_errors(8)
    assert conv.preds.shape[1] == 5 + 8 and conv.labels.shape[1] == 6 and conv.missed_labels is None


def test_trainer_steps_match_oracle():
    """two full optimisation steps (fwd, loss, bwd, clamp, AdamW with cosine LR) against the CPU oracle"""
    from yogo_amd.model import YOGO
    from yogo_amd.train import HipTrainer
    from yogo_amd.yogo_loss import YOGOLoss

    torch.manual_seed(1)
    Himg, Wimg, C, B = 96, 128, 7, 3
    model = YOGO((Himg, Wimg), 0.0425, 0.0555, C).to("cuda")
    model.train()
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout2d):
            m.p = 0.0
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    spec = O.arch("base_model", C)
    names = [k for k, v in sd.items() if k.startswith("model.") and v.is_floating_point() and "running" not in k]
    mom = {k: (torch.zeros_like(sd[k]), torch.zeros_like(sd[k])) for k in names}
    tr = HipTrainer(model, YOGOLoss().to("cuda"), total_steps=7)
    for step in (1, 2):
        x = O.synthetic_images(B, Himg, Wimg, seed=10 + step)
        lab = O.synthetic_labels(B, model.Sx, model.Sy, K=5, num_classes=C, seed=20 + step)
        tr.step(x.cuda(), lab.cuda())
        leaf = {k: sd[k].clone().requires_grad_(True) for k in names}
        sdl = dict(sd)
        sdl.update(leaf)
        ns = {}
        pred = O.yogo_forward(x, sdl, spec, 0.0425, 0.0555, train=True, new_stats=ns)
        loss, comps = O.yogo_loss(pred, lab)
        loss.backward()
        g = O.clamp_grads({k: v.grad for k, v in leaf.items()})
        lr = O.cosine_lr(step - 1, 3e-4, 7, 3e-5)
        for k in names:
            p, m1, v1 = O.adamw_step(sd[k], g[k], mom[k][0], mom[k][1], step, lr)
            sd[k], mom[k] = p.detach(), (m1, v1)
        sd.update(ns)
        got = tr.loss_components()
        assert abs(got["loss"] - float(loss.detach())) < 2e-4 * abs(float(loss.detach()))
        for k2 in ("iou_loss", "objectness_loss", "classification_loss"):
            assert abs(got[k2] - comps[k2]) < 2e-4 * abs(comps[k2]) + 1e-6
    now = model.state_dict()
    for k in names:
        # Adam normalises each element's step to ~lr whatever the gradient magnitude, so an element whose true gradient is
        # ~0 (e.g. model.5.0.bias in front of BatchNorm, or weights fed by dead channels) takes a +-lr step whose SIGN is
        # rounding noise -- in the reference too.  Hence: (almost) all elements agree tightly, none drifts further than
        # the two steps could possibly move it.
        d = (now[k].cpu() - sd[k]).abs()
        assert float(d.max()) < 2.5 * 2 * 3e-4, k
        if k != "model.5.0.bias":
            assert float((d > 3e-5).float().mean()) < 2e-3, (k, float((d > 3e-5).float().mean()))
    for k, v in sd.items():
        if "running" in k:
            torch.testing.assert_close(now[k].cpu(), v, rtol=1e-4, atol=1e-4)
        if "num_batches" in k:
            assert int(now[k]) == 2
    # the flat buffer really is the module's storage
    assert all(p.data_ptr() >= tr.flat.flat.data_ptr() for p in model.parameters())


@pytest.mark.parametrize("fix", ["net_base_64x96.npz", "net_quarter_rgb_50x70.npz", "net_depth0_40x56.npz", "net_silu_64x96.npz"])
def test_bf16_inference_forward(fix):
    """`yogo infer` runs the model under bf16 autocast (yogo/infer.py:313-317).  bf16 storage + bf16 MFMA with fp32
    accumulation against the fp32 reference output: tolerance = bf16 rounding through 8 layers (stated: 3e-2 of the range)."""
    meta, x, sd, grads, after, outs = load_net_fixture(fix)
    m = build_model(meta, sd, inference=True)
    m.eval()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        raw = m.model(x.cuda())
        out = m(x.cuda())
    assert raw.dtype == torch.float32 and out.dtype == torch.float32
    ref = outs["raw_eval"]
    err = float((raw.cpu() - ref).abs().max() / ref.abs().max())
    assert err < 3e-2, err
    # decoded: boxes / objectness / class probabilities are all O(1) quantities
    d = (out.cpu() - outs["out_inf"]).abs()
    assert float(d[:, [0, 1, 4]].max()) < 2e-2 and float(d[:, 5:].max()) < 6e-2
    rel_wh = d[:, 2:4] / outs["out_inf"][:, 2:4].abs().clamp_min(1e-6)
    assert float(rel_wh.max()) < 0.15
    # and the fp32 path is untouched outside autocast
    with torch.no_grad():
        raw32 = m.model(x.cuda().float())
    assert rel_err(raw32.cpu(), ref) < 1e-4


def test_bf16_inference_full_size():
    z = load_npz("net_base_full_eval.npz")
    meta = json.loads(str(z["meta"]))
    _, _, sd, *_ = load_net_fixture("net_base_64x96.npz")
    for k, v in z.items():
        if k.startswith("sd/"):
            sd[k[3:]] = as_t(v)
    sd["_Cxs"], sd["_Cys"] = O.make_grids(129, 97)
    sd["img_size"] = torch.tensor([772, 1032])
    m = build_model(dict(model="base_model", H=772, W=1032, num_classes=7, anchor_w=0.0425, anchor_h=0.0555), sd, inference=True)
    m.eval()
    g = torch.Generator().manual_seed(meta["x_seed"])
    x = torch.randint(0, 256, (1, 1, 772, 1032), dtype=torch.uint8, generator=g)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        out = m(x.cuda()).cpu()
    want = as_t(z["out_inf"])
    d = (out - want).abs()
    # bf16 keeps 8 significand bits: logits of magnitude ~10 carry ~4e-2 absolute error after 8 layers, which the sigmoid /
    # softmax map to at most a few 1e-2 of probability
    assert float(d[:, [0, 1, 4]].max()) < 4e-2 and float(d[:, 5:].max()) < 8e-2
    assert float(d[:, [0, 1, 4]].mean()) < 2e-3
