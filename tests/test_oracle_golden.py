"""The CPU oracle (oracle/yogo_oracle.py) against the golden vectors generated from the real reference
(tests/golden/make_golden.py) and against the reference's own known-answer tests.  CPU only."""
import json

import numpy as np
import pytest
import torch

import yogo_oracle as O
from _util import as_t, load_net_fixture, load_npz, rel_err

NETS = ["net_base_64x96.npz", "net_silu_64x96.npz", "net_quarter_rgb_50x70.npz", "net_depth0_40x56.npz"]


def test_grid_sizes_and_linspace_grids():
    z = load_npz("grid.npz")
    sizes = json.loads(str(z["sizes"]))
    spec = O.arch("base_model", 7)
    for k, (sx, sy) in sizes.items():
        h, w = (int(v) for v in k.split("x"))
        assert O.grid_size(spec, h, w) == (sx, sy)
    cxs, cys = O.make_grids(129, 97)
    assert torch.equal(cxs, as_t(z["Cxs"])) and torch.equal(cys, as_t(z["Cys"]))


def test_checkpoint_key_set_matches_reference():
    keys = json.load(open(__import__("os").path.join(__import__("_util").GOLDEN, "ckpt_keys.json")))
    for name, info in keys.items():
        spec = O.arch(name, 7)
        sd = O.init_state(spec)
        want = [(k, tuple(s)) for k, s, _ in info["keys"] if k.startswith("model.")]
        got = [(k, tuple(v.shape)) for k, v in sd.items()]
        assert got == want, name
        nparams = sum(v.numel() for k, v in sd.items() if "running" not in k and "num_batches" not in k)
        assert nparams == info["num_params"]
        assert O.grid_size(spec, 772, 1032) == (info["Sx"], info["Sy"])


def test_kaiming_init_statistics():
    stats = json.load(open(__import__("os").path.join(__import__("_util").GOLDEN, "init_stats.json")))
    sd = O.init_state(O.arch("base_model", 7), seed=5)
    for k, (mean, std, shape) in stats.items():
        w = sd[k]
        assert list(w.shape) == shape
        # same distribution (different RNG draw): std within 3 standard errors of the analytic value
        n = w.numel()
        assert abs(float(w.std()) - std) < 6 * std / np.sqrt(2 * n) + 1e-3 * std


@pytest.mark.parametrize("fix", NETS)
def test_backbone_decode_eval(fix):
    meta, x, sd, grads, after, outs = load_net_fixture(fix)
    spec = O.arch(meta["model"], meta["num_classes"])
    raw = O.backbone_forward(x.float(), sd, spec, train=False)
    assert rel_err(raw, outs["raw_eval"]) < 1e-5
    for inference, key in ((False, "out_eval"), (True, "out_inf")):
        out = O.yogo_forward(x, sd, spec, meta["anchor_w"], meta["anchor_h"], inference=inference)
        torch.testing.assert_close(out, outs[key], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("fix", NETS)
def test_backbone_train_forward_backward(fix):
    meta, x, sd, grads, after, outs = load_net_fixture(fix)
    spec = O.arch(meta["model"], meta["num_classes"])
    leaf = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and (k.endswith("weight") or k.endswith("bias")) and k.startswith("model.")}
    sdl = dict(sd)
    sdl.update(leaf)
    new_stats = {}
    out = O.yogo_forward(x, sdl, spec, meta["anchor_w"], meta["anchor_h"], inference=False, train=True, new_stats=new_stats)
    torch.testing.assert_close(out, outs["out_train"], rtol=1e-4, atol=1e-5)
    out.backward(outs["upstream"])
    g = O.clamp_grads({k: v.grad for k, v in leaf.items()}, 1.0)
    assert set(g) == set(grads)
    for k in grads:
        assert rel_err(g[k], grads[k]) < 2e-4, k
    for k, v in after.items():
        torch.testing.assert_close(new_stats[k].to(v.dtype), v, rtol=1e-4, atol=1e-5)


def test_full_size_eval():
    z = load_npz("net_base_full_eval.npz")
    meta = json.loads(str(z["meta"]))
    _, _, sd, *_ = load_net_fixture("net_base_64x96.npz")
    for k, v in z.items():
        if k.startswith("sd/"):
            sd[k[3:]] = as_t(v)
    sd.pop("_Cxs"), sd.pop("_Cys")
    g = torch.Generator().manual_seed(meta["x_seed"])
    x = torch.randint(0, 256, (1, 1, 772, 1032), dtype=torch.uint8, generator=g)
    out = O.yogo_forward(x, sd, O.arch("base_model", 7), 0.0425, 0.0555, inference=True)
    torch.testing.assert_close(out, as_t(z["out_inf"]), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("fix", ["loss_2x12x13x17", "loss_3x9x24x33"])
def test_loss_against_reference_control_flow(fix):
    z = load_npz(fix + ".npz")
    for suffix in ("", "_w2"):
        zz = load_npz(fix + suffix + ".npz")
        now, iw, cw, ls = (float(v) for v in zz["weights"])
        pred = as_t(z["pred"]).requires_grad_(True)
        loss, comps = O.yogo_loss(pred, as_t(z["label"]), now, iw, cw, ls)
        loss.backward()
        assert abs(float(loss.detach()) - float(zz["loss"])) <= 1e-6 * abs(float(zz["loss"]))
        np.testing.assert_allclose([comps["iou_loss"], comps["objectness_loss"], comps["classification_loss"]], zz["comps"], rtol=1e-6)
        torch.testing.assert_close(pred.grad, as_t(zz["grad"]), rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("fix", ["fmt_sparse_12x24x33", "fmt_dense_12x24x33", "fmt_ties_12x24x33", "fmt_logits_12x24x33"])
def test_format_preds_against_reference_control_flow(fix):
    z = load_npz(fix + ".npz")
    i = 0
    while f"out{i}" in z:
        kw = json.loads(str(z[f"kw{i}"]))
        out = O.format_preds(as_t(z["pred"]), **kw)
        assert torch.equal(out, as_t(z[f"out{i}"])), (fix, kw)
        i += 1
    assert i == 5


@pytest.mark.parametrize("fix", ["match_sparse_0", "match_sparse_9", "match_dense_0", "match_dense_9"])
def test_format_preds_and_labels_v2_against_reference(fix):
    """prediction <-> label matching (prediction_formatting.py:254-330); fixtures written by the real function"""
    z = load_npz(fix + ".npz")
    thr = 0.9 if fix.endswith("9") else 0.0
    got = O.format_preds_and_labels_v2(as_t(z["pred"]), as_t(z["label"]), 0.5, thr)
    for g, name in zip(got, ("preds", "labels", "missed", "extra")):
        assert torch.equal(g, as_t(z[name]).reshape(g.shape)), (fix, name)
    # the product's host-side matcher on the oracle's kept rows gives the same partition
    from yogo_amd.utils.prediction_formatting import _match_rows_to_labels

    rows = O.format_preds(as_t(z["pred"]), 0.5, 0.5, "xyxy", thr)
    m = _match_rows_to_labels(rows, as_t(z["label"]))
    for g, name in zip((m.preds, m.labels, m.missed_labels, m.extra_predictions), ("preds", "labels", "missed", "extra")):
        assert torch.equal(g, as_t(z[name]).reshape(g.shape)), (fix, name)


def test_format_to_numpy_against_reference():
    """inference output array of yogo/utils/prediction_formatting.py:96-156 (fixture written by the real function)"""
    z = load_npz("fnp_12x24x33.npz")
    for name, img_id, h, w in (("sparse", 3, 772, 1032), ("dense", 7, 193, 1032)):
        got = O.format_to_numpy(img_id, np.array(z[name]), h, w)
        want = z["out_" + name]
        assert got.dtype == want.dtype and got.shape == want.shape and want.shape[0] == 8 + 7
        assert np.array_equal(got, want), name


def test_save_predictions_text_known_answer():
    """row format of yogo/infer.py:52-55: first-argmax class, then xc yc w h printed as Python floats of the float32 values"""
    rows = torch.tensor([[0.5, 0.25, 0.125, 0.0625, 0.9, 0.1, 0.7, 0.7, 0.0],
                         [0.1, 0.2, 0.3, 0.4, 0.8, 0.6, 0.1, 0.1, 0.2]])
    txt = O.save_predictions_text(rows)
    assert txt.split("\n")[0] == "1 0.5 0.25 0.125 0.0625"            # tie between classes 1 and 2 -> the first
    assert txt.split("\n")[1] == f"0 {float(np.float32(0.1))} {float(np.float32(0.2))} {float(np.float32(0.3))} {float(np.float32(0.4))}"
    assert O.save_predictions_text(torch.zeros(0, 9)) == ""


def test_host_side_writers_match_oracle():
    """the product's host-side formatting (no GPU involved) against the oracle's restatement"""
    from yogo_amd.utils.prediction_formatting import _rows_xyxy_to_numpy, prediction_rows_to_text

    z = load_npz("fnp_12x24x33.npz")
    pred = as_t(z["sparse"])
    rows = O.format_preds(pred)
    assert prediction_rows_to_text(rows) == O.save_predictions_text(rows)
    rows_xyxy = O.format_preds(pred, box_format="xyxy")
    assert np.array_equal(_rows_xyxy_to_numpy(3, rows_xyxy, 772, 1032, np.float32), z["out_sparse"])


# ---- the reference's own known-answer tests (tests/test_utils_tensor_formatting.py:8-68) ------------------
def _kat_tensors():
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
    return none, single, box


def test_reference_kat_format_preds():
    none, single, box = _kat_tensors()
    torch.testing.assert_close(O.format_preds(none), torch.empty(0, 12))
    torch.testing.assert_close(O.format_preds(single), single[:, 0, 0].unsqueeze(0))
    torch.testing.assert_close(O.format_preds(box), box[:, 1, 1].unsqueeze(0))
    actual = box[:, 1, 1].unsqueeze(0).clone()
    actual[:, 0] = actual[:, 0] - actual[:, 2] / 2
    actual[:, 1] = actual[:, 1] - actual[:, 3] / 2
    actual[:, 2] = actual[:, 0] + actual[:, 2]
    actual[:, 3] = actual[:, 1] + actual[:, 3]
    torch.testing.assert_close(O.format_preds(box, box_format="xyxy"), actual)
    with pytest.raises(ValueError):
        O.format_preds(torch.zeros(1, 12, 4, 4))
    with pytest.raises(ValueError):
        O.format_preds(none, box_format="xywh")


# ---- tests/test_count_predictions.py:7-42 ---------------------------------------------------------------
def test_reference_kat_count_cells():
    inp = torch.zeros(3, 5)
    inp[:, 0] = 1
    torch.testing.assert_close(O.count_cells_for_formatted_preds(inp), torch.tensor([3, 0, 0, 0, 0]))
    row = torch.tensor([0.1, 0.2, 0.3, 0.4])
    torch.testing.assert_close(O.count_cells_for_formatted_preds(torch.stack([row] * 3)), torch.tensor([0, 0, 0, 3]))
    inp = torch.tensor([[0.2, 0.4, 0.2, 0.2]] * 3)
    torch.testing.assert_close(O.count_cells_for_formatted_preds(inp, 0.6), torch.tensor([0, 0, 0, 0]))
    inp = torch.tensor([[0.2, 0.7, 0.2, 0.2], [0.2, 0.4, 0.2, 0.2], [0.2, 0.4, 0.9, 0.2]])
    torch.testing.assert_close(O.count_cells_for_formatted_preds(inp, 0.6), torch.tensor([0, 1, 1, 0]))


def test_nms_suppression_order_and_nan():
    # hand-computable: box1 overlaps box0 with IoU 0.6 > 0.5 (suppressed), box2 disjoint, box3 zero-area twin of box0
    boxes = np.array([[0, 0, 1, 1], [0, 0.25, 1, 1.0], [2, 2, 3, 3], [0.5, 0.5, 0.5, 0.5], [0.5, 0.5, 0.5, 0.5]], dtype=np.float32)
    scores = np.array([0.9, 0.8, 0.95, 0.7, 0.7], dtype=np.float32)
    keep = O.nms_numpy(boxes, scores, 0.5)
    assert keep.tolist() == [2, 0, 3, 4]   # 3 and 4: 0/0 = NaN vs each other, never suppressed; tie keeps index order
    assert O.nms_numpy(boxes, scores, 0.8).tolist() == [2, 0, 1, 3, 4]


def test_adamw_and_cosine_match_torch():
    torch.manual_seed(0)
    p = torch.nn.Parameter(torch.randn(50))
    opt = torch.optim.AdamW([p], lr=3e-4, weight_decay=5e-2)
    sch = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=20, eta_min=3e-5)
    q, m, v = p.detach().clone(), torch.zeros(50), torch.zeros(50)
    for step in range(1, 8):
        g = torch.randn(50)
        p.grad = g.clone()
        lr = O.cosine_lr(step - 1, 3e-4, 20, 3e-5)
        assert abs(lr - opt.param_groups[0]["lr"]) < 1e-12
        opt.step()
        sch.step()
        q, m, v = O.adamw_step(q, g, m, v, step, lr)
        torch.testing.assert_close(q, p.detach(), rtol=1e-6, atol=1e-8)
