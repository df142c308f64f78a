"""The inference driver's fused decode + threshold + NMS (`yogo_decode_format_preds_batched`, SURVEY.md 8(b) `decode_nms_batched`)
against the two separate calls it replaces (yogo/model.py:277-313 -> yogo/utils/prediction_formatting.py:23-93, call sites
yogo/infer.py:45,73).  The bar is bit-identity of rows, cells and counts with `format_preds_batched(decode(raw))`; that two-pass
form is itself held bit-exact to the oracle's format_preds on the same decoded tensor in test_gpu_parity.py, which pins the fused
form by transitivity (the decode's own transcendental rounding is the one thing the CPU oracle cannot reproduce bit for bit)."""
import pytest
import torch

import yogo_oracle as O

pytestmark = pytest.mark.gpu


def _raw_cases():
    g = torch.Generator().manual_seed(90)
    Sx, Sy, B = 129, 97, 5
    dense = torch.randn(B, 12, Sy, Sx, generator=g) * 1.5
    dense[:, 4] += 2.0                                   # ~90 % of the cells pass 0.5: the NMS worst case
    sparse = torch.randn(B, 12, Sy, Sx, generator=g)
    sparse[:, 4] = torch.randn(B, Sy, Sx, generator=g) - 2.5
    # objects seen by 2-4 neighbouring cells with nearly the same box: suppression has work to do
    for b in range(B):
        ys = torch.randint(1, Sy - 2, (120,), generator=g)
        xs = torch.randint(1, Sx - 2, (120,), generator=g)
        for y, x in zip(ys.tolist(), xs.tolist()):
            sparse[b, 4, y, x] = 2.0 + torch.rand(1, generator=g).item()
            for dy, dx in ((0, 1), (1, 0), (1, 1))[: int(torch.randint(1, 4, (1,), generator=g))]:
                sparse[b, :, y + dy, x + dx] = sparse[b, :, y, x] + 0.05 * torch.randn(12, generator=g)
    odd = torch.randn(3, 9, 7, 11, generator=g)          # 4 classes, a grid that is no multiple of anything
    odd[0, 2, 1, 1] = 95.0                               # width clamp at exp(80) -> inf
    odd[0, 4, 1, 1] = 4.0
    odd[1, 5:, 2, 3] = float("nan")                      # NaN class logits
    odd[1, 4, 2, 3] = 5.0
    odd[1, 4, 0, 0] = float("nan")                       # NaN objectness: not > thresh
    odd[2, 4] = -20.0                                    # nothing passes
    return {"dense": dense, "sparse": sparse, "odd": odd}


def _two_pass(raw, grids, scal, inference, **kw):
    from yogo_amd.utils.prediction_formatting import RawPredictions, format_preds_batched

    rp = RawPredictions(raw, grids[0], grids[1], *scal, inference)
    return format_preds_batched(rp.decoded(), **kw), format_preds_batched(rp, **kw)


@pytest.mark.parametrize("kind", ["dense", "sparse", "odd"])
def test_fused_decode_format_preds_is_bit_identical_to_the_two_passes(kind):
    raw = _raw_cases()[kind].cuda()
    B, P, Sy, Sx = raw.shape
    cxs, cys = (t.cuda() for t in O.make_grids(Sx, Sy))
    total = 0
    for inference in (True, False):
        for scal in ((0.0425, 0.0555, 1.0, 1.0), (0.05, 0.07, 1.0, 772 / 193)):
            for kw in (dict(), dict(box_format="xyxy", min_class_confidence_threshold=0.3), dict(iou_thresh=0.0),
                       dict(obj_thresh=0.8, iou_thresh=0.2)):
                (r0, c0, n0), (r1, c1, n1) = _two_pass(raw, (cxs, cys), scal, inference, **kw)
                assert torch.equal(n0, n1), (kind, inference, kw)
                for b, n in enumerate(n0.cpu().tolist()):
                    assert torch.equal(c0[b, :n], c1[b, :n]), (kind, inference, kw, b)
                    # bitwise (NaN rows included): compare the float32 bit patterns
                    assert torch.equal(r0[b, :n].view(torch.int32), r1[b, :n].view(torch.int32)), (kind, inference, kw, b)
                    total += n
    assert total > 0
    if kind == "odd":
        (_, c, n), _ = _two_pass(raw, (cxs, cys), (0.0425, 0.0555, 1.0, 1.0), True)
        assert int(n[2]) == 0 and int(n[0]) > 0 and 1 * 11 + 1 in c[0, : int(n[0])].tolist()


def test_forward_raw_feeds_the_inference_outputs_without_a_decode_pass(tmp_path):
    """YOGO.forward_raw -> save_predictions / class counts / numpy rows: same files and arrays as from model(x), with ONE
    nms_batched_kernel<true> launch per output and no decode_fwd_kernel launch (launch log)"""
    import numpy as np

    from yogo_amd import _hip
    from yogo_amd.model import YOGO
    from yogo_amd.utils import format_to_numpy_batched, get_prediction_class_counts, save_predictions

    torch.manual_seed(5)
    m = YOGO((193, 258), 0.0425, 0.0555, 7, inference=True).cuda().eval()
    x = O.synthetic_images(4, 193, 258, seed=91).cuda()
    for half in (False, True):
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=half):
            full = m(x)
            _hip.launch_log(True)
            raw = m.forward_raw(x)
            names = [str(tmp_path / f"a{int(half)}_{i}.txt") for i in range(4)]
            save_predictions(names, raw, obj_thresh=0.4, iou_thresh=0.5)
            counts = get_prediction_class_counts(raw, obj_thresh=0.4)
            arrs = format_to_numpy_batched([0, 1, 2, 3], raw, 193, 258)
            log = _hip.read_launch_log()
            _hip.launch_log(False)
        assert sum("nms_batched_kernel<true>" in ln for ln in log) == 3, log
        assert not any("decode_fwd_kernel" in ln or "nms_batched_kernel<false>" in ln for ln in log), log
        assert torch.equal(raw.decoded(), full)
        ref = [str(tmp_path / f"b{int(half)}_{i}.txt") for i in range(4)]
        save_predictions(ref, full, obj_thresh=0.4, iou_thresh=0.5)
        assert [open(n).read() for n in names] == [open(n).read() for n in ref]
        assert any(open(n).read() for n in names)
        assert torch.equal(counts, get_prediction_class_counts(full, obj_thresh=0.4))
        for a, b in zip(arrs, format_to_numpy_batched([0, 1, 2, 3], full, 193, 258)):
            assert np.array_equal(a, b, equal_nan=True)


def test_nms_degenerate_boxes_bit_exact_vs_oracle():
    """zero-area, negative-extent, infinite, NaN, identical and exactly-touching boxes: the kept rows must be the CPU algorithm's
    (torchvision nms semantics: 0/0 = NaN does not suppress, no eps), index for index"""
    from yogo_amd.utils import format_preds_batched

    g = torch.Generator().manual_seed(93)
    B, Sy, Sx = 6, 16, 16
    pred = O.synthetic_predictions(B, Sx, Sy, num_classes=7, K=40, seed=94)
    pred[:, 4] = torch.rand(B, Sy, Sx, generator=g) * 0.6 + 0.4
    flat = pred.view(B, 12, Sy * Sx)
    n = Sy * Sx
    idx = torch.randperm(n, generator=g)
    flat[:, 2, idx[:30]] = 0.0                                   # zero width
    flat[:, 3, idx[20:50]] = 0.0                                 # zero height (some both)
    flat[:, 2, idx[50:70]] = -0.05                               # negative extent
    flat[:, 2, idx[70:75]] = float("inf")
    flat[:, 0, idx[75:80]] = float("nan")
    flat[:, :4, idx[80:110]] = flat[:, :4, idx[110:140]]         # identical boxes (IoU exactly 1)
    # exactly touching neighbours: same size, centres one width apart in binary-exact numbers
    flat[:, 0, idx[140:150]] = 0.25
    flat[:, 0, idx[150:160]] = 0.375
    flat[:, 1, idx[140:160]] = 0.5
    flat[:, 2, idx[140:160]] = 0.125
    flat[:, 3, idx[140:160]] = 0.125
    for kw in (dict(), dict(iou_thresh=0.01), dict(iou_thresh=0.999, box_format="xyxy")):
        rows, cells, counts = format_preds_batched(pred.cuda(), **kw)
        rows, cells, counts = rows.cpu(), cells.cpu(), counts.cpu()
        for b in range(B):
            want, wcells = O.format_preds(pred[b], return_cells=True, **kw)
            k = int(counts[b])
            assert k == want.shape[0], (kw, b, k, want.shape[0])
            assert torch.equal(cells[b, :k], wcells), (kw, b)
            assert torch.equal(rows[b, :k].view(torch.int32), want.view(torch.int32)), (kw, b)


def test_configs4_batch_256_inference_plan_and_results():
    """BASELINE configs[4] at its real size: `yogo infer`'s path at batch 256 (bf16 eval forward at 772x1032 -> YOGO.forward_raw -> the
    fused decode + threshold + NMS launch, yogo/infer.py:300-386) on 128 copies of two images.  (i) the launch log holds every
    convolution / post-process kernel of the committed rocprofv3 summary of this path (profiles/r*_infer_kernel_stats.txt): the
    batch-256 plan the bench measures is the plan tested here; (ii) the copies agree with each other bit for bit; (iii) the rows
    and cells of the two images are the CPU algorithm's (oracle format_preds) on the decoded tensors of the same run -- index
    for index, bit for bit; (iv) the decoded tensor is within the bf16 inference bound of the fp32 CPU oracle's forward."""
    import glob
    import os
    import re

    from yogo_amd import _hip
    from yogo_amd.model import YOGO
    from yogo_amd.utils import format_preds_batched

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    torch.manual_seed(7)
    m = YOGO((772, 1032), 0.0425, 0.0555, 7, inference=True).cuda().eval()
    for k, v in m.state_dict().items():   # trained-like running statistics keep the eval-mode activations finite (as bench.py's CPU leg)
        if k.endswith("running_var"):
            v.fill_(5000.0)
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    x2 = O.synthetic_images(2, 772, 1032, seed=95)
    x = x2.cuda().repeat(128, 1, 1, 1)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        _hip.launch_log(True)
        raw = m.forward_raw(x)
        rows, cells, counts = format_preds_batched(raw, obj_thresh=0.5, iou_thresh=0.5)
        torch.cuda.synchronize()
        log = _hip.read_launch_log()
        _hip.launch_log(False)
        dec = raw.decoded()
    # (i) the kernel set of the committed profile of this path
    launched = {re.sub(r"\s+", "", ln.split("|")[0]) for ln in log}
    profs = sorted(glob.glob(os.path.join(root, "profiles", "r*_infer_kernel_stats.txt")))
    assert profs
    want = set()
    for ln in open(profs[-1]):
        if ln.startswith("# rocprofv3") and "post" in ln:
            break   # (the second table is the post-process alone on synthetic predictions)
        mm = re.search(r"((?:conv_bf16_kernel|conv_bf16_ws\d?_kernel|nms_batched_kernel)<[^>]*>)", ln)
        if mm and "nms_batched_kernel<false>" not in mm.group(1):   # (the profile also times the two-pass form)
            want.add(re.sub(r"\s+", "", mm.group(1)))
    assert want and not (want - launched), (sorted(want - launched), sorted(launched))
    # (ii) replication
    n = counts.cpu().tolist()
    assert n[0::2] == [n[0]] * 128 and n[1::2] == [n[1]] * 128
    for b in (2, 3, 254, 255):
        assert torch.equal(rows[b, :n[b]].view(torch.int32), rows[b & 1, :n[b & 1]].view(torch.int32)) and torch.equal(cells[b, :n[b]], cells[b & 1, :n[b & 1]])
    assert torch.equal(dec[254:256].view(torch.int32), dec[0:2].view(torch.int32))
    # (iii) the CPU algorithm on the same decoded tensors
    assert n[0] > 0 and n[1] > 0
    for b in (0, 1):
        ref = O.format_preds(dec[b].cpu(), obj_thresh=0.5, iou_thresh=0.5)
        assert ref.shape[0] == n[b]
        assert torch.equal(ref.view(torch.int32), rows[b, :n[b]].cpu().view(torch.int32))
    # (iv) the forward itself against the fp32 oracle (the bound of test_gpu_bf16.py's inference tests: 3e-2 of the output range)
    want_out = O.yogo_forward(x2, sd, O.arch("base_model", 7), 0.0425, 0.0555, inference=True)
    err = (dec[0:2].cpu() - want_out).abs().max().item()
    assert err <= 3e-2 * float(want_out.abs().max()), err


@pytest.mark.parametrize("num_classes,hw", [(7, (193, 258)), (1, (130, 70)), (11, (96, 128)), (7, (772, 1032))])
def test_eval_model_call_runs_head_and_decode_as_one_launch_bit_identical(num_classes, hw):
    """`model(x)` in eval mode on the bf16 path (yogo/model.py:275-313 under yogo/infer.py:313-317's autocast): the 1x1 head and the box
    decode in ONE launch (yogo_head1x1_decode_fwd_bf16, SURVEY.md 8(b) head1x1_decode_fwd) -- the bits of the two launches it replaces
    (conv_bf16_1x1_f32_kernel + decode_fwd_kernel, reached through forward_raw(x).decoded()), softmax and raw class channels both."""
    from yogo_amd import _hip as Hh
    from yogo_amd.model import YOGO

    torch.manual_seed(3)
    m = YOGO(hw, 0.0425, 0.0555, num_classes).cuda().eval()
    for k, v in m.state_dict().items():   # trained-like running statistics keep the activations finite
        if k.endswith("running_var"):
            v.fill_(900.0)
    x = torch.randint(0, 256, (2, 1, *hw), dtype=torch.uint8, generator=torch.Generator().manual_seed(4)).cuda()
    for inference in (True, False):
        m.inference = inference
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            Hh.launch_log(True)
            try:
                fused = m(x)
                torch.cuda.synchronize()
                log = Hh.read_launch_log()
            finally:
                Hh.launch_log(False)
            two = m.forward_raw(x).decoded()
        assert any(ln.startswith("conv_bf16_1x1_f32_kernel<8, true>") for ln in log), log
        assert not any(ln.startswith("decode_fwd_kernel") for ln in log), log
        assert fused.shape == two.shape == (2, 5 + num_classes, m.Sy, m.Sx)
        assert torch.isfinite(fused).all()
        assert torch.equal(fused.view(torch.int32), two.view(torch.int32)), float((fused - two).abs().max())
