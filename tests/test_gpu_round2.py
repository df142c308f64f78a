"""Round-2 regression tests of the HIP path: cache invalidation after training, input guards of the bf16 path, optimiser
state round trip, and canaries around every partial-sum buffer of the layer-0 kernels (exact-size buffers between sentinel
words that must survive; see DESIGN.md "The 03:52 abort")."""
import pytest
import torch

import yogo_oracle as O

pytestmark = pytest.mark.gpu


def H():
    from yogo_amd import _hip

    return _hip


def _model(hw=(96, 128), C=7, seed=0, **kw):
    from yogo_amd.model import YOGO

    torch.manual_seed(seed)
    m = YOGO(hw, 0.0425, 0.0555, C, **kw).cuda()
    m.train()
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout2d):
            mod.p = 0.0
    return m


# ---------------------------------------------------------------------------------------------------------------------
# stale bf16-inference cache (VERDICT r01 weak #9 / ADVICE): train -> bf16 eval -> train -> bf16 eval
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("half", [False, True])
def test_bf16_inference_follows_training(half):
    """the reference validates every 4th epoch between training epochs (yogo/train.py:341-342): an eval forward under bf16
    autocast after more optimisation steps must use the NEW weights and running statistics, i.e. equal a freshly built model
    loaded from the current state_dict"""
    from yogo_amd.model import YOGO
    from yogo_amd.train import HipTrainer
    from yogo_amd.yogo_loss import YOGOLoss

    hw, C, B = (96, 128), 7, 4
    m = _model(hw, C, seed=11)
    x = O.synthetic_images(B, *hw, seed=12).cuda()
    lab = O.synthetic_labels(B, m.Sx, m.Sy, K=6, num_classes=C, seed=13).cuda()
    tr = HipTrainer(m, YOGOLoss().cuda(), learning_rate=3e-3, total_steps=20, half=half)

    def infer(model):
        model.eval()
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            out = model(x).clone()
        model.train()
        return out

    def fresh():
        f = YOGO(hw, 0.0425, 0.0555, C).cuda()
        f.load_state_dict({k: v.clone() for k, v in m.state_dict().items()})
        return infer(f)

    tr.step(x, lab)
    o1 = infer(m)
    assert torch.equal(o1, fresh())
    for _ in range(3):
        tr.step(x, lab)
    o2 = infer(m)
    assert torch.equal(o2, fresh()), "bf16 inference ran on weights / statistics from before the last training steps"
    assert not torch.equal(o1, o2)
    # the fp32 eval path as well
    m.eval()
    with torch.no_grad():
        o32 = m(x).clone()
    m.train()
    f = YOGO(hw, 0.0425, 0.0555, C).cuda()
    f.load_state_dict({k: v.clone() for k, v in m.state_dict().items()})
    f.eval()
    with torch.no_grad():
        assert torch.equal(o32, f(x))


def test_bf16_training_input_guards():
    """forward_bf16_train checks the batch against the model before any kernel is launched (a 1-channel batch on an RGB model
    would make the layer-0 kernels read 3x the buffer)"""
    from yogo_amd import engine as E
    from yogo_amd.model import YOGO

    torch.manual_seed(0)
    rgb = YOGO((64, 96), 0.0425, 0.0555, 4, is_rgb=True).cuda().train()
    gray = YOGO((64, 96), 0.0425, 0.0555, 4).cuda().train()
    x1 = torch.zeros(2, 1, 64, 96, dtype=torch.uint8, device="cuda")
    x3 = torch.zeros(2, 3, 64, 96, dtype=torch.uint8, device="cuda")
    with pytest.raises(RuntimeError, match="expects 3 channels"):
        E.forward_bf16_train(E.get_engine(rgb.model), x1)
    with pytest.raises(RuntimeError, match="expects 1 channels"):
        E.forward_bf16_train(E.get_engine(gray.model), x3)
    with pytest.raises(RuntimeError, match=r"\[B,C,H,W\]"):
        E.forward_bf16_train(E.get_engine(gray.model), x1[0])
    # BatchNorm2d(momentum=None) (cumulative average) is refused on both paths instead of freezing the running statistics
    for mod in gray.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.momentum = None
    with pytest.raises(RuntimeError, match="momentum=None"):
        E.forward_bf16_train(E.get_engine(gray.model), x1)
    with pytest.raises(RuntimeError, match="momentum=None"):
        E.get_engine(gray.model).forward(x1, need_grad=True)


def test_autograd_gradient_is_not_overwritten():
    """a custom ModelDefn ending in a BatchNorm block: Engine.backward must not run the in-place BatchNorm backward on the
    gradient tensor autograd handed over"""
    from yogo_amd.engine import get_engine
    from yogo_amd.model_defns import HipBackbone

    torch.manual_seed(1)
    bb = HipBackbone(
        torch.nn.Sequential(torch.nn.Conv2d(1, 16, 3, stride=2, padding=1, bias=False), torch.nn.BatchNorm2d(16), torch.nn.LeakyReLU()),
        torch.nn.Sequential(torch.nn.Conv2d(16, 8, 3, padding=1), torch.nn.BatchNorm2d(8), torch.nn.LeakyReLU()),
    ).cuda().train()
    eng = get_engine(bb)
    x = torch.randint(0, 256, (2, 1, 32, 48), dtype=torch.uint8, device="cuda")
    raw, saved = eng.forward(x, need_grad=True)
    g = torch.randn_like(raw)
    g0 = g.clone()
    eng.backward(saved, g)
    assert torch.equal(g, g0)


def test_trainer_state_dict_round_trip():
    """HipTrainer.state_dict() has torch.optim.AdamW's layout (what the reference checkpoint stores as optimizer_state_dict,
    yogo/train.py:280-293): it loads into a torch AdamW over the same parameters, and a trainer restored from it continues
    bit-identically"""
    from yogo_amd.train import HipTrainer
    from yogo_amd.yogo_loss import YOGOLoss

    hw, C, B = (64, 96), 5, 2
    x = O.synthetic_images(B, *hw, seed=3).cuda()
    m1 = _model(hw, C, seed=4)
    lab = O.synthetic_labels(B, m1.Sx, m1.Sy, K=4, num_classes=C, seed=5).cuda()
    t1 = HipTrainer(m1, YOGOLoss().cuda(), total_steps=10)
    assert t1.state_dict()["state"] == {}
    t1.step(x, lab)
    t1.step(x, lab)
    osd = t1.state_dict()
    msd = {k: v.clone() for k, v in m1.state_dict().items()}
    opt = torch.optim.AdamW(m1.parameters(), lr=3e-4, weight_decay=5e-2)
    opt.load_state_dict(osd)                      # torch accepts the layout
    assert len(opt.state_dict()["state"]) == len(list(m1.parameters()))
    m2 = _model(hw, C, seed=99)
    m2.load_state_dict(msd)
    t2 = HipTrainer(m2, YOGOLoss().cuda(), total_steps=10)
    t2.load_state_dict(osd)
    assert t2.global_step == 2
    t1.step(x, lab)
    t2.step(x, lab)
    assert torch.equal(t1.flat.flat, t2.flat.flat)
    assert torch.equal(t1.flat.exp_avg_sq, t2.flat.exp_avg_sq)
    # a model moved / cast after the trainer was built no longer lives in the flat buffer: refused, not silently ignored
    m2.model[1][0].weight.data = m2.model[1][0].weight.data.clone()
    with pytest.raises(RuntimeError, match="flat buffer"):
        t2.step(x, lab)


# ---------------------------------------------------------------------------------------------------------------------
# canaries: every partial-sum / scratch buffer of the layer-0 kernels, exact size, between sentinel words
# ---------------------------------------------------------------------------------------------------------------------
GUARD = 4096


class Guarded:
    """n elements between two runs of GUARD sentinel words; the payload starts as NaN (floats) / 0x7F bytes so unwritten rows show"""

    def __init__(self, n, dtype=torch.float32):
        self.n = n
        self.sent = {torch.float32: 12345.678, torch.int32: 0x5A5A5A5A, torch.float64: -7.25}[dtype]
        self.buf = torch.full((n + 2 * GUARD,), self.sent, dtype=dtype, device="cuda")
        self.view = self.buf[GUARD:GUARD + n]
        if dtype.is_floating_point:
            self.view.fill_(float("nan"))
        else:
            self.view.fill_(0x7F7F7F7F)

    def check(self, what):
        torch.cuda.synchronize()
        lo, hi = self.buf[:GUARD], self.buf[GUARD + self.n:]
        assert bool((lo == self.sent).all()) and bool((hi == self.sent).all()), f"{what}: wrote outside its buffer"


@pytest.mark.parametrize("B,IH,IW", [(4, 96, 128), (2, 772, 1032), (1, 130, 258), (3, 20, 24)])
def test_layer0_partial_buffers_exact_size(B, IH, IW):
    """the abort of round 1 (gpurun_out/crash.log, 03:52) happened inside a HipTrainer.step(half=True) while the row counts of the
    layer-0 partial buffers were being changed (CFW_PPT 16 -> 32, per-wavefront -> per-workgroup statistics rows).  Every buffer
    whose size comes from a `*_rows` / `*_cols` query is allocated here at EXACTLY that size between sentinel words: the kernels
    must fill all of it (no NaN survives the reduction) and nothing beyond it."""
    h = H()
    st = h.stream_ptr()
    Cout, OH, OW = 16, IH // 2, IW // 2
    g = torch.Generator().manual_seed(IH + IW)
    x = torch.randint(0, 256, (B, 1, IH, IW), generator=g, dtype=torch.uint8).cuda()
    w = (torch.randn(Cout, 1, 3, 3, generator=g) * 0.05).cuda()
    # ---- Gram sweep ------------------------------------------------------------------------------------------------
    rows = h.query_ints("yogo_conv_first_gram_rows", 1, B, IH, IW)[0]
    gpart = Guarded(rows * 54, torch.int32)
    gram = Guarded(90, torch.float64)
    gram32 = Guarded(90)
    h.call("yogo_conv_first_gram", x, gpart.view, gram.view, gram32.view, B, IH, IW, st)
    gpart.check("gram partials"); gram.check("gram"); gram32.check("gram f32")
    assert torch.isfinite(gram.view).all() and torch.isfinite(gram32.view).all()
    # ---- statistics sweep of the matrix-core kernel ---------------------------------------------------------------------
    rows = h.query_ints("yogo_conv_first_mfma_stats_rows", 1, B, IH, IW)[0]
    stats = Guarded(rows * 16 * 2)
    h.call("yogo_conv_first_mfma", x, w, None, None, None, None, None, None, None, stats.view, B, Cout, IH, IW, 1, st)
    stats.check("conv_first_mfma statistics rows")
    assert torch.isfinite(stats.view).all(), "a statistics row was left unwritten"
    mean, invstd = Guarded(Cout), Guarded(Cout)
    rm, rv = torch.zeros(Cout, device="cuda"), torch.ones(Cout, device="cuda")
    nbt = torch.zeros(1, dtype=torch.int64, device="cuda")
    h.call("yogo_bn_finalize", stats.view, rows, 16, Cout, B * OH * OW, 1e-5, 0.1, mean.view, invstd.view, rm, rv, nbt, st)
    mean.check("bn_finalize mean"); invstd.check("bn_finalize invstd")
    assert torch.isfinite(mean.view).all() and torch.isfinite(invstd.view).all()
    # ---- z / y sweep ---------------------------------------------------------------------------------------------------------
    nz = B * 2 * OH * OW * 8
    zg, yg = Guarded(nz // 2), Guarded(nz // 2)      # bf16 tensors viewed through fp32 words
    z = zg.view.view(torch.bfloat16).view(B, 2, OH, OW, 8)
    y = yg.view.view(torch.bfloat16).view(B, 2, OH, OW, 8)
    gamma, beta = torch.ones(Cout, device="cuda"), torch.zeros(Cout, device="cuda")
    h.call("yogo_conv_first_mfma", x, w, None, z, y, mean.view, invstd.view, gamma, beta, None, B, Cout, IH, IW, 1, st)
    zg.check("conv_first_mfma z"); yg.check("conv_first_mfma y")
    assert torch.isfinite(z.float()).all() and torch.isfinite(y.float()).all()
    # ---- fused layer-0 backward: partial rows x cols, reduced sums, finalize ---------------------------------------------
    gy = (torch.randn(B, 2, OH, OW, 8, generator=g) * 0.1).to(torch.bfloat16).cuda()
    rows = h.query_ints("yogo_conv_first_wgrad_rows", 1, B, IH, IW, 2)[0]
    cols = h.query_ints("yogo_conv_first_bn_wgrad_cols", 1, 1, Cout)[0]
    for xg in ("", "_xg"):
        part, sums = Guarded(rows * cols), Guarded(cols)
        h.call("yogo_conv_first_bn_wgrad_bf16" + xg, x, 0, gy, z, mean.view, invstd.view, gamma, beta, part.view, B, 1, Cout, IH, IW, 2, 1, st)
        part.check("conv_first_bn_wgrad partial rows" + xg)
        h.call("yogo_partials_reduce", part.view, rows, cols, 0.0, sums.view, st)
        sums.check("partials_reduce sums" + xg)
        dw, dg, db = Guarded(Cout * 9), Guarded(Cout), Guarded(Cout)
        extra = (gram32.view,) if xg else ()
        h.call("yogo_conv_first_bn_wgrad_finalize" + xg, sums.view, *extra, mean.view, invstd.view, gamma, w, dw.view, dg.view, db.view, B, 1, Cout,
               IH, IW, 2, 1, 0.0, st)
        dw.check("dw"); dg.check("dgamma"); db.check("dbeta")
        assert torch.isfinite(dw.view).all() and torch.isfinite(dg.view).all() and torch.isfinite(db.view).all(), xg
    # ---- the separate-pass weight gradient (bf16 g) ----------------------------------------------------------------------------
    nj = 10
    part = Guarded(rows * Cout * nj)
    h.call("yogo_conv_first_wgrad_bf16g", x, 0, gy, part.view, B, 1, Cout, IH, IW, 2, st)
    part.check("conv_first_wgrad partial rows")
    red = Guarded(Cout * nj)
    h.call("yogo_partials_reduce", part.view, rows, Cout * nj, 0.0, red.view, st)
    red.check("partials_reduce")
    assert torch.isfinite(red.view).all()
    # the image itself is read with 16-bit pair loads (FAST path): a batch that ENDS at its allocation's last byte
    xt = torch.randint(0, 256, (B * IH * IW + GUARD,), generator=g, dtype=torch.uint8).cuda()
    x_end = xt[GUARD:].view(B, 1, IH, IW)            # nothing behind the last image
    part2 = Guarded(rows * cols)
    h.call("yogo_conv_first_bn_wgrad_bf16_xg", x_end, 0, gy, z, mean.view, invstd.view, gamma, beta, part2.view, B, 1, Cout, IH, IW, 2, 1, st)
    part2.check("conv_first_bn_wgrad on the last image of an allocation")


def test_host_tensor_at_the_boundary_is_an_error_not_a_fault():
    """a host tensor handed to an entry point would make a kernel read host memory (a GPU memory fault that aborts the process --
    how a `.cuda()`-moved model whose resize_model rebuilt its grids on the stale `self.device` died in round 2): refused in _hip.call"""
    from yogo_amd.model import YOGO

    h = H()
    var, out = torch.ones(4), torch.empty(4, device="cuda")
    with pytest.raises(RuntimeError, match="host tensor"):
        h.call("yogo_bn_invstd", var, 1e-5, out, 4, h.stream_ptr())
    # the regression itself: .cuda() does not update YOGO.device (only .to() does, as in the reference); resize_model must
    # rebuild the grids where the old ones live
    net = YOGO((128, 160), 0.0425, 0.0555, 3, inference=True).cuda().eval()
    net.resize_model(64)
    assert net._Cxs.is_cuda and net._Cys.is_cuda and net.height_multiplier.is_cuda and (net.Sx, net.Sy) == (20, 8)
    with torch.no_grad():
        y = net(torch.zeros(1, 1, 64, 160, dtype=torch.uint8, device="cuda"))
    assert tuple(y.shape) == (1, 8, 8, 20) and bool(torch.isfinite(y).all())


@pytest.mark.parametrize("shape", [(3, 12, 25, 33), (2, 7, 9, 11), (1, 69, 5, 6), (2, 12, 97, 129)])
def test_fused_decode_loss_backward_is_the_three_calls(shape):
    """yogo_decode_loss_bwd_bf16 (the trainer's one-pass form) against yogo_decode_fwd + yogo_loss_fwd_bwd + yogo_decode_bwd_bf16:
    the same statements in one kernel -> the gradient units and the four loss values are identical bit for bit.  Labels with
    and without objects, boxes that leave the unit square (clamp derivatives), raw widths beyond the exp clamp."""
    from yogo_amd import _hip as h

    B, P, Sy, Sx = shape
    C = P - 5
    g = torch.Generator().manual_seed(B * 1000 + P)
    raw = torch.randn(B, P, Sy, Sx, generator=g) * 1.5
    raw[:, 2:4] += 1.0
    raw[0, 2, 0, 0] = 90.0   # beyond the exp clamp of the decode
    lab = torch.zeros(B, 6, Sy, Sx)
    obj = torch.rand(B, Sy, Sx, generator=g) < 0.3
    lab[:, 0] = obj.float()
    x1 = torch.rand(B, Sy, Sx, generator=g) * 0.8
    y1 = torch.rand(B, Sy, Sx, generator=g) * 0.8
    lab[:, 1], lab[:, 2] = x1, y1
    lab[:, 3] = x1 + 0.02 + torch.rand(B, Sy, Sx, generator=g) * 0.15
    lab[:, 4] = y1 + 0.02 + torch.rand(B, Sy, Sx, generator=g) * 0.15
    lab[:, 5] = torch.randint(0, C, (B, Sy, Sx), generator=g).float()
    raw, lab = raw.cuda(), lab.cuda()
    cxs = (torch.arange(Sx, dtype=torch.float32) / Sx).expand(Sy, Sx).contiguous().cuda()
    cys = (torch.arange(Sy, dtype=torch.float32) / Sy)[:, None].expand(Sy, Sx).contiguous().cuda()
    aw, ah, wm, hm = 0.0425, 0.0555, 1.0, 1.3
    w_no, w_iou, w_cls, ls = 0.5, 5.0, 1.0, 0.01
    st = h.stream_ptr()
    nws = h.query_size("yogo_loss_workspace_bytes", B, Sy, Sx) // 4
    # the three calls
    pred = torch.empty_like(raw)
    h.call("yogo_decode_fwd", raw, pred, cxs, cys, B, P, Sy, Sx, aw, ah, wm, hm, 0, st)
    gpred, out3, ws3 = torch.empty_like(raw), torch.empty(4, device="cuda"), torch.empty(nws, device="cuda")
    h.call("yogo_loss_fwd_bwd", pred, lab, gpred, out3, ws3, B, P, Sy, Sx, w_no, w_iou, w_cls, ls, st)
    Pb = ((P + 15) // 16) * 2
    g3 = torch.full((B, Pb, Sy, Sx, 8), float("nan"), dtype=torch.bfloat16, device="cuda")
    h.call("yogo_decode_bwd_bf16", raw, pred, gpred, g3, B, P, Sy, Sx, 0, st)
    # the fused call
    g1 = torch.full((B, Pb, Sy, Sx, 8), float("nan"), dtype=torch.bfloat16, device="cuda")
    out1, ws1 = torch.empty(4, device="cuda"), torch.empty(nws, device="cuda")
    h.call("yogo_decode_loss_bwd_bf16", raw, lab, cxs, cys, g1, out1, ws1, B, P, Sy, Sx, aw, ah, wm, hm, w_no, w_iou, w_cls, ls, st)
    torch.cuda.synchronize()
    assert torch.equal(g1.view(torch.int16), g3.view(torch.int16))
    assert torch.equal(out1, out3) and bool(torch.isfinite(out1).all())
    assert float(g1.float().abs().max()) > 0
