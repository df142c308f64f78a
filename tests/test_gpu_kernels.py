"""Kernel-level parity on a real MI355X: every C-ABI compute entry point against plain torch fp32 on the CPU
(the oracle's building blocks), over shapes that exercise tiling edges (odd sizes, narrow bands, channel padding)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import yogo_oracle as O
from _util import rel_err

pytestmark = pytest.mark.gpu


def _hip():
    from yogo_amd import _hip

    return _hip


def dev(t):
    return t.cuda().contiguous()


def conv_fwd_hip(x, w, b, stride, act=0, mask=None, want_pre=False, want_stats=False):
    H = _hip()
    B, Cin, IH, IW = x.shape
    Cout, _, k, _ = w.shape
    pad = 1 if k == 3 else 0
    OH, OW = (IH + 2 * pad - k) // stride + 1, (IW + 2 * pad - k) // stride + 1
    st = H.stream_ptr()
    nbytes = H.query_size("yogo_conv_packed_bytes", Cin, Cout, k, stride, 0)
    packed = torch.empty(nbytes // 4, device="cuda")
    wd = dev(w)
    H.call("yogo_conv_pack_f32", wd, packed, Cin, Cout, k, stride, 0, st)
    out = torch.full((B, Cout, OH, OW), float("nan"), device="cuda")
    pre = torch.full_like(out, float("nan")) if want_pre else None
    stats = None
    rows = mpad = 0
    if want_stats:
        rows, mpad = H.query_ints("yogo_conv2d_fwd_stats_shape", 2, B, Cin, Cout, IH, IW, k, stride)
        stats = torch.full((rows, mpad, 2), float("nan"), device="cuda")
    H.call("yogo_conv2d_fwd_f32", dev(x), packed, dev(b) if b is not None else None, out, pre, dev(mask) if mask is not None else None,
           stats, B, Cin, Cout, IH, IW, k, stride, act, st)
    torch.cuda.synchronize()
    return out.cpu(), (pre.cpu() if want_pre else None), (stats.cpu() if want_stats else None)


CONV_CASES = [
    # B, Cin, Cout, IH, IW, k, stride
    (2, 16, 32, 20, 37, 3, 1),
    (1, 32, 64, 21, 40, 3, 2),
    (2, 64, 128, 13, 19, 3, 1),
    (1, 128, 128, 25, 33, 3, 2),
    (2, 128, 128, 9, 129, 3, 1),
    (1, 128, 12, 7, 11, 1, 1),
    (1, 6, 10, 11, 9, 3, 1),
    (1, 48, 96, 8, 70, 3, 2),
    (1, 128, 256, 6, 9, 3, 1),
    (1, 16, 32, 3, 600, 3, 1),
    (1, 32, 64, 5, 1032, 3, 2),
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd(case):
    B, Cin, Cout, IH, IW, k, s = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    x = torch.randn(B, Cin, IH, IW, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / np.sqrt(Cin * k * k)
    b = torch.randn(Cout, generator=g)
    mask = (torch.rand(B, Cout, generator=g) > 0.3).float() * 1.5
    ref_pre = F.conv2d(x, w, b, stride=s, padding=1 if k == 3 else 0)
    for act, name in ((0, None), (1, "leaky"), (2, "silu")):
        out, pre, stats = conv_fwd_hip(x, w, b, s, act=act, mask=mask, want_pre=True, want_stats=True)
        ref = O._act(ref_pre, name) * mask[:, :, None, None]
        assert not torch.isnan(out).any()
        assert rel_err(out, ref) < 2e-5, (case, act)
        assert rel_err(pre, ref_pre) < 2e-5
        ssum = stats.double().sum(0)[:Cout]
        torch.testing.assert_close(ssum[:, 0], ref_pre.double().sum((0, 2, 3)), rtol=1e-4, atol=1e-3)
        torch.testing.assert_close(ssum[:, 1], (ref_pre.double() ** 2).sum((0, 2, 3)), rtol=1e-4, atol=1e-3)
    out, _, _ = conv_fwd_hip(x, w, None, s)
    assert rel_err(out, F.conv2d(x, w, None, stride=s, padding=1 if k == 3 else 0)) < 2e-5


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_dgrad_wgrad(case):
    H = _hip()
    B, Cin, Cout, IH, IW, k, s = case
    g = torch.Generator().manual_seed(7 + hash(case) % 1000)
    x = torch.randn(B, Cin, IH, IW, generator=g, requires_grad=True)
    w = (torch.randn(Cout, Cin, k, k, generator=g) / np.sqrt(Cin * k * k)).requires_grad_(True)
    b = torch.randn(Cout, generator=g, requires_grad=True)
    y = F.conv2d(x, w, b, stride=s, padding=1 if k == 3 else 0)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    st = H.stream_ptr()
    # dgrad with a fused activation derivative + channel mask
    ref_y = torch.randn(B, Cin, IH, IW, generator=g)
    mask = (torch.rand(B, Cin, generator=g) > 0.3).float() * 1.25
    nbytes = H.query_size("yogo_conv_packed_bytes", Cin, Cout, k, s, 1)
    packed = torch.empty(nbytes // 4, device="cuda")
    H.call("yogo_conv_pack_f32", dev(w.detach()), packed, Cin, Cout, k, s, 1, st)
    dx = torch.full((B, Cin, IH, IW), float("nan"), device="cuda")
    H.call("yogo_conv2d_dgrad_f32", dev(gy), packed, dx, None, 0, None, B, Cin, Cout, IH, IW, k, s, st)
    assert rel_err(dx.cpu(), x.grad) < 2e-5, case
    dx.fill_(float("nan"))
    H.call("yogo_conv2d_dgrad_f32", dev(gy), packed, dx, dev(ref_y), 1, dev(mask), B, Cin, Cout, IH, IW, k, s, st)
    want = x.grad * torch.where(ref_y > 0, 1.0, 0.01) * mask[:, :, None, None]
    assert rel_err(dx.cpu(), want) < 2e-5, case
    dx.fill_(float("nan"))
    H.call("yogo_conv2d_dgrad_f32", dev(gy), packed, dx, dev(ref_y), 2, None, B, Cin, Cout, IH, IW, k, s, st)
    sg = torch.sigmoid(ref_y)
    assert rel_err(dx.cpu(), x.grad * (sg * (1 + ref_y * (1 - sg)))) < 2e-5, case
    # wgrad + bias grad, unclamped and clamped
    wsb = H.query_size("yogo_conv2d_wgrad_workspace_bytes", B, Cin, Cout, IH, IW, k, s)
    ws = torch.empty(wsb // 4, device="cuda")
    for clip in (0.0, 0.05):
        dw = torch.full((Cout, Cin, k, k), float("nan"), device="cuda")
        db = torch.full((Cout,), float("nan"), device="cuda")
        H.call("yogo_conv2d_wgrad_f32", dev(x.detach()), dev(gy), dw, db, ws, B, Cin, Cout, IH, IW, k, s, clip, st)
        wg, bg = w.grad, b.grad
        if clip > 0:
            wg, bg = wg.clamp(-clip, clip), bg.clamp(-clip, clip)
        # tolerance relative to the UNCLAMPED magnitude (the clamp only removes the large values)
        assert float((dw.cpu() - wg).abs().max()) < 3e-5 * float(w.grad.abs().max()), (case, clip)
        assert float((db.cpu() - bg).abs().max()) < 3e-5 * float(b.grad.abs().max()), (case, clip)


@pytest.mark.parametrize("cin,cout,stride,u8", [(1, 16, 2, True), (3, 4, 2, True), (1, 48, 2, False), (3, 20, 1, False)])
def test_conv_first(cin, cout, stride, u8):
    H = _hip()
    B, IH, IW = 2, 37, 53
    g = torch.Generator().manual_seed(3)
    xi = torch.randint(0, 256, (B, cin, IH, IW), dtype=torch.uint8, generator=g)
    x = xi if u8 else xi.float() / 7.0
    w = (torch.randn(cout, cin, 3, 3, generator=g) / 3).requires_grad_(True)
    b = torch.randn(cout, generator=g, requires_grad=True)
    ref = F.conv2d(x.float(), w, b, stride=stride, padding=1)
    OH, OW = ref.shape[2:]
    st = H.stream_ptr()
    rows = H.query_ints("yogo_conv_first_stats_rows", 1, B, IH, IW, stride)[0]
    out = torch.full(ref.shape, float("nan"), device="cuda")
    pre = torch.full(ref.shape, float("nan"), device="cuda")
    stats = torch.full((rows, cout, 2), float("nan"), device="cuda")
    H.call("yogo_conv_first_fwd", dev(x), 0 if u8 else 1, dev(w.detach()), dev(b.detach()), out, pre, None, stats, B, cin, cout, IH, IW,
           stride, 1, st)
    assert rel_err(pre.cpu(), ref) < 1e-5
    assert rel_err(out.cpu(), F.leaky_relu(ref, 0.01)) < 1e-5
    s = stats.cpu().double().sum(0)
    torch.testing.assert_close(s[:, 0], ref.detach().double().sum((0, 2, 3)), rtol=1e-5, atol=1e-2)
    torch.testing.assert_close(s[:, 1], (ref.detach().double() ** 2).sum((0, 2, 3)), rtol=1e-5, atol=1e-2)
    gy = torch.randn(ref.shape, generator=g)
    ref.backward(gy)
    nj = cin * 9 + 1
    rows = H.query_ints("yogo_conv_first_wgrad_rows", 1, B, IH, IW, stride)[0]
    part = torch.empty(rows * cout * nj, device="cuda")
    H.call("yogo_conv_first_wgrad", dev(x), 0 if u8 else 1, dev(gy), part, B, cin, cout, IH, IW, stride, st)
    red = torch.empty(cout, nj, device="cuda")
    H.call("yogo_partials_reduce", part, rows, cout * nj, 0.0, red, st)
    red = red.cpu()
    assert rel_err(red[:, : nj - 1].reshape(w.shape), w.grad) < 2e-5
    assert rel_err(red[:, nj - 1], b.grad) < 2e-5


@pytest.mark.parametrize("shape", [(3, 16, 20, 36), (2, 128, 7, 9), (4, 5, 11, 13)])
def test_batchnorm_train_and_backward(shape):
    H = _hip()
    B, C, Hh, W = shape
    g = torch.Generator().manual_seed(5)
    z = (torch.randn(shape, generator=g) * 40 + 300).requires_grad_(True)
    gamma = (1 + 0.1 * torch.randn(C, generator=g)).requires_grad_(True)
    beta = (0.1 * torch.randn(C, generator=g)).requires_grad_(True)
    rm, rv = torch.randn(C, generator=g), torch.rand(C, generator=g) + 0.5
    rm_ref, rv_ref = rm.clone(), rv.clone()
    y_ref = F.leaky_relu(F.batch_norm(z, rm_ref, rv_ref, gamma, beta, training=True, momentum=0.1, eps=1e-5), 0.01)
    gy = torch.randn(shape, generator=g)
    y_ref.backward(gy)
    st = H.stream_ptr()
    HW = Hh * W
    # partial sums the way a conv epilogue would have produced them: 3 row-chunks
    zz = z.detach()
    chunks = torch.chunk(zz.permute(1, 0, 2, 3).reshape(C, -1), 3, dim=1)
    part = torch.stack([torch.stack([c.sum(1), (c * c).sum(1)], dim=1) for c in chunks]).contiguous()  # [3][C][2]
    mean = torch.empty(C, device="cuda")
    invstd = torch.empty(C, device="cuda")
    rmd, rvd = dev(rm), dev(rv)
    nbt = torch.zeros((), dtype=torch.long, device="cuda")
    H.call("yogo_bn_finalize", dev(part), 3, C, C, B * HW, 1e-5, 0.1, mean, invstd, rmd, rvd, nbt, st)
    zd = dev(zz)
    y = torch.empty_like(zd)
    H.call("yogo_bn_apply_act", zd, y, mean, invstd, 0, 1e-5, dev(gamma.detach()), dev(beta.detach()), B, C, HW, 1, st)
    torch.testing.assert_close(y.cpu(), y_ref.detach(), rtol=2e-4, atol=2e-4)
    torch.testing.assert_close(rmd.cpu(), rm_ref, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(rvd.cpu(), rv_ref, rtol=1e-3, atol=1e-3)
    assert int(nbt.item()) == 1
    # backward: g is the gradient w.r.t. the block output; the kernels apply LeakyReLU' from the recomputed pre-activation
    gbn = dev(gy)
    rows = H.query_ints("yogo_bn_bwd_rows", 1, B, HW)[0]
    partb = torch.empty(rows * C * 2, device="cuda")
    sums = torch.empty(2 * C, device="cuda")
    dgamma, dbeta = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    dz = torch.empty_like(zd)
    H.call("yogo_bn_bwd", gbn, zd, dz, mean, invstd, dev(gamma.detach()), dev(beta.detach()), 1, dgamma, dbeta, partb, sums, B, C, HW,
           1, 0.0, st)
    assert rel_err(dz.cpu(), z.grad) < 1e-3
    assert rel_err(dgamma.cpu(), gamma.grad) < 1e-3
    assert rel_err(dbeta.cpu(), beta.grad) < 1e-4
    # eval-mode apply from running stats
    y2 = torch.empty_like(zd)
    H.call("yogo_bn_apply_act", zd, y2, rmd, rvd, 1, 1e-5, dev(gamma.detach()), dev(beta.detach()), B, C, HW, 0, st)
    ref2 = F.batch_norm(zz, rm_ref, rv_ref, gamma.detach(), beta.detach(), training=False, eps=1e-5)
    torch.testing.assert_close(y2.cpu(), ref2, rtol=2e-4, atol=2e-3)


def test_adamw_matches_oracle():
    H = _hip()
    g = torch.Generator().manual_seed(9)
    n = 10007
    p = torch.randn(n, generator=g)
    m = torch.zeros(n)
    v = torch.zeros(n)
    pd, md, vd = dev(p), dev(m), dev(v)
    for step in range(1, 6):
        gr = torch.randn(n, generator=g) * 0.1
        lr = O.cosine_lr(step - 1, 3e-4, 100, 3e-5)
        p, m, v = O.adamw_step(p, gr, m, v, step, lr)
        H.call("yogo_adamw_step", pd, dev(gr), md, vd, n, step, lr, 0.9, 0.999, 1e-8, 5e-2, 1.0, H.stream_ptr())
        torch.testing.assert_close(pd.cpu(), p, rtol=1e-6, atol=1e-7)
        torch.testing.assert_close(vd.cpu(), v, rtol=1e-6, atol=1e-12)
