"""bf16 path (NCHW8c activations, bf16 MFMA, fp32 accumulation / statistics / parameter gradients) against fp32 references.
Tolerances are bf16's: 8 significand bits -> ~4e-3 relative rounding per stored value; stated per check."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import yogo_oracle as O
from _util import load_net_fixture, rel_err

pytestmark = pytest.mark.gpu


def H():
    from yogo_amd import _hip

    return _hip


def to8c(t):
    h = H()
    B, C, Hh, W = t.shape
    cb = h.lib().yogo_bf16_channel_blocks(C)
    out = torch.empty(B, cb, Hh, W, 8, dtype=torch.bfloat16, device="cuda")
    h.call("yogo_nchw_f32_to_bf16_8c", t.cuda().contiguous().float(), out, B, C, Hh * W, h.stream_ptr())
    return out


def from8c(t, C):
    h = H()
    B, cb, Hh, W, _ = t.shape
    out = torch.empty(B, C, Hh, W, device="cuda")
    h.call("yogo_bf16_8c_to_nchw_f32", t, out, B, C, Hh * W, h.stream_ptr())
    return out.cpu()


def bf(t):
    return t.to(torch.bfloat16).float()


def sign_map(y8, C):
    """include/yogo_hip.h, yogo_bf16_signs_bytes: [B][2][H][W][Cpad/16] bytes; byte (h, pixel, q), bit i + 4e = (channel 4h + i of
    channel block 2q + e > 0)"""
    B, Mb, Hh, W, _ = y8.shape
    cpad = 32 if C <= 32 else (64 if C <= 64 else (C + 127) // 128 * 128)
    pos = torch.zeros(B, cpad // 8, Hh, W, 8, dtype=torch.int64)
    pos[:, :Mb] = (y8 > 0)
    pos = pos.view(B, cpad // 16, 2, Hh, W, 2, 4).permute(0, 5, 3, 4, 1, 2, 6)   # [B][h][H][W][q][e][i]
    wts = torch.tensor([[1, 2, 4, 8], [16, 32, 64, 128]])
    return (pos * wts).sum((-1, -2)).to(torch.uint8).reshape(-1)


def sign_bytes_used(t, C):
    """the bytes of a sign map that belong to existing channel blocks (bytes of padding blocks are unspecified)"""
    cpad = 32 if C <= 32 else (64 if C <= 64 else (C + 127) // 128 * 128)
    return t.view(-1, cpad // 16)[:, : (C + 15) // 16]


def test_layout_round_trip():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 12, 7, 9, generator=g)
    y = to8c(x)
    assert y.shape == (2, 2, 7, 9, 8)
    assert torch.equal(from8c(y, 12), bf(x))
    # the layout itself: unit (b, cb, h, w) holds channels cb*8 .. cb*8+7, padding channels are zero
    ref = torch.zeros(2, 16, 7, 9)
    ref[:, :12] = bf(x)
    assert torch.equal(y.cpu().float(), ref.view(2, 2, 8, 7, 9).permute(0, 1, 3, 4, 2))


CASES = [(2, 16, 32, 20, 37, 3, 1), (1, 32, 64, 21, 40, 3, 2), (2, 64, 128, 13, 19, 3, 1), (1, 128, 128, 25, 33, 3, 2),
         (2, 128, 128, 9, 129, 3, 1), (1, 128, 12, 7, 11, 1, 1), (1, 24, 40, 11, 9, 3, 1), (1, 16, 32, 3, 600, 3, 1), (1, 16, 32, 70, 45, 3, 1),
         # round 2: the chunk ring of the 128-channel stride-2 data gradient (tails, tiny tiles, several images) and the unrolled
         # step loop of the 4-wavefront tiles (one and two chunks, stride 1 and 2, odd sizes)
         (3, 128, 128, 7, 5, 3, 2), (2, 128, 128, 64, 70, 3, 2), (1, 256, 128, 31, 29, 3, 2), (2, 32, 64, 33, 35, 3, 2), (1, 32, 32, 50, 37, 3, 1),
         (2, 16, 16, 41, 23, 3, 1),
         # weight-gradient step loops unrolled per steps-per-row count (16 / 32 / 48 / 64 staged columns), pixel-split and
         # two-co-block tilings, tall (8-row) and short units
         (1, 16, 32, 9, 12, 3, 1), (1, 16, 32, 66, 60, 3, 1), (2, 64, 128, 10, 14, 3, 1), (1, 64, 128, 7, 64, 3, 1), (2, 32, 64, 17, 30, 3, 1),
         (1, 128, 128, 12, 60, 3, 2),
         # round 3: row-wise LDS-DMA staging of the lean 4-wavefront tiles -- widths around the 64-unit row pitch (one band of 62
         # columns, 63 -> two bands, narrow last bands), several tiles per band, stride 2 with bands of at most 31 columns
         (1, 16, 32, 40, 62, 3, 1), (1, 16, 32, 40, 63, 3, 1), (2, 32, 16, 23, 125, 3, 1), (1, 16, 32, 33, 130, 3, 1),
         (1, 32, 64, 30, 124, 3, 2), (1, 32, 64, 31, 127, 3, 2)]


@pytest.mark.parametrize("case", CASES)
def test_conv_bf16_fwd_dgrad_wgrad(case):
    h = H()
    B, Cin, Cout, IH, IW, k, s = case
    g = torch.Generator().manual_seed(hash(case) % 997)
    x = bf(torch.randn(B, Cin, IH, IW, generator=g)).requires_grad_(True)
    w = (torch.randn(Cout, Cin, k, k, generator=g) / np.sqrt(Cin * k * k)).requires_grad_(True)
    b = torch.randn(Cout, generator=g, requires_grad=True)
    wb = bf(w.detach())
    pad = 1 if k == 3 else 0
    ref_pre = F.conv2d(x, wb.requires_grad_(True), b, stride=s, padding=pad)
    OH, OW = ref_pre.shape[2:]
    st = h.stream_ptr()
    mask = (torch.rand(B, Cout, generator=g) > 0.3).float() * 1.5
    # ---- forward: bias + leaky + channel mask + BatchNorm partial sums ----------------------------------------------
    packed = torch.empty(h.query_size("yogo_conv_bf16_packed_bytes", Cin, Cout, k, 0), dtype=torch.uint8, device="cuda")
    h.call("yogo_conv_bf16_pack", w.detach().cuda(), None, packed, Cin, Cout, k, 0, st)
    rows, mpad = h.query_ints("yogo_conv2d_fwd_bf16_stats_shape", 2, B, Cin, Cout, IH, IW, k, s)
    stats = torch.full((rows, mpad, 2), float("nan"), device="cuda")
    out = torch.full((B, h.lib().yogo_bf16_channel_blocks(Cout), OH, OW, 8), float("nan"), dtype=torch.bfloat16, device="cuda")
    x8 = to8c(x.detach())
    h.call("yogo_conv2d_fwd_bf16", x8, packed, b.detach().cuda(), out, None, mask.cuda(), stats, B, Cin, Cout, IH, IW, k, s, 1, st)
    want = F.leaky_relu(ref_pre.detach(), 0.01) * mask[:, :, None, None]
    got = from8c(out, Cout)
    assert rel_err(got, want) < 8e-3, case           # one bf16 rounding of the output
    ssum = stats.cpu().double().sum(0)[:Cout]
    torch.testing.assert_close(ssum[:, 0], ref_pre.detach().double().sum((0, 2, 3)), rtol=1e-3, atol=2e-2)
    torch.testing.assert_close(ssum[:, 1], (ref_pre.detach().double() ** 2).sum((0, 2, 3)), rtol=1e-3, atol=2e-2)
    # padding channels of the output tensor are zero
    padc = out.cpu().float().view(B, -1, OH, OW, 8).permute(0, 1, 4, 2, 3).reshape(B, -1, OH, OW)[:, Cout:]
    assert padc.numel() == 0 or float(padc.abs().max()) == 0.0
    # fp32 NCHW output mode (the head)
    o32 = torch.full((B, Cout, OH, OW), float("nan"), device="cuda")
    h.call("yogo_conv2d_fwd_bf16", x8, packed, b.detach().cuda(), None, o32, None, None, B, Cin, Cout, IH, IW, k, s, 0, st)
    assert rel_err(o32.cpu(), ref_pre.detach()) < 2e-5 * max(1.0, np.sqrt(Cin * k * k) / 8), case
    # ---- backward ---------------------------------------------------------------------------------------------------------
    gy = bf(torch.randn(ref_pre.shape, generator=g))
    ref_pre.backward(gy)
    gy8 = to8c(gy)
    dmode = 2 if (s == 2 and k == 3) else 1   # stride-2 data gradients take their weight slices in parity-class order
    pd = torch.empty(h.query_size("yogo_conv_bf16_packed_bytes", Cin, Cout, k, dmode), dtype=torch.uint8, device="cuda")
    h.call("yogo_conv_bf16_pack", w.detach().cuda(), None, pd, Cin, Cout, k, dmode, st)
    dx = torch.full((B, h.lib().yogo_bf16_channel_blocks(Cin), IH, IW, 8), float("nan"), dtype=torch.bfloat16, device="cuda")
    refy = bf(torch.randn(B, Cin, IH, IW, generator=g))
    cmask = (torch.rand(B, Cin, generator=g) > 0.3).float() * 1.25
    h.call("yogo_conv2d_dgrad_bf16", gy8, pd, dx, None, 0, None, B, Cin, Cout, IH, IW, k, s, st)
    assert rel_err(from8c(dx, Cin), x.grad) < 8e-3, case
    h.call("yogo_conv2d_dgrad_bf16", gy8, pd, dx, to8c(refy), 1, cmask.cuda(), B, Cin, Cout, IH, IW, k, s, st)
    want = x.grad * torch.where(refy > 0, 1.0, 0.01) * cmask[:, :, None, None]
    assert rel_err(from8c(dx, Cin), want) < 8e-3, case
    # LeakyReLU reference as a sign map: the forward writes it next to the (unchanged) output, the data gradient reads it in
    # place of the reference tensor -- both bit-identical to the bf16-reference route
    out_s = torch.full_like(out, float("nan"))
    sg = torch.full((h.query_size("yogo_bf16_signs_bytes", B, Cout, OH, OW),), 0xA5, dtype=torch.uint8, device="cuda")
    h.call("yogo_conv2d_fwd_bf16_signs", x8, packed, b.detach().cuda(), out_s, sg, mask.cuda(), B, Cin, Cout, IH, IW, k, s, 1, st)
    out_p = torch.full_like(out, float("nan"))   # (the launch above took the order of operations of the BatchNorm sums)
    h.call("yogo_conv2d_fwd_bf16", x8, packed, b.detach().cuda(), out_p, None, mask.cuda(), None, B, Cin, Cout, IH, IW, k, s, 1, st)
    assert torch.equal(out_s.view(torch.int16), out_p.view(torch.int16)), case
    assert torch.equal(sign_bytes_used(sg.cpu(), Cout), sign_bytes_used(sign_map(out_p.cpu().float(), Cout), Cout)), case
    dx_ref = dx.clone()
    h.launch_log(True)
    try:
        h.call("yogo_conv2d_dgrad_bf16_signs", gy8, pd, dx, sign_map(to8c(refy).cpu().float(), Cin).cuda(), cmask.cuda(), B, Cin, Cout, IH, IW, k, s, st)
        torch.cuda.synchronize()
        log_signs = h.read_launch_log()
    finally:
        h.launch_log(False)
    # bit-identity of the two routes is the contract wherever they run on the same kernel; the one-ulp branch below is taken ONLY when the
    # launch log shows that the sign-map route really ran on the direct stride-2 kernel (include/yogo_hip.h: yogo_conv2d_dgrad_bf16_signs)
    on_direct = any(ln.startswith("conv_bf16_s2d_direct_kernel") for ln in log_signs)
    if not on_direct:
        assert torch.equal(dx.view(torch.int16), dx_ref.view(torch.int16)), (case, log_signs)
    if not torch.equal(dx.view(torch.int16), dx_ref.view(torch.int16)):
        # the two routes may run on different kernels (stride-2 data gradients into <= 32 channels: the sign-map route takes the direct kernel
        # -- 16-channel steps --, the bf16-reference route the tiled one -- 32 / 64-channel chunks): another fp32 summation order, the bf16
        # results within one rounding step of each other on a few values
        a_, b_ = dx.float(), dx_ref.float()
        ulp = 2.0 ** -7 * torch.maximum(a_.abs(), b_.abs()) + 1e-6 * a_.abs().max()
        assert s == 2 and on_direct and bool(((a_ - b_).abs() <= ulp).all()) and (a_ != b_).float().mean().item() < 5e-3, case
    # wgrad from bf16 inputs is exact fp32 MFMA on the widened values
    ws = torch.empty(h.query_size("yogo_conv2d_wgrad_workspace_bytes", B, Cin, Cout, IH, IW, k, s) // 4, device="cuda")
    dw = torch.full((Cout, Cin, k, k), float("nan"), device="cuda")
    db = torch.full((Cout,), float("nan"), device="cuda")
    h.call("yogo_conv2d_wgrad_bf16in", x8, gy8, dw, db, ws, B, Cin, Cout, IH, IW, k, s, 0.0, st)
    assert rel_err(dw.cpu(), wb.grad) < 3e-5, case
    assert rel_err(db.cpu(), b.grad) < 3e-5, case
    # ... and on the bf16 matrix cores (contraction over pixels via transposed LDS reads): products of bf16 values are exact in
    # fp32, only the accumulation order differs
    ws2 = torch.empty(h.query_size("yogo_conv2d_wgrad_bf16_workspace_bytes", B, Cin, Cout, IH, IW, k, s) // 4, device="cuda")
    for clip in (0.0, 0.05):
        dw2 = torch.full((Cout, Cin, k, k), float("nan"), device="cuda")
        db2 = torch.full((Cout,), float("nan"), device="cuda")
        h.call("yogo_conv2d_wgrad_bf16", x8, gy8, dw2, db2, ws2, B, Cin, Cout, IH, IW, k, s, clip, st)
        wg, bg = wb.grad, b.grad
        if clip > 0:
            wg, bg = wg.clamp(-clip, clip), bg.clamp(-clip, clip)
        assert float((dw2.cpu() - wg).abs().max()) < 1e-4 * float(wb.grad.abs().max()), (case, clip)
        assert float((db2.cpu() - bg).abs().max()) < 1e-4 * float(b.grad.abs().max()), (case, clip)


@pytest.mark.parametrize("case", [(1, 128, 32, 9, 12, 3, 1), (2, 128, 24, 11, 20, 3, 1), (1, 128, 128, 9, 17, 3, 1), (2, 16, 32, 9, 12, 3, 1),
                                  (1, 128, 12, 7, 11, 1, 1), (1, 160, 32, 8, 16, 1, 1)])
def test_wgrad_bf16_workspace_is_large_enough(case):
    """the workspace yogo_conv2d_wgrad_bf16_workspace_bytes asks for holds everything the launch writes: slabs and the bias
    partial rows of EVERY tiling (the lean KS = 1 tilings write one row per wavefront sharing a gradient operand: up to
    NBW * 3 = 12 rows per slab for <= 32 output channels x >= 97 input channels) -- a poisoned guard region directly behind
    the workspace must survive the launch, and the gradients must match"""
    h = H()
    B, Cin, Cout, IH, IW, k, s = case
    g = torch.Generator().manual_seed(sum(case))
    x = bf(torch.randn(B, Cin, IH, IW, generator=g)).requires_grad_(True)
    w = torch.zeros(Cout, Cin, k, k, requires_grad=True)
    b = torch.zeros(Cout, requires_grad=True)
    o = F.conv2d(x, w, b, stride=s, padding=1 if k == 3 else 0)
    gy = bf(torch.randn(o.shape, generator=g))
    o.backward(gy)
    st = h.stream_ptr()
    nbytes = h.query_size("yogo_conv2d_wgrad_bf16_workspace_bytes", B, Cin, Cout, IH, IW, k, s)
    assert nbytes % 4 == 0
    guard = 1 << 18   # floats
    buf = torch.full((nbytes // 4 + guard,), float("nan"), device="cuda")
    dw = torch.full((Cout, Cin, k, k), float("nan"), device="cuda")
    db = torch.full((Cout,), float("nan"), device="cuda")
    h.call("yogo_conv2d_wgrad_bf16", to8c(x.detach()), to8c(gy), dw, db, buf, B, Cin, Cout, IH, IW, k, s, 0.0, st)
    torch.cuda.synchronize()
    assert bool(torch.isnan(buf[nbytes // 4:]).all()), (case, "the launch wrote behind its workspace")
    assert float((dw.cpu() - w.grad).abs().max()) < 1e-4 * float(w.grad.abs().max()), case
    assert float((db.cpu() - b.grad).abs().max()) < 1e-4 * float(b.grad.abs().max()), case


def test_deferred_wgrad_reductions_through_the_c_abi():
    """yogo_conv2d_wgrad_bf16_deferred + yogo_wgrad_reduce_flush against yogo_conv2d_wgrad_bf16, 20 reductions of six shapes in one
    queue (a queue holds 16: the 17th flushes the first sixteen), with and without bias gradient and clamp: the same bits; a reset
    queue runs nothing."""
    import ctypes
    h = H()
    st = h.stream_ptr()
    qh = ctypes.c_void_p(0)
    h.call("yogo_wgrad_reduce_queue_create", ctypes.addressof(qh))
    q = int(qh.value)
    try:
        shapes = [(1, 128, 32, 9, 12, 3, 1), (2, 128, 24, 11, 20, 3, 1), (1, 128, 128, 9, 17, 3, 1), (2, 16, 32, 9, 12, 3, 1),
                  (1, 128, 12, 7, 11, 1, 1), (2, 32, 64, 18, 22, 3, 2)]
        g = torch.Generator().manual_seed(77)
        jobs = []
        for j in range(20):
            B, Cin, Cout, IH, IW, k, s = shapes[j % len(shapes)]
            pad = 1 if k == 3 else 0
            OH, OW = (IH + 2 * pad - k) // s + 1, (IW + 2 * pad - k) // s + 1
            x8 = to8c(bf(torch.randn(B, Cin, IH, IW, generator=g)))
            g8 = to8c(bf(torch.randn(B, Cout, OH, OW, generator=g)))
            nbytes = h.query_size("yogo_conv2d_wgrad_bf16_workspace_bytes", B, Cin, Cout, IH, IW, k, s)
            clip = 0.0 if j % 3 else 0.05
            want_db = j % 2 == 0
            ref = (torch.full((Cout, Cin, k, k), float("nan"), device="cuda"), torch.full((Cout,), float("nan"), device="cuda") if want_db else None)
            got = (torch.full_like(ref[0], float("nan")), torch.full_like(ref[1], float("nan")) if want_db else None)
            ws0 = torch.empty(nbytes // 4, device="cuda")
            ws1 = torch.empty(nbytes // 4, device="cuda")
            h.call("yogo_conv2d_wgrad_bf16", x8, g8, ref[0], ref[1], ws0, B, Cin, Cout, IH, IW, k, s, clip, st)
            h.call("yogo_conv2d_wgrad_bf16_deferred", x8, g8, got[0], got[1], ws1, B, Cin, Cout, IH, IW, k, s, clip, q, st)
            jobs.append((ref, got, ws1, x8, g8))
        torch.cuda.synchronize()
        assert not bool(torch.isnan(jobs[0][1][0]).any())      # the first sixteen ran when the seventeenth was recorded
        assert bool(torch.isnan(jobs[19][1][0]).all())         # the last four wait for the flush
        h.call("yogo_wgrad_reduce_flush", q, st)
        torch.cuda.synchronize()
        for ref, got, *_ in jobs:
            assert torch.equal(ref[0], got[0]) and (ref[1] is None or torch.equal(ref[1], got[1]))
        # recorded, then forgotten: nothing is written
        B, Cin, Cout, IH, IW, k, s = shapes[0]
        dwx = torch.full((Cout, Cin, k, k), float("nan"), device="cuda")
        h.call("yogo_conv2d_wgrad_bf16_deferred", jobs[0][3], jobs[0][4], dwx, None, jobs[0][2], B, Cin, Cout, IH, IW, k, s, 0.0, q, st)
        h.call("yogo_wgrad_reduce_queue_reset", q)
        h.call("yogo_wgrad_reduce_flush", q, st)
        torch.cuda.synchronize()
        assert bool(torch.isnan(dwx).all())
    finally:
        h.call("yogo_wgrad_reduce_queue_destroy", q)


@pytest.mark.parametrize("B,IH,IW,Cout,use_bias", [(3, 20, 24, 16, False), (2, 36, 70, 7, True), (2, 772, 1032, 16, False)])
def test_layer0_on_matrix_cores(B, IH, IW, Cout, use_bias):
    """yogo_conv_first_mfma: uint8 image -> conv (bf16-rounded weights, exact bf16 inputs, fp32 accumulation) + BatchNorm sums,
    and z / y = act(BN(z)) of the second sweep."""
    h = H()
    assert h.lib().yogo_conv_first_mfma_supported(0, 1, Cout, IH, IW, 2) == 1
    assert h.lib().yogo_conv_first_mfma_supported(0, 1, Cout, IH + 1, IW, 2) == 0      # odd sizes, other strides, RGB: old kernels
    assert h.lib().yogo_conv_first_mfma_supported(0, 3, Cout, IH, IW, 2) == 0
    g = torch.Generator().manual_seed(IH)
    x = torch.randint(0, 256, (B, 1, IH, IW), generator=g, dtype=torch.uint8)
    w = torch.randn(Cout, 1, 3, 3, generator=g) * 0.02
    b = torch.randn(Cout, generator=g) if use_bias else None
    ref = F.conv2d(x.float(), bf(w), b, stride=2, padding=1)
    OH, OW = ref.shape[2:]
    st = h.stream_ptr()
    rows = h.query_ints("yogo_conv_first_mfma_stats_rows", 1, B, IH, IW)[0]
    stats = torch.full((rows, 16, 2), float("nan"), device="cuda")
    xc, wc, bc = x.cuda(), w.cuda(), (b.cuda() if use_bias else None)
    h.call("yogo_conv_first_mfma", xc, wc, bc, None, None, None, None, None, None, stats, B, Cout, IH, IW, 1, st)
    ssum = stats.cpu().double().sum(0)
    torch.testing.assert_close(ssum[:Cout, 0], ref.double().sum((0, 2, 3)), rtol=1e-5, atol=1e-2)
    torch.testing.assert_close(ssum[:Cout, 1], (ref.double() ** 2).sum((0, 2, 3)), rtol=1e-5, atol=1e-2)
    assert float(ssum[Cout:].abs().max()) == 0.0 if Cout < 16 else True
    mean = ref.mean((0, 2, 3))
    invstd = 1.0 / torch.sqrt(ref.var((0, 2, 3), unbiased=False) + 1e-5)
    gamma, beta = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g)
    z = torch.full((B, 2, OH, OW, 8), float("nan"), dtype=torch.bfloat16, device="cuda")
    y = torch.full_like(z, float("nan"))
    h.call("yogo_conv_first_mfma", xc, wc, bc, z, y, mean.cuda(), invstd.cuda(), gamma.cuda(), beta.cuda(), None, B, Cout, IH, IW, 1, st)
    assert rel_err(from8c(z, Cout), ref) < 4e-3          # one bf16 rounding of the output
    padc = z.cpu().float().permute(0, 1, 4, 2, 3).reshape(B, 16, OH, OW)[:, Cout:]
    assert padc.numel() == 0 or float(padc.abs().max()) == 0.0
    # y is BatchNorm + LeakyReLU of the stored (rounded) z: bit-identical to the stand-alone kernel on that tensor
    y2 = torch.full_like(z, float("nan"))
    h.call("yogo_bn_apply_act_bf16", z, y2, mean.cuda(), invstd.cuda(), 0, 1e-5, gamma.cuda(), beta.cuda(), B, Cout, OH * OW, 1, st)
    assert torch.equal(y.view(torch.int16), y2.view(torch.int16))
    # the variant with the sign map of the BatchNorm output (ABI 5): same y with or without z; a pixel's two bytes hold, in the kernel's
    # lane order, the bits (z * sc + sh > 0) of the ROUNDED z -- the value y is made of -- and padding channels read "not positive"
    y3 = torch.full_like(z, float("nan"))
    sg = torch.full((B, OH * OW * 2), 0xAA, dtype=torch.uint8, device="cuda")
    h.call("yogo_conv_first_mfma_signs", xc, wc, bc, None, y3, sg, mean.cuda(), invstd.cuda(), gamma.cuda(), beta.cuda(), B, Cout, IH, IW, 1, st)
    assert torch.equal(y.view(torch.int16), y3.view(torch.int16))
    want = O.l0_sign_map(dict(z=from8c(z, Cout).float().cpu(), mean=mean, invstd=invstd, gamma=gamma, beta=beta))
    nflip = int((sg.cpu() != want).sum())
    assert nflip <= 1e-5 * want.numel() + 1, (nflip, want.numel())   # (an fma against a multiply and an add, next to zero)
    # the two-pixels-per-lane form of the sweep (even output widths; the default) against the one-pixel form: the same bits
    from _util import hooks_library

    with hooks_library():   # (the product library has no plan switch: the one-pixel form runs in the test-hooks build of the same objects)
        h.call("yogo_hook_conv_first_mfma_pairs", 0)
        z1, y1 = torch.full_like(z, float("nan")), torch.full_like(z, float("nan"))
        sg1 = torch.full_like(sg, 0x55)
        h.call("yogo_conv_first_mfma_signs", xc, wc, bc, z1, y1, sg1, mean.cuda(), invstd.cuda(), gamma.cuda(), beta.cuda(), B, Cout, IH, IW, 1, st)
        torch.cuda.synchronize()
    assert torch.equal(z1.view(torch.int16), z.view(torch.int16)) and torch.equal(y1.view(torch.int16), y.view(torch.int16))
    assert torch.equal(sg1, sg)
    # and the direct kernel with the same (rounded) weights agrees to the output rounding
    z_old = torch.full_like(z, float("nan"))
    h.call("yogo_conv_first_fwd_train_bf16", xc, 0, bf(w).cuda(), bc, z_old, None, None, B, 1, Cout, IH, IW, 2, 0, st)
    assert rel_err(z.float().cpu(), z_old.float().cpu()) < 4e-3


@pytest.mark.parametrize("B,IH,IW", [(3, 20, 24), (2, 18, 28), (2, 772, 1032)])   # (output width 14: the one-pixel-per-step kernel)
def test_layer0_statistics_from_gram(B, IH, IW):
    """yogo_conv_first_gram + yogo_bn_stats_from_gram: patch sums are exact integers; mean / invstd / running statistics equal
    those of the convolution output (float64 reference)."""
    h = H()
    g = torch.Generator().manual_seed(IW)
    x = torch.randint(0, 256, (B, 1, IH, IW), generator=g, dtype=torch.uint8)
    x[0, 0, : IH // 2] = 200                        # a flat region: the statistics must survive the cancellation
    Cout = 16
    w = torch.randn(Cout, 1, 3, 3, generator=g) * 0.1
    w[3] -= w[3].mean()                              # a zero-mean filter
    b = torch.randn(Cout, generator=g)
    st = h.stream_ptr()
    rows = h.query_ints("yogo_conv_first_gram_rows", 1, B, IH, IW)[0]
    part = torch.empty(rows * 54, dtype=torch.int32, device="cuda")
    gram, gram32 = torch.empty(90, dtype=torch.float64, device="cuda"), torch.empty(90, device="cuda")
    xc = x.cuda()
    h.call("yogo_conv_first_gram", xc, part, gram, gram32, B, IH, IW, st)
    patches = F.unfold(x.double(), 3, padding=1, stride=2)            # [B, 9, L]
    P = patches.sum((0, 2))
    G = torch.einsum("bjl,bkl->jk", patches, patches)
    assert torch.equal(gram.cpu(), torch.cat((P, G.reshape(-1))))    # exact
    assert torch.equal(gram32.cpu(), torch.cat((P, G.reshape(-1))).float())
    OH, OW = IH // 2, IW // 2
    mean, invstd = torch.empty(Cout, device="cuda"), torch.empty(Cout, device="cuda")
    rm, rv = torch.zeros(Cout, device="cuda"), torch.ones(Cout, device="cuda")
    nbt = torch.zeros(1, dtype=torch.int64, device="cuda")
    h.call("yogo_bn_stats_from_gram", gram, w.cuda(), b.cuda(), Cout, B * OH * OW, 1e-5, 0.1, mean, invstd, rm, rv, nbt, st)
    z = F.conv2d(x.double(), bf(w).double(), b.double(), stride=2, padding=1)
    m_ref, v_ref = z.mean((0, 2, 3)), z.var((0, 2, 3), unbiased=False)
    n = B * OH * OW
    torch.testing.assert_close(mean.cpu().double(), m_ref, rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(invstd.cpu().double(), 1.0 / torch.sqrt(v_ref + 1e-5), rtol=1e-6, atol=0)
    torch.testing.assert_close(rm.cpu().double(), 0.1 * m_ref, rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(rv.cpu().double(), 0.9 + 0.1 * v_ref * n / (n - 1), rtol=1e-6, atol=0)
    assert int(nbt.item()) == 1


@pytest.mark.parametrize("inference", [0, 1])
def test_decode_backward_straight_to_bf16(inference):
    """yogo_decode_bwd_bf16 == yogo_decode_bwd followed by the fp32 -> bf16 NCHW8c conversion, bit for bit"""
    h = H()
    B, P, Sy, Sx = 3, 12, 13, 17
    g = torch.Generator().manual_seed(9 + inference)
    raw = torch.randn(B, P, Sy, Sx, generator=g).cuda()
    raw[0, 2, 0, 0] = 90.0                         # beyond the exp clamp: zero gradient
    gout = torch.randn(B, P, Sy, Sx, generator=g).cuda()
    st = h.stream_ptr()
    cx, cy = torch.linspace(0, 1 - 1 / Sx, Sx).cuda(), torch.linspace(0, 1 - 1 / Sy, Sy).cuda()
    out = torch.empty_like(raw)
    h.call("yogo_decode_fwd", raw, out, cx, cy, B, P, Sy, Sx, 0.05, 0.06, 1.0, 1.0, inference, st)
    g32 = torch.empty_like(raw)
    h.call("yogo_decode_bwd", raw, out, gout, g32, B, P, Sy, Sx, inference, st)
    want = torch.empty(B, 2, Sy, Sx, 8, dtype=torch.bfloat16, device="cuda")
    h.call("yogo_nchw_f32_to_bf16_8c", g32, want, B, P, Sy * Sx, st)
    got = torch.full_like(want, float("nan"))
    h.call("yogo_decode_bwd_bf16", raw, out, gout, got, B, P, Sy, Sx, inference, st)
    assert torch.equal(got.view(torch.int16), want.view(torch.int16))


def test_batchnorm_bf16():
    h = H()
    B, C, Hh, W = 3, 20, 11, 13
    g = torch.Generator().manual_seed(4)
    z = bf(torch.randn(B, C, Hh, W, generator=g) * 3 + 2).requires_grad_(True)
    gamma = (1 + 0.1 * torch.randn(C, generator=g)).requires_grad_(True)
    beta = (0.1 * torch.randn(C, generator=g)).requires_grad_(True)
    y_ref = F.leaky_relu(F.batch_norm(z, None, None, gamma, beta, training=True, eps=1e-5), 0.01)
    gy = bf(torch.randn(z.shape, generator=g))
    y_ref.backward(gy)
    st = h.stream_ptr()
    zz = z.detach()
    mean = zz.mean((0, 2, 3)).cuda()
    invstd = (1 / torch.sqrt(zz.var((0, 2, 3), unbiased=False) + 1e-5)).cuda()
    z8 = to8c(zz)
    y8 = torch.empty_like(z8)
    h.call("yogo_bn_apply_act_bf16", z8, y8, mean, invstd, 0, 1e-5, gamma.detach().cuda(), beta.detach().cuda(), B, C, Hh * W, 1, st)
    assert rel_err(from8c(y8, C), y_ref.detach()) < 8e-3
    rows = h.query_ints("yogo_bn_bwd_bf16_rows", 1, B, Hh * W)[0]
    part = torch.empty(rows * C * 2, device="cuda")
    sums = torch.empty(2 * C, device="cuda")
    dgamma, dbeta = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    g8 = to8c(gy)
    dz8 = torch.empty_like(g8)
    h.call("yogo_bn_bwd_bf16", g8, z8, dz8, mean, invstd, gamma.detach().cuda(), beta.detach().cuda(), 1, dgamma, dbeta, part, sums, B,
           C, Hh * W, 1, 0.0, st)
    assert rel_err(from8c(dz8, C), z.grad) < 1e-2
    assert rel_err(dgamma.cpu(), gamma.grad) < 1e-3 and rel_err(dbeta.cpu(), beta.grad) < 1e-3


def test_bf16_training_step_tracks_fp32():
    """one optimisation step with bf16 activations vs the same step in fp32 (HipTrainer half=True / False)"""
    from yogo_amd.model import YOGO
    from yogo_amd.train import HipTrainer
    from yogo_amd.yogo_loss import YOGOLoss

    Himg, Wimg, C, B = 96, 128, 7, 4
    x = O.synthetic_images(B, Himg, Wimg, seed=31).cuda()
    res = {}
    for half in (False, True):
        torch.manual_seed(3)
        model = YOGO((Himg, Wimg), 0.0425, 0.0555, C, clip_value=1e9).cuda()   # unclamped: compare the raw gradients
        model.train()
        for m in model.modules():
            if isinstance(m, torch.nn.Dropout2d):
                m.p = 0.0
        lab = O.synthetic_labels(B, model.Sx, model.Sy, K=6, num_classes=C, seed=32).cuda()
        tr = HipTrainer(model, YOGOLoss().cuda(), total_steps=5, half=half)
        tr.step(x, lab)
        res[half] = (tr.loss_components(), tr.flat.grad.clone().cpu(), [n for n, _ in model.named_parameters()],
                     [p.numel() for p in model.parameters()], {k: v.cpu().clone() for k, v in model.state_dict().items() if "running" in k})
    l32, g32, names, sizes, rs32 = res[False]
    l16, g16, _, _, rs16 = res[True]
    assert abs(l16["loss"] - l32["loss"]) < 2e-2 * abs(l32["loss"]), (l16, l32)
    off = 0
    for n, sz in zip(names, sizes):
        a, b_ = g16[off:off + sz], g32[off:off + sz]
        off += sz
        if float(b_.abs().max()) < 1e-6:
            continue   # e.g. a conv bias in front of BatchNorm: mathematically zero
        cos = float((a * b_).sum() / (a.norm() * b_.norm() + 1e-30))
        ratio = float(a.norm() / (b_.norm() + 1e-30))
        print(f"{n:24s} cos {cos:.4f} norm ratio {ratio:.4f}")
        # bf16 activations and activation gradients (8 significand bits) through 8 layers: direction and scale are kept,
        # element-wise agreement is not expected
        assert cos > 0.95 and 0.9 < ratio < 1.1, (n, cos, ratio)
    for k in rs32:
        torch.testing.assert_close(rs16[k], rs32[k], rtol=2e-2, atol=2e-2)


def test_fused_layer0_backward_matches_separate_passes():
    """BatchNorm backward + LeakyReLU derivative + first-conv weight gradient in one sweep (conv_first_bn_wgrad_kernel)
    against the separate bn_bwd -> bf16 dz -> conv_first_wgrad passes: same gradients up to the bf16 rounding of dz that the
    fused path no longer performs (stated: 1e-2 of the tensor's max)."""
    from yogo_amd import engine as E
    from yogo_amd.model import YOGO
    from yogo_amd.train import HipTrainer
    from yogo_amd.yogo_loss import YOGOLoss

    Himg, Wimg, C, B = 96, 128, 7, 4
    x = O.synthetic_images(B, Himg, Wimg, seed=41).cuda()
    out = {}
    old, old01 = E._FUSE_LAYER0_BWD, E._L01_FUSE_BWD
    try:
        E._L01_FUSE_BWD = False   # (the sweep that also takes layer 1's gradients needs the fused layer-0 backward: tests/test_gpu_first_fused_bwd.py)
        for fused in (False, True):
            E._FUSE_LAYER0_BWD = fused
            torch.manual_seed(5)
            model = YOGO((Himg, Wimg), 0.0425, 0.0555, C, clip_value=1e9).cuda()
            model.train()
            for m in model.modules():
                if isinstance(m, torch.nn.Dropout2d):
                    m.p = 0.0
            lab = O.synthetic_labels(B, model.Sx, model.Sy, K=6, num_classes=C, seed=42).cuda()
            tr = HipTrainer(model, YOGOLoss().cuda(), total_steps=5, half=True)
            tr.step(x, lab)
            names = [n for n, _ in model.named_parameters()]
            sizes = [p.numel() for p in model.parameters()]
            out[fused] = (tr.flat.grad.clone().cpu(), names, sizes)
    finally:
        E._FUSE_LAYER0_BWD, E._L01_FUSE_BWD = old, old01
    g0, names, sizes = out[False]
    g1 = out[True][0]
    off = 0
    for n, sz in zip(names, sizes):
        a, b_ = g1[off:off + sz], g0[off:off + sz]
        off += sz
        if n.startswith("model.0."):   # conv weight, BatchNorm weight / bias of layer 0
            assert float((a - b_).abs().max()) < 1e-2 * float(b_.abs().max()) + 1e-7, (n, float((a - b_).abs().max()), float(b_.abs().max()))
        else:                          # every other gradient comes from identical kernels on identical inputs
            assert torch.equal(a, b_), n


def test_layer0_backward_without_conv_output_matches_with_it():
    """engine._L0_NO_Z: the forward pass keeps the SIGN MAP of layer 0's BatchNorm output instead of its conv output z (1/16 of the
    bytes, yogo_conv_first_mfma_signs) and the fused backward sweep takes the LeakyReLU derivative from it and derives sum g * xhat
    from its own weight-gradient sums (yogo_conv_first_bn_wgrad_bf16_xs / _finalize_xs).  Against the sweep that reads z: y and every
    gradient behind layer 0 are bit-identical; layer 0's own gradients differ by what z's bf16 rounding (2^-9 of |z|, and |z| is a few
    sigma) puts into sum g * xhat on the path that reads z -- measured 1e-3 .. 4e-3 (conv weight) and 3e-3 .. 1e-2
    (BatchNorm weight; the larger on the 65 x 35 map) of the tensor's max; stated: 3e-2.  (The path without z is the one the
    teacher-forced checks pin against the oracle, at 5e-4: tests/_util.py, oracle.bf16_block_backward(l0_no_z=True).)"""
    from yogo_amd import engine as E
    from yogo_amd.model import YOGO
    from yogo_amd.train import HipTrainer
    from yogo_amd.yogo_loss import YOGOLoss

    for Himg, Wimg, B in ((96, 128, 4), (130, 70, 3)):
        x = O.synthetic_images(B, Himg, Wimg, seed=43).cuda()
        out = {}
        old = E._L0_NO_Z
        try:
            for no_z in (False, True):
                E._L0_NO_Z = no_z
                torch.manual_seed(6)
                model = YOGO((Himg, Wimg), 0.0425, 0.0555, 7, clip_value=1e9).cuda()
                model.train()
                lab = O.synthetic_labels(B, model.Sx, model.Sy, K=6, num_classes=7, seed=44).cuda()
                tr = HipTrainer(model, YOGOLoss().cuda(), total_steps=5, half=True)
                tr.trace = {}
                tr.step(x, lab)
                S0 = tr.trace["saved"][0]
                assert (S0.z is None) == no_z and (S0.signs0 is not None) == no_z
                names = [n for n, _ in model.named_parameters()]
                sizes = [p.numel() for p in model.parameters()]
                out[no_z] = (tr.flat.grad.clone().cpu(), S0.y.clone().cpu(), names, sizes)
        finally:
            E._L0_NO_Z = old
        g0, y0, names, sizes = out[False]
        g1, y1 = out[True][:2]
        assert torch.equal(y0.view(torch.int16), y1.view(torch.int16))
        off = 0
        for n, sz in zip(names, sizes):
            a, b_ = g1[off:off + sz], g0[off:off + sz]
            off += sz
            if n.startswith("model.0."):
                d = float((a - b_).abs().max())
                print(f"   {n:20s} max|d|/max|g| {d / float(b_.abs().max()):.2e}")
                assert d < 3e-2 * float(b_.abs().max()) + 1e-7, (n, d, float(b_.abs().max()))
            else:
                assert torch.equal(a, b_), n


def test_deferred_weight_gradient_reductions_are_bit_identical():
    """engine._WGRAD_DEFER_REDUCE: the split-K reductions of all weight gradients in one launch behind the last layer
    (yogo_conv2d_wgrad_bf16_deferred + yogo_wgrad_reduce_flush) against one launch per layer: the same bits in every gradient."""
    from yogo_amd import engine as E
    from yogo_amd.model import YOGO
    from yogo_amd.train import HipTrainer
    from yogo_amd.yogo_loss import YOGOLoss

    for Himg, Wimg, B in ((96, 128, 4), (193, 258, 3)):
        x = O.synthetic_images(B, Himg, Wimg, seed=45).cuda()
        out = {}
        old = E._WGRAD_DEFER_REDUCE
        try:
            for defer in (False, True):
                E._WGRAD_DEFER_REDUCE = defer
                torch.manual_seed(7)
                model = YOGO((Himg, Wimg), 0.0425, 0.0555, 7, clip_value=1.0).cuda()
                model.train()
                lab = O.synthetic_labels(B, model.Sx, model.Sy, K=6, num_classes=7, seed=46).cuda()
                tr = HipTrainer(model, YOGOLoss().cuda(), total_steps=5, half=True)
                tr.step(x, lab)
                out[defer] = tr.flat.grad.clone().cpu()
        finally:
            E._WGRAD_DEFER_REDUCE = old
        assert torch.isfinite(out[True]).all() and float(out[True].abs().max()) > 0
        assert torch.equal(out[False], out[True])


def test_bf16_dropout_masks():
    """Dropout2d in the bf16 training path: all layers' channel masks come from one rand call; every layer keeps its own p,
    the survivors are scaled by 1 / (1 - p), dropped channels are zero in the block output."""
    from yogo_amd import engine as E
    from yogo_amd.model import YOGO

    torch.manual_seed(3)
    m = YOGO((96, 128), 0.0425, 0.0555, 7).cuda()
    m.train()
    eng = E.get_engine(m.model)
    x = torch.randint(0, 256, (64, 1, 96, 128), dtype=torch.uint8).cuda()
    with torch.no_grad():
        raw, saved = E.forward_bf16_train(eng, x)
    assert torch.isfinite(raw).all()
    ps = [float(L.drop.p) for L in eng.layers if L.drop is not None]
    assert len(set(ps)) > 1                      # the reference uses different rates (model_defns.py:38-52)
    seen = 0
    for L, S in zip(eng.layers, saved):
        if L.drop is None:
            assert S.mask is None
            continue
        p = float(L.drop.p)
        mask = S.mask.cpu()
        assert mask.shape == (64, L.cout) and S.mask.is_contiguous()
        vals = mask.unique().tolist()
        assert all(v == 0.0 or abs(v - 1 / (1 - p)) < 1e-6 for v in vals) and len(vals) == 2
        frac = float((mask == 0).float().mean())
        assert abs(frac - p) < 4 * (p * (1 - p) / mask.numel()) ** 0.5 + 1e-3
        y = from8c(S.y, L.cout)
        assert torch.all(y[mask == 0] == 0)
        seen += 1
    assert seen == len(ps) == 3
    m.eval()
    with torch.no_grad():
        _, saved = E.forward_bf16_train(eng, x[:2])
    assert all(S.mask is None for S in saved)


# (name, hw, rgb, batch, end-to-end cosine bound).  The END-TO-END cosine of a gradient tensor is a property of the input as much
# as of the kernels (tests/_util.py: one stored value on a rounding boundary, one LeakyReLU' sign in front of a sparse gradient):
# the points marked None are the ones of tools/probes/sweep_models.py where ONE tensor sits at 0.968 ... 0.9947 with every kernel
# right (round 3: 7 of 33; round 4, whose persistent convolution kernel sums the small images' 16-channel chunks in another order:
# depth_ver_3 96x128 joined them at 0.9937).  They run here with that bound WAIVED (a floor of 0.95 stays) and the tight
# teacher-forced per-kernel check ON, so the suite shows the exclusion to be harmless instead of hiding it.
_ARCH_CASES = [("silu_model", (96, 128), False, 2, 0.995), ("quarter_filters", (130, 70), True, 2, 0.995),
               ("depth_ver_3", (96, 128), False, 2, None), ("triple_filters", (64, 96), False, 2, 0.995),
               ("triple_filters", (193, 258), False, 2, 0.995), ("double_filters", (96, 128), False, 2, 0.995),
               ("base_model", (97, 131), False, 2, 0.995),   # odd sizes: the direct layer-0 kernels
               ("base_model", (193, 258), False, 3, None), ("half_filters", (193, 258), False, 3, None),
               ("depth_ver_3", (130, 70), True, 1, None), ("depth_ver_4", (96, 128), False, 2, None),
               ("depth_ver_4", (193, 258), False, 3, None), ("depth_ver_4", (130, 70), True, 1, None),
               ("depth_ver_2", (193, 258), False, 3, None)]


@pytest.mark.parametrize("name,hw,rgb,B,cos_min", _ARCH_CASES)
def test_bf16_training_other_architectures(name, hw, rgb, B, cos_min):
    """two bf16 optimisation steps of other registered ModelDefns (SiLU blocks keep their pre-activation for the backward pass;
    widths 4..384; rgb input; the direct layer-0 kernels at odd sizes) against the oracle's bf16-storage emulation
    (O.bf16_train_step): step 1 -- end to end loss 1e-3 and every gradient tensor cosine >= 0.995, and TEACHER-FORCED every
    stored tensor / statistic / parameter gradient given the step's own inputs (one bf16 ulp, 2e-4 of max|g|: tests/_util.py);
    then the update: the fused AdamW on the step's own gradients against the oracle's AdamW, and the SECOND step's loss against
    the emulation started from the step's own updated parameters (1e-3)"""
    from _util import BF16_STEP_LOSS_RTOL, assert_grads_match_bf16_oracle, teacher_forced_bf16_step_check
    from yogo_amd.model import YOGO
    from yogo_amd.model_defns import MODELS
    from yogo_amd.train import HipTrainer
    from yogo_amd.yogo_loss import YOGOLoss

    H_, W_ = hw
    x = torch.randint(0, 256, (B, 3 if rgb else 1, H_, W_), dtype=torch.uint8, generator=torch.Generator().manual_seed(2))
    torch.manual_seed(1)
    m = YOGO((H_, W_), 0.0425, 0.0555, 5, is_rgb=rgb, model_func=MODELS[name], clip_value=1e9).cuda()
    m.train()
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout2d):
            mod.p = 0.0
    sd0 = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    lab = O.synthetic_labels(B, m.Sx, m.Sy, K=4, num_classes=5, seed=3)
    tr = HipTrainer(m, YOGOLoss().cuda(), total_steps=4, half=True)
    tr.trace = {}
    tr.step(x.cuda(), lab.cuda())
    torch.cuda.synchronize()
    spec = O.arch(name, 5)
    teacher_forced_bf16_step_check(O, tr, m, x, lab, spec, sd0, f"{name} {H_}x{W_}")   # (tight, per kernel: never waived)
    tr.trace = None
    loss_ref, _, grads_ref, _ = O.bf16_train_step(x, sd0, spec, lab, 0.0425, 0.0555)
    got = tr.loss_components()["loss"]
    assert abs(got - loss_ref) < BF16_STEP_LOSS_RTOL * abs(loss_ref), (name, got, loss_ref)
    mine, off = {}, 0
    for pname, p in m.named_parameters():
        mine[pname] = tr.flat.grad[off:off + p.numel()].view(p.shape).cpu()
        off += p.numel()
    assert_grads_match_bf16_oracle(mine, grads_ref, f"{name} {H_}x{W_}", cos_min=0.95 if cos_min is None else cos_min)
    # ---- the update and step 2, again with the step's own tensors (the END-TO-END second loss is ill-conditioned: the loss falls by
    # 50-80 % in this one step and Adam's first update is lr * sign(g), so gradient components that are rounding noise move their
    # weights either way -- measured 1e-4 ... 1e-1 between two correct implementations on the wide models):
    #   (i) the fused AdamW kernel on the step's OWN gradients == the oracle's AdamW on those gradients;
    #   (ii) the second step's loss == the emulation's loss from the step's OWN updated parameters (repacked bf16 weights, forward,
    #        decode, loss), 1e-3.
    sd1 = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    for k, g in mine.items():
        want_p, _, _ = O.adamw_step(sd0[k], g, torch.zeros_like(g), torch.zeros_like(g), 1, tr.lr, weight_decay=tr.wd)
        assert float((sd1[k] - want_p).abs().max()) < 2e-6 + 1e-5 * tr.lr, (name, k)
    loss2_ref, _, _, _ = O.bf16_train_step(x, sd1, spec, lab, 0.0425, 0.0555)
    tr.step(x.cuda(), lab.cuda())
    got2 = tr.loss_components()["loss"]
    print(f"{name} {H_}x{W_}: loss {got:.5f} -> {got2:.5f}; oracle {loss_ref:.5f} -> {loss2_ref:.5f} (from the step's own parameters)")
    assert abs(got2 - loss2_ref) < BF16_STEP_LOSS_RTOL * abs(loss2_ref), (name, got2, loss2_ref)
