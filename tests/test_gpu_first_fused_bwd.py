"""Layer 1's data gradient folded into layer 0's backward sums (yogo_conv2d_dgrad_bf16_first_bwd, conv_first_fused_bwd.hip) against
(a) a CPU fp64 restatement of what the unfused pair computes and (b) the unfused pair itself (yogo_conv2d_dgrad_bf16 +
yogo_conv_first_bn_wgrad_bf16_xs), and the training step with and without it.  Reference: autograd of yogo/model_defns.py:34-41."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

NJ, PER = 9, 20


def _to8c(t):
    """fp32 NCHW [B][C][H][W] (C a multiple of 8) -> bf16 NCHW8c [B][C/8][H][W][8]"""
    B, C, H, W = t.shape
    return t.reshape(B, C // 8, 8, H, W).permute(0, 1, 3, 4, 2).contiguous().to(torch.bfloat16)


def _sign_bits(signs_u8, H, W):
    """[B][H*W][2] bytes -> bool [B][16][H][W]: byte h of a pixel: bit i = channel 4h + i, bit 4 + i = channel 8 + 4h + i (include/yogo_hip.h)"""
    s = signs_u8.reshape(-1, H, W, 2).to(torch.int32)
    out = torch.zeros(s.shape[0], 16, H, W, dtype=torch.bool)
    for h in range(2):
        for i in range(4):
            out[:, 4 * h + i] = ((s[..., h] >> i) & 1).bool()
            out[:, 8 + 4 * h + i] = ((s[..., h] >> (4 + i)) & 1).bool()
    return out


def _cpu_sums(g, w, img, signs_u8, leaky):
    """fp64: dy = conv_transpose(g, bf16(w)) rounded to bf16; gb = dy * (sign ? 1 : 0.01); A1[c][j] = sum gb[c] patch_j, S1[c] = sum gb[c]"""
    B, _, H, W = g.shape
    wb = w.to(torch.bfloat16).double()
    dy = F.conv_transpose2d(g.double(), wb, padding=1)
    dyb = dy.float().to(torch.bfloat16).double()
    if leaky:
        pos = _sign_bits(signs_u8, H, W)
        gb = dyb * torch.where(pos, torch.tensor(1.0, dtype=torch.float64), torch.tensor(float(np.float32(0.01)), dtype=torch.float64))
    else:
        gb = dyb
    patches = F.unfold(img.double(), 3, padding=1, stride=2).reshape(B, 9, H * W)
    A1 = torch.einsum("bcp,bjp->cj", gb.reshape(B, 16, H * W), patches)
    S1 = gb.sum((0, 2, 3))
    return A1, S1, dy


def _run_pair(h, g8, pk, img, signs, B, H, W, act0, fused, x8=None, clip=0.0):
    """fused: 1 = the data-gradient sweep, 2 = with layer 1's weight gradient (returns dw, db as well)"""
    st = h.stream_ptr()
    cols = h.query_ints("yogo_conv_first_bn_wgrad_cols", 1, 1, 16)[0]
    extra = ()
    if fused == 2:
        rows = h.query_ints("yogo_conv2d_dgrad_first_bwd_rows", 1, B, H, W, 1)[0]
        part = torch.full((rows * cols,), float("nan"), dtype=torch.float32, device="cuda")
        ws = torch.full((h.query_size("yogo_conv2d_dgrad_wgrad_first_bwd_workspace_bytes", B, H, W) // 4,), float("nan"), dtype=torch.float32, device="cuda")
        dw = torch.full((32, 16, 3, 3), float("nan"), dtype=torch.float32, device="cuda")
        db = torch.full((32,), float("nan"), dtype=torch.float32, device="cuda")
        h.call("yogo_conv2d_dgrad_wgrad_bf16_first_bwd", g8, pk, x8, img, signs if act0 else None, part, dw, db, ws, B, 16, 32, H, W, act0, clip, None, st)
        extra = (dw, db)
    elif fused:
        rows = h.query_ints("yogo_conv2d_dgrad_first_bwd_rows", 1, B, H, W, 0)[0]
        part = torch.full((rows * cols,), float("nan"), dtype=torch.float32, device="cuda")
        h.call("yogo_conv2d_dgrad_bf16_first_bwd", g8, pk, img, signs if act0 else None, part, B, 16, 32, H, W, act0, st)
    else:
        dx = torch.full((B, 2, H, W, 8), float("nan"), dtype=torch.bfloat16, device="cuda")
        h.call("yogo_conv2d_dgrad_bf16", g8, pk, dx, None, 0, None, B, 16, 32, H, W, 3, 1, st)
        rows = h.query_ints("yogo_conv_first_wgrad_rows", 1, B, 2 * H, 2 * W, 2)[0]
        part = torch.full((rows * cols,), float("nan"), dtype=torch.float32, device="cuda")
        one = torch.ones(16, device="cuda")
        h.call("yogo_conv_first_bn_wgrad_bf16_xs", img, 0, dx, signs if act0 else None, one, one, one, one, part, B, 1, 16, 2 * H, 2 * W, 2, act0, st)
    sums = torch.empty(cols, dtype=torch.float32, device="cuda")
    h.call("yogo_partials_reduce", part, rows, cols, 0.0, sums, st)
    torch.cuda.synchronize()
    s = sums.cpu()[: 16 * PER].reshape(16, PER)
    return (s[:, :NJ].double(), s[:, 2 * NJ].double()) + tuple(t.cpu().double() for t in extra)


@pytest.mark.parametrize("B,H,W,act0", [(2, 37, 70, 1), (3, 64, 96, 1), (1, 9, 34, 1), (2, 21, 30, 0), (2, 4, 2, 1), (2, 386, 516, 1)])
def test_fused_sweep_against_cpu_and_the_unfused_pair(B, H, W, act0):
    from yogo_amd import _hip as h

    assert h.lib().yogo_conv2d_dgrad_first_bwd_supported(16, 32, H, W, B, act0) == 1
    gen = torch.Generator().manual_seed(1000 * H + W)
    g = (torch.randn(B, 32, H, W, generator=gen) * 0.5).to(torch.bfloat16).float()
    w = torch.randn(32, 16, 3, 3, generator=gen) * 0.1
    img = torch.randint(0, 256, (B, 1, 2 * H, 2 * W), generator=gen, dtype=torch.uint8)
    signs = torch.randint(0, 256, (B, H * W * 2), generator=gen, dtype=torch.uint8)
    st = h.stream_ptr()
    pk = torch.empty(h.query_size("yogo_conv_bf16_packed_bytes", 16, 32, 3, 1), dtype=torch.uint8, device="cuda")
    h.call("yogo_conv_bf16_pack", w.cuda(), None, pk, 16, 32, 3, 1, st)
    g8, imgc, sgc = _to8c(g).cuda(), img.cuda(), signs.cuda()
    A1f, S1f = _run_pair(h, g8, pk, imgc, sgc, B, H, W, act0, 1)
    A1u, S1u = _run_pair(h, g8, pk, imgc, sgc, B, H, W, act0, 0)
    # ... and with layer 1's weight gradient in the same sweep: the same sums (another tile height: another summation order), dw / db against
    # the weight-gradient kernel and fp64
    x = (torch.randn(B, 16, H, W, generator=gen)).to(torch.bfloat16).float()
    x8 = _to8c(x).cuda()
    A1w, S1w, dw, db = _run_pair(h, g8, pk, imgc, sgc, B, H, W, act0, 2, x8=x8)
    dwu = torch.full((32, 16, 3, 3), float("nan"), dtype=torch.float32, device="cuda")
    dbu = torch.full((32,), float("nan"), dtype=torch.float32, device="cuda")
    wsu = torch.empty(h.query_size("yogo_conv2d_wgrad_bf16_workspace_bytes", B, 16, 32, H, W, 3, 1) // 4, dtype=torch.float32, device="cuda")
    h.call("yogo_conv2d_wgrad_bf16", x8, g8, dwu, dbu, wsu, B, 16, 32, H, W, 3, 1, 0.0, st)
    torch.cuda.synchronize()
    dwu, dbu = dwu.cpu().double(), dbu.cpu().double()
    # products of bf16 values are exact, both kernels sum them in fp32 (in different orders) and reduce the partial sums in double
    assert float((dw - dwu).abs().max()) < 1e-4 * float(dwu.abs().max()) + 1e-5, (float((dw - dwu).abs().max()), float(dwu.abs().max()))
    assert float((db - dbu).abs().max()) < 1e-4 * float(dbu.abs().max()) + 1e-5, (float((db - dbu).abs().max()), float(dbu.abs().max()))
    if B * H * W <= 40000:
        dw64 = torch.nn.grad.conv2d_weight(x.double(), (32, 16, 3, 3), g.double(), padding=1)
        assert float((dw - dw64).abs().max()) < 1e-4 * float(dw64.abs().max()) + 1e-5
        assert float((db - g.double().sum((0, 2, 3))).abs().max()) < 1e-4 * float(g.double().sum((0, 2, 3)).abs().max()) + 1e-4
    assert float((A1w - A1u).abs().max()) < 4e-3 * float(A1u.abs().max()) + 1e-3
    assert float((S1w - S1u).abs().max()) < 4e-3 * float(S1u.abs().max()) + 1e-3
    # fused against unfused: the same elements, another summation order (and one bf16 ulp of dy where the 32-channel MFMA rounds otherwise)
    sa, ss = float(A1u.abs().max()), float(S1u.abs().max())
    assert float((A1f - A1u).abs().max()) < 4e-3 * sa + 1e-3, (float((A1f - A1u).abs().max()), sa)
    assert float((S1f - S1u).abs().max()) < 4e-3 * ss + 1e-3, (float((S1f - S1u).abs().max()), ss)
    if B * H * W <= 40000:   # the independent reference
        A1, S1, dy = _cpu_sums(g, w, img, signs, act0 == 1)
        # a term is |dy| * 255 at most and dy's bf16 rounding moves it by 2^-9: the bound scales with sum |gb| * patch, not with the (cancelling) sum
        scale = float(dy.abs().sum()) * 255.0 * 2.0 ** -9
        assert float((A1f - A1).abs().max()) < 0.05 * scale + 1e-3, (float((A1f - A1).abs().max()), scale)
        assert float((S1f - S1).abs().max()) < 0.05 * scale / 255.0 + 1e-3
        assert float((A1u - A1).abs().max()) < 0.05 * scale + 1e-3
        # and relative to the sums themselves where they do not cancel
        assert float((A1f - A1).abs().max()) < 5e-3 * float(A1.abs().max()) + 1e-3, (float((A1f - A1).abs().max()), float(A1.abs().max()))


def test_unsupported_shapes_are_refused():
    from yogo_amd import _hip as h

    L = h.lib()
    assert L.yogo_conv2d_dgrad_first_bwd_supported(16, 32, 64, 97, 2, 1) == 0   # odd width
    assert L.yogo_conv2d_dgrad_first_bwd_supported(8, 32, 64, 96, 2, 1) == 0
    assert L.yogo_conv2d_dgrad_first_bwd_supported(16, 64, 64, 96, 2, 1) == 0
    assert L.yogo_conv2d_dgrad_first_bwd_supported(16, 32, 64, 96, 2, 2) == 0   # SiLU
    part = torch.zeros(16, device="cuda")
    with pytest.raises(RuntimeError, match="unsupported shape"):
        h.call("yogo_conv2d_dgrad_bf16_first_bwd", part, part, part, part, part, 2, 16, 32, 64, 97, 1, h.stream_ptr())


def test_training_step_with_and_without_the_fused_sweep():
    """engine._L01_FUSE_BWD: every gradient above layer 1 is bit-identical (nothing they depend on changes), layer 0's gradients and layer
    1's weight / bias gradient agree to the rounding of their sums; the launch log shows which sweep ran."""
    from yogo_amd import _hip as h
    from yogo_amd import engine as E
    from yogo_amd.model import YOGO
    from yogo_amd.train import HipTrainer
    from yogo_amd.yogo_loss import YOGOLoss
    import yogo_oracle as O

    for Himg, Wimg, B in ((96, 128, 4), (132, 72, 3)):
        x = O.synthetic_images(B, Himg, Wimg, seed=43).cuda()
        out = {}
        old = E._L01_FUSE_BWD
        try:
            for fused in (False, True):
                E._L01_FUSE_BWD = fused
                torch.manual_seed(6)
                model = YOGO((Himg, Wimg), 0.0425, 0.0555, 7, clip_value=1e9).cuda()
                model.train()
                lab = O.synthetic_labels(B, model.Sx, model.Sy, K=6, num_classes=7, seed=44).cuda()
                tr = HipTrainer(model, YOGOLoss().cuda(), total_steps=5, half=True)
                h.launch_log(True)
                tr.step(x, lab)
                torch.cuda.synchronize()
                log = "\n".join(h.read_launch_log())
                h.launch_log(False)
                names = [n for n, _ in model.named_parameters()]
                sizes = [p.numel() for p in model.parameters()]
                out[fused] = (tr.flat.grad.clone().cpu(), names, sizes, log)
        finally:
            E._L01_FUSE_BWD = old
        g0, names, sizes, log0 = out[False]
        g1, log1 = out[True][0], out[True][3]
        assert "conv_bf16_dgrad_first_bwd_kernel" in log1 and "conv_bf16_dgrad_first_bwd_kernel" not in log0
        off = 0
        for n, sz in zip(names, sizes):
            a, b_ = g1[off:off + sz], g0[off:off + sz]
            off += sz
            if n.startswith("model.0.") or n.startswith("model.1.0."):
                # layer 0: sums of the same elements in another order; layer 1's weight / bias gradient: exact bf16 products summed in fp32 in
                # another order (the sweep's per-wavefront partial sums against wgrad_bf16_kernel's split-K slabs)
                d = float((a - b_).abs().max())
                print(f"   {n:20s} max|d|/max|g| {d / float(b_.abs().max()):.2e}")
                assert d < (5e-3 if n.startswith("model.0.") else 1e-4) * float(b_.abs().max()) + 1e-7, (n, d, float(b_.abs().max()))
            else:
                assert torch.equal(a, b_), n
