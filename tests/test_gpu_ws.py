"""The persistent wavefront-specialised convolution kernel (conv_bf16_ws_kernel) against the tiled 8-wavefront kernel it replaces
(conv_bf16_kernel<4,2,8,...>): same MFMA sequence per accumulator and the same epilogue formula, so every output bit and every
sign-map byte has to agree -- for the forward of yogo/model_defns.py:49-65's 128-channel blocks (bias + LeakyReLU + Dropout2d
channel mask + sign map, or bias only in front of BatchNorm) and for their data gradients (no bias, optional channel mask).
The tiled kernel itself is checked against CPU fp32 convolutions in test_gpu_bf16.py / test_gpu_production_shapes.py (which now
also run through the persistent kernel where it is eligible)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _blocks(c):
    return ((c + 15) // 16) * 2


def _run(persistent, kind, B, Cin, H, W, seed, ws16=True, product=False):
    """product: the PRODUCT library (libyogo_hip.so, which has no plan switch) with its own plan.  Otherwise the test-hooks library -- the
    same objects plus the yogo_hook_* switches (tests/_util.py:hooks_library): tiled (persistent = False), persistent without the 16x16x32
    member (ws16 = False: conv_bf16_ws_kernel<0> takes the plain-epilogue launches as it did until round 5), or persistent with the
    16x16x32 member on EVERY eligible launch (ws16 = True: the biased forward too, which the product leaves on conv_bf16_ws_kernel<0>)"""
    import contextlib

    from _util import hooks_library

    with (contextlib.nullcontext() if product else hooks_library()):
        return _run_in(persistent, kind, B, Cin, H, W, seed, ws16, product)


def _run_in(persistent, kind, B, Cin, H, W, seed, ws16=True, product=False):
    from yogo_amd import _hip as Hh

    Cout = 128
    st = Hh.stream_ptr()
    g = torch.Generator(device="cuda").manual_seed(seed)
    w = torch.randn(Cout, Cin, 3, 3, device="cuda", generator=g) * 0.05
    x8 = torch.randn(B, _blocks(Cin), H, W, 8, device="cuda", generator=g).to(torch.bfloat16)
    y8 = torch.full((B, _blocks(Cout), H, W, 8), 7.0, device="cuda").to(torch.bfloat16)   # poisoned: every unit must be written
    bias = torch.randn(Cout, device="cuda", generator=g)
    msk = (torch.rand(B, Cout, device="cuda", generator=g) > 0.2).float() / 0.8
    if product:
        pass
    elif not persistent:
        Hh.call("yogo_hook_conv_bf16_persistent", 0)
    else:
        # (ws16 = True: every eligible launch, the biased forward too -- the product sends only the bias-free data gradients there)
        Hh.call("yogo_hook_conv_bf16_ws16", 2 if ws16 else 0)
    try:
        Hh.launch_log(True)
        sg = None
        if kind == "fwd_signs":       # layer 3: bias + LeakyReLU + channel mask + sign map
            packed = torch.empty(Hh.query_size("yogo_conv_bf16_packed_bytes", Cin, Cout, 3, 0), dtype=torch.uint8, device="cuda")
            Hh.call("yogo_conv_bf16_pack", w, None, packed, Cin, Cout, 3, 0, st)
            sg = torch.full((Hh.query_size("yogo_bf16_signs_bytes", B, Cout, H, W),), 0x5A, dtype=torch.uint8, device="cuda")
            Hh.call("yogo_conv2d_fwd_bf16_signs", x8, packed, bias, y8, sg, msk, B, Cin, Cout, H, W, 3, 1, 1, st)
        elif kind == "fwd_plain":     # layers 5 / 6: conv + bias in front of BatchNorm
            packed = torch.empty(Hh.query_size("yogo_conv_bf16_packed_bytes", Cin, Cout, 3, 0), dtype=torch.uint8, device="cuda")
            Hh.call("yogo_conv_bf16_pack", w, None, packed, Cin, Cout, 3, 0, st)
            Hh.call("yogo_conv2d_fwd_bf16", x8, packed, bias, y8, None, None, None, B, Cin, Cout, H, W, 3, 1, 0, st)
        elif kind == "fwd_leaky":     # bias + LeakyReLU, no mask, no sign map (eval-mode forward)
            packed = torch.empty(Hh.query_size("yogo_conv_bf16_packed_bytes", Cin, Cout, 3, 0), dtype=torch.uint8, device="cuda")
            Hh.call("yogo_conv_bf16_pack", w, None, packed, Cin, Cout, 3, 0, st)
            Hh.call("yogo_conv2d_fwd_bf16", x8, packed, bias, y8, None, None, None, B, Cin, Cout, H, W, 3, 1, 1, st)
        else:                         # data gradient of a 128 -> Cin... here: dy has Cin channels, dx 128 (roles as the GEMM sees them)
            # yogo_conv2d_dgrad_bf16(dy [Cout_f], packed, dx [Cin_f]): forward conv Cin_f = 128 -> Cout_f = Cin
            wf = torch.randn(Cin, 128, 3, 3, device="cuda", generator=g) * 0.05
            packed = torch.empty(Hh.query_size("yogo_conv_bf16_packed_bytes", 128, Cin, 3, 1), dtype=torch.uint8, device="cuda")
            Hh.call("yogo_conv_bf16_pack", wf, None, packed, 128, Cin, 3, 1, st)
            Hh.call("yogo_conv2d_dgrad_bf16", x8, packed, y8, None, 0, msk if kind == "dgrad_mask" else None, B, 128, Cin, H, W, 3, 1, st)
        torch.cuda.synchronize()
        log = Hh.read_launch_log()
    finally:
        Hh.launch_log(False)
    return y8, sg, log


CASES = [
    # (kind, B, Cin, H, W)
    ("fwd_plain", 2, 128, 97, 129),     # layers 5 / 6 of base_model at 772x1032
    ("dgrad", 2, 128, 97, 129),
    ("fwd_signs", 2, 64, 193, 258),     # layer 3
    ("fwd_leaky", 3, 128, 13, 17),      # fewer tiles than CUs, one partial tile per image
    ("fwd_signs", 1, 64, 40, 300),      # several column bands
    ("dgrad_mask", 2, 96, 31, 45),      # 6 chunks
    ("fwd_plain", 1, 128, 5, 700),      # short and wide: bands of a few rows
    ("fwd_signs", 5, 128, 3, 3),        # one pixel group in use
    ("fwd_plain", 40, 128, 97, 129),    # 1 000 tiles: every workgroup walks several tiles, image changes at the seams
    ("fwd_signs", 24, 64, 193, 258),
    ("dgrad", 2, 256, 20, 22),          # 16 chunks per tile
    ("fwd_leaky", 1, 128, 300, 3),      # tall and three pixels wide: one band of width 3
    ("fwd_plain", 1, 64, 3, 1000),      # three rows: every tile holds the whole height
    ("dgrad_mask", 7, 128, 50, 131),    # widths one past a multiple of the band width
]


@pytest.mark.parametrize("kind,B,Cin,H,W", CASES)
def test_persistent_kernel_is_bit_identical_to_the_tiled_kernel(kind, B, Cin, H, W):
    y_old, s_old, log_old = _run(False, kind, B, Cin, H, W, seed=11)
    y_new, s_new, log_new = _run(True, kind, B, Cin, H, W, seed=11, ws16=False)
    assert any(ln.startswith("conv_bf16_kernel<4, 2, 8") for ln in log_old), log_old
    assert any(ln.startswith("conv_bf16_ws_kernel<") for ln in log_new), log_new
    plan = next(ln for ln in log_old if ln.startswith("conv_bf16_kernel<4, 2, 8"))
    if " CKb=2 " in plan:
        # the tiled kernel stepped through K in 16-channel chunks too: the same MFMA sequence per accumulator, bit for bit
        assert torch.equal(y_old.view(torch.int16), y_new.view(torch.int16)), (
            f"{(y_old.float() - y_new.float()).abs().max().item()} max abs difference, "
            f"{(y_old.view(torch.int16) != y_new.view(torch.int16)).float().mean().item()} of the values differ")
        if s_old is not None:
            assert torch.equal(s_old, s_new)
    else:
        # small images: the tiled kernel takes 32- or 64-channel chunks (tap-major inside a chunk), a different fp32 summation
        # order -- the two bf16 results may differ by one rounding step on a few values, the sign map where a value is ~0
        a, b = y_old.float(), y_new.float()
        ulp = 2.0 ** -7 * torch.maximum(a.abs(), b.abs()) + 1e-6 * a.abs().max()
        assert bool(((a - b).abs() <= ulp).all()), f"{((a - b).abs() - ulp).max().item()} beyond one bf16 step"
        assert (a != b).float().mean().item() < 5e-3
        if s_old is not None:
            assert (s_old != s_new).float().mean().item() < 5e-3


# ---- the 16x16x32 member of the family (conv_bf16_ws16_kernel: convolution [+ bias], K a multiple of 64) ------------------------------
WS16_CASES = [
    ("fwd_plain", 2, 128, 97, 129),     # layer 5's forward at 772x1032
    ("dgrad", 2, 128, 97, 129),         # the data gradients of layers 5 / 6
    ("fwd_plain", 3, 64, 13, 17),       # two chunk pairs = six periods, fewer tiles than CUs, a partial tile per image
    ("dgrad", 1, 64, 40, 300),          # several column bands
    ("fwd_plain", 1, 128, 5, 700),      # short and wide
    ("dgrad", 5, 128, 3, 3),            # 9 pixels: one pixel block partly in use
    ("fwd_plain", 40, 128, 97, 129),    # 1 000 tiles: every workgroup walks several tiles, the image changes at the seams
    ("dgrad", 2, 256, 20, 22),          # eight chunk pairs per tile
    ("fwd_plain", 1, 128, 300, 3),      # one band of width 3
    ("fwd_plain", 1, 64, 3, 1000),      # three rows: every tile holds the whole height
    ("dgrad", 7, 128, 50, 131),
    ("fwd_plain", 1, 192, 33, 35),      # six chunk pairs... (K = 192: Kb = 24)
]


@pytest.mark.parametrize("kind,B,Cin,H,W", WS16_CASES)
def test_ws16_against_the_32x32x16_kernel_and_cpu(kind, B, Cin, H, W):
    """conv_bf16_ws16_kernel sums the two 16-channel chunks of a pair inside one v_mfma_f32_16x16x32_bf16: another fp32 summation order
    than conv_bf16_ws_kernel<0> (bit-identical to the tiled kernel, above) -- the bf16 outputs may differ by one rounding step on a few
    values.  Plus an independent reference: torch's CPU float64 convolution of the same bf16-rounded operands
    (yogo/model_defns.py:54-65: conv + bias in front of BatchNorm; the data gradient = conv_transpose2d)."""
    import torch.nn.functional as F

    from yogo_amd import _hip as Hh

    y_old, _, log_old = _run(True, kind, B, Cin, H, W, seed=29, ws16=False)
    y_new, _, log_new = _run(True, kind, B, Cin, H, W, seed=29)
    assert any(ln.startswith("conv_bf16_ws_kernel<0>") for ln in log_old), log_old
    assert any(ln.startswith("conv_bf16_ws16_kernel<") for ln in log_new), log_new
    a, b = y_old.float(), y_new.float()
    ulp = 2.0 ** -7 * torch.maximum(a.abs(), b.abs()) + 1e-6 * a.abs().max()
    bad = (a - b).abs() > ulp
    assert not bool(bad.any()), f"{int(bad.sum())} values beyond one bf16 step, first at {bad.nonzero()[0].tolist()}, max excess {((a - b).abs() - ulp).max().item()}"
    assert (a != b).float().mean().item() < 5e-3
    # two launches give the same bits
    y_again, _, _ = _run(True, kind, B, Cin, H, W, seed=29)
    assert torch.equal(y_new.view(torch.int16), y_again.view(torch.int16))
    if B * H * W <= 40000:
        st = Hh.stream_ptr()
        g = torch.Generator(device="cuda").manual_seed(29)   # (the operands of _run_in, drawn again in the same order)
        w = torch.randn(128, Cin, 3, 3, device="cuda", generator=g) * 0.05
        x8 = torch.randn(B, _blocks(Cin), H, W, 8, device="cuda", generator=g).to(torch.bfloat16)
        bias = torch.randn(128, device="cuda", generator=g)
        _ = torch.rand(B, 128, device="cuda", generator=g)
        x = torch.empty(B, Cin, H, W, device="cuda")
        Hh.call("yogo_bf16_8c_to_nchw_f32", x8, x, B, Cin, H * W, st)
        if kind == "fwd_plain":
            want = F.conv2d(x.double().cpu(), w.to(torch.bfloat16).double().cpu(), bias.double().cpu(), padding=1)
        else:
            wf = torch.randn(Cin, 128, 3, 3, device="cuda", generator=g) * 0.05
            want = F.conv_transpose2d(x.double().cpu(), wf.to(torch.bfloat16).double().cpu(), padding=1)
        got = torch.empty(B, 128, H, W, device="cuda")
        Hh.call("yogo_bf16_8c_to_nchw_f32", y_new, got, B, 128, H * W, st)
        got = got.cpu().double()
        tol = 2.0 ** -8 * want.abs() + 2e-5 * float(want.abs().max())
        assert bool(((got - want).abs() <= tol).all()), float(((got - want).abs() - tol).max())


def test_product_plan_of_the_plain_epilogue_launches():
    """the product library (no plan switch): the bias-free data gradients of the 128 -> 128 layers run on conv_bf16_ws16_kernel, the biased
    forward (layer 5) on conv_bf16_ws_kernel<0> -- and give the bits the hooks library gives for the same plan"""
    y_d, _, log_d = _run(True, "dgrad", 2, 128, 97, 129, seed=29, product=True)
    y_f, _, log_f = _run(True, "fwd_plain", 2, 128, 97, 129, seed=29, product=True)
    assert any(ln.startswith("conv_bf16_ws16_kernel<false>") for ln in log_d), log_d
    assert any(ln.startswith("conv_bf16_ws_kernel<0>") for ln in log_f), log_f
    y_d2, _, _ = _run(True, "dgrad", 2, 128, 97, 129, seed=29)                   # hooks library, 16x16x32 everywhere
    y_f2, _, _ = _run(True, "fwd_plain", 2, 128, 97, 129, seed=29, ws16=False)   # hooks library, 32x32x16
    assert torch.equal(y_d.view(torch.int16), y_d2.view(torch.int16)) and torch.equal(y_f.view(torch.int16), y_f2.view(torch.int16))


def test_persistent_kernel_repeats_itself():
    """two launches on the same input give the same bits (no read of a buffer before its LDS-DMA has landed)"""
    a, sa, _ = _run(True, "fwd_signs", 16, 128, 97, 129, seed=5)
    for _ in range(3):
        b, sb, _ = _run(True, "fwd_signs", 16, 128, 97, 129, seed=5)
        assert torch.equal(a.view(torch.int16), b.view(torch.int16)) and torch.equal(sa, sb)


@pytest.mark.parametrize("mode,B,Cin,H,W", [(4, 2, 128, 23, 37), (5, 2, 64, 19, 45), (4, 1, 96, 40, 41), (5, 3, 128, 9, 11)])
def test_scaled_modes_against_cpu_fp32(mode, B, Cin, H, W):
    """conv_bf16_ws_kernel<4> (channel scale, no activation: the data gradient into a Dropout2d block) and <5> (LeakyReLU + channel
    scale, no sign map: an eval-style forward with a mask) are never launched by base_model's training step, so beside the
    bit-identity with the tiled kernel they get an independent reference here: a CPU fp32 convolution of the same bf16-rounded
    operands (yogo/model_defns.py:49-65's blocks: conv + bias -> LeakyReLU -> Dropout2d scale), one bf16 rounding of the output."""
    import torch.nn.functional as F

    from yogo_amd import _hip as Hh

    st = Hh.stream_ptr()
    g = torch.Generator().manual_seed(100 + mode + H)
    w = (torch.randn(128, Cin, 3, 3, generator=g) * 0.05).to(torch.bfloat16).float()
    x = torch.randn(B, Cin, H, W, generator=g).to(torch.bfloat16).float()
    bias = torch.randn(128, generator=g)
    msk = (torch.rand(B, 128, generator=g) > 0.2).float() / 0.8
    x8 = torch.empty(B, _blocks(Cin), H, W, 8, dtype=torch.bfloat16, device="cuda")
    Hh.call("yogo_nchw_f32_to_bf16_8c", x.cuda(), x8, B, Cin, H * W, st)
    y8 = torch.full((B, 16, H, W, 8), 7.0, device="cuda").to(torch.bfloat16)
    Hh.launch_log(True)
    try:
        if mode == 5:
            packed = torch.empty(Hh.query_size("yogo_conv_bf16_packed_bytes", Cin, 128, 3, 0), dtype=torch.uint8, device="cuda")
            Hh.call("yogo_conv_bf16_pack", w.cuda(), None, packed, Cin, 128, 3, 0, st)
            Hh.call("yogo_conv2d_fwd_bf16", x8, packed, bias.cuda(), y8, None, msk.cuda(), None, B, Cin, 128, H, W, 3, 1, 1, st)
            want = F.leaky_relu(F.conv2d(x.double(), w.double(), bias.double(), padding=1), 0.01) * msk.double()[:, :, None, None]
        else:
            # data gradient of a forward conv 128 -> Cin: dy has Cin channels, dx 128; dx = conv_transpose(dy, wf) * mask
            wf = (torch.randn(Cin, 128, 3, 3, generator=g) * 0.05).to(torch.bfloat16).float()
            packed = torch.empty(Hh.query_size("yogo_conv_bf16_packed_bytes", 128, Cin, 3, 1), dtype=torch.uint8, device="cuda")
            Hh.call("yogo_conv_bf16_pack", wf.cuda(), None, packed, 128, Cin, 3, 1, st)
            Hh.call("yogo_conv2d_dgrad_bf16", x8, packed, y8, None, 0, msk.cuda(), B, 128, Cin, H, W, 3, 1, st)
            want = F.conv_transpose2d(x.double(), wf.double(), padding=1) * msk.double()[:, :, None, None]
        torch.cuda.synchronize()
        log = Hh.read_launch_log()
    finally:
        Hh.launch_log(False)
    assert any(ln.startswith(f"conv_bf16_ws_kernel<{mode}>") for ln in log), log
    got = torch.empty(B, 128, H, W, device="cuda")
    Hh.call("yogo_bf16_8c_to_nchw_f32", y8, got, B, 128, H * W, st)
    got = got.cpu().double()
    # one bf16 rounding of an fp32-accumulated value: |d| <= 2^-8 |want| (+ fp32 accumulation noise of a K = 9 Cin contraction)
    tol = 2.0 ** -8 * want.abs() + 2e-5 * float(want.abs().max())
    assert bool(((got - want).abs() <= tol).all()), float(((got - want).abs() - tol).max())


# ---- the stride-2 forward: conv_bf16_ws3_kernel against conv_bf16_kernel<4, 1, 8> -------------------------------------------------
def _run_s2f(persistent, kind, B, Cin, H, W, seed):
    """y [B][128][ceil(H/2)][ceil(W/2)] = [mask x] [LeakyReLU] (conv3x3 stride 2 (x [B][Cin][H][W]) + bias): yogo/model_defns.py:54-56"""
    import contextlib

    from _util import hooks_library
    from yogo_amd import _hip as Hh

    with (contextlib.nullcontext() if persistent else hooks_library()):
        st = Hh.stream_ptr()
        g = torch.Generator(device="cuda").manual_seed(seed)
        OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        w = torch.randn(128, Cin, 3, 3, device="cuda", generator=g) * 0.05
        x8 = torch.randn(B, _blocks(Cin), H, W, 8, device="cuda", generator=g).to(torch.bfloat16)
        y8 = torch.full((B, 16, OH, OW, 8), 7.0, device="cuda").to(torch.bfloat16)   # poisoned: every unit must be written
        bias = torch.randn(128, device="cuda", generator=g)
        msk = (torch.rand(B, 128, device="cuda", generator=g) > 0.2).float() / 0.8
        packed = torch.empty(Hh.query_size("yogo_conv_bf16_packed_bytes", Cin, 128, 3, 0), dtype=torch.uint8, device="cuda")
        Hh.call("yogo_conv_bf16_pack", w, None, packed, Cin, 128, 3, 0, st)
        if not persistent:
            Hh.call("yogo_hook_conv_bf16_persistent", 0)
        Hh.launch_log(True)
        try:
            Hh.call("yogo_conv2d_fwd_bf16", x8, packed, None if kind == "nobias" else bias, y8, None, msk if kind == "mask" else None, None, B, Cin, 128, H, W, 3, 2,
                    1 if kind in ("leaky", "mask") else 0, st)
            torch.cuda.synchronize()
            log = Hh.read_launch_log()
        finally:
            Hh.launch_log(False)
        return y8, log, (w, x8, bias, msk)


S2F_CASES = [
    # (kind, B, Cin, H, W)
    ("nobias", 2, 128, 193, 258),     # layer 4 of base_model at 772x1032
    ("plain", 1, 128, 20, 91),        # one band of 46 columns
    ("leaky", 2, 64, 37, 41),         # odd sizes, 4 chunks
    ("mask", 3, 128, 50, 66),
    ("plain", 40, 128, 97, 129),      # every workgroup walks several tiles, image changes at the seams
    ("leaky", 1, 96, 9, 300),         # short and wide: several bands, 6 chunks
    ("plain", 2, 128, 300, 5),        # tall and narrow: three output columns
    ("mask", 5, 128, 3, 3),           # two output pixels per row
    ("plain", 3, 128, 64, 64),        # even sizes
    ("leaky", 2, 256, 33, 95),        # 16 chunks
]


@pytest.mark.parametrize("kind,B,Cin,H,W", S2F_CASES)
def test_persistent_stride2_forward_is_bit_identical_to_the_tiled_kernel(kind, B, Cin, H, W):
    y_old, log_old, _ = _run_s2f(False, kind, B, Cin, H, W, seed=41)
    y_new, log_new, _ = _run_s2f(True, kind, B, Cin, H, W, seed=41)
    assert any(ln.startswith("conv_bf16_kernel<4, 1, 8, false") for ln in log_old), log_old
    assert any(ln.startswith("conv_bf16_ws3_kernel<") for ln in log_new), log_new
    plan = next(ln for ln in log_old if ln.startswith("conv_bf16_kernel<4, 1, 8, false"))
    if " CKb=2 " in plan:   # the tiled kernel stepped through K in 16-channel chunks too: the same MFMA sequence per accumulator
        assert torch.equal(y_old.view(torch.int16), y_new.view(torch.int16)), (
            f"{(y_old.float() - y_new.float()).abs().max().item()} max abs difference, "
            f"{(y_old.view(torch.int16) != y_new.view(torch.int16)).float().mean().item()} of the values differ")
    else:
        a, b = y_old.float(), y_new.float()
        ulp = 2.0 ** -7 * torch.maximum(a.abs(), b.abs()) + 1e-6 * a.abs().max()
        assert bool(((a - b).abs() <= ulp).all()), f"{((a - b).abs() - ulp).max().item()} beyond one bf16 step"
        assert (a != b).float().mean().item() < 5e-3


@pytest.mark.parametrize("kind,B,Cin,H,W", [("mask", 2, 128, 45, 53), ("plain", 1, 64, 33, 70), ("leaky", 2, 128, 18, 16)])
def test_persistent_stride2_forward_against_cpu_fp32(kind, B, Cin, H, W):
    """an independent reference for the new kernel: torch's CPU conv2d (float64) on the same bf16-rounded operands, one bf16 rounding of the result"""
    import torch.nn.functional as F

    from yogo_amd import _hip as Hh

    y_new, log, (w, x8, bias, msk) = _run_s2f(True, kind, B, Cin, H, W, seed=43)
    assert any(ln.startswith("conv_bf16_ws3_kernel<") for ln in log), log
    st = Hh.stream_ptr()
    OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    x = torch.empty(B, Cin, H, W, device="cuda")
    Hh.call("yogo_bf16_8c_to_nchw_f32", x8, x, B, Cin, H * W, st)
    got = torch.empty(B, 128, OH, OW, device="cuda")
    Hh.call("yogo_bf16_8c_to_nchw_f32", y_new, got, B, 128, OH * OW, st)
    want = F.conv2d(x.double().cpu(), w.to(torch.bfloat16).double().cpu(), bias.double().cpu(), stride=2, padding=1)
    if kind in ("leaky", "mask"):
        want = F.leaky_relu(want, 0.01)
    if kind == "mask":
        want = want * msk.double().cpu()[:, :, None, None]
    got = got.cpu().double()
    tol = 2.0 ** -8 * want.abs() + 2e-5 * float(want.abs().max())
    assert bool(((got - want).abs() <= tol).all()), float(((got - want).abs() - tol).max())


# ---- the direct stride-2 data gradient of the thin layers: conv_bf16_s2d_direct_kernel against conv_bf16_kernel<1, 2, 4, S2D> ----
def _run_s2d_thin(direct, kind, B, K, M, OH, OW, seed):
    """dx [B][M <= 32][OH][OW] = conv_transpose(dy [B][K][ceil(OH/2)][ceil(OW/2)]) x (LeakyReLU'(sign map) x) channel mask: autograd of
    yogo/model_defns.py:44-46 (conv 32 -> 64, stride 2) into the LeakyReLU + Dropout2d block in front of it"""
    import contextlib

    from _util import hooks_library
    from yogo_amd import _hip as Hh

    with (contextlib.nullcontext() if direct else hooks_library()):
        st = Hh.stream_ptr()
        g = torch.Generator(device="cuda").manual_seed(seed)
        IH, IW = (OH + 1) // 2, (OW + 1) // 2
        wf = torch.randn(K, M, 3, 3, device="cuda", generator=g) * 0.05
        dy8 = torch.randn(B, _blocks(K), IH, IW, 8, device="cuda", generator=g).to(torch.bfloat16)
        dx8 = torch.full((B, _blocks(M), OH, OW, 8), 7.0, device="cuda").to(torch.bfloat16)   # poisoned: every unit must be written
        msk = (torch.rand(B, M, device="cuda", generator=g) > 0.2).float() / 0.8
        sg = torch.randint(0, 256, (Hh.query_size("yogo_bf16_signs_bytes", B, M, OH, OW),), dtype=torch.uint8, device="cuda", generator=g)
        packed = torch.empty(Hh.query_size("yogo_conv_bf16_packed_bytes", M, K, 3, 2), dtype=torch.uint8, device="cuda")
        Hh.call("yogo_conv_bf16_pack", wf, None, packed, M, K, 3, 2, st)
        if not direct:
            Hh.call("yogo_hook_conv_bf16_direct", 0)
        Hh.launch_log(True)
        try:
            if kind == "signs":
                Hh.call("yogo_conv2d_dgrad_bf16_signs", dy8, packed, dx8, sg, msk, B, M, K, OH, OW, 3, 2, st)
            else:
                Hh.call("yogo_conv2d_dgrad_bf16", dy8, packed, dx8, None, 0, msk if kind == "mask" else None, B, M, K, OH, OW, 3, 2, st)
            torch.cuda.synchronize()
            log = Hh.read_launch_log()
        finally:
            Hh.launch_log(False)
            if not direct:
                Hh.call("yogo_hook_conv_bf16_direct", 1)
        return dx8, log, (wf, dy8, msk, sg)


S2D_THIN_CASES = [
    # (kind, B, K, M, OH, OW)
    ("signs", 2, 64, 32, 386, 516),   # layer 2 of base_model at 772x1032
    ("signs", 1, 64, 32, 20, 22),
    ("mask", 2, 64, 32, 37, 41),      # odd sizes: the last quad row / column has no odd member
    ("plain", 3, 32, 32, 50, 66),     # two 16-channel steps
    ("signs", 5, 64, 24, 33, 29),     # 24 real channels: the padding channels of the last block come out as zeros
    ("plain", 4, 32, 16, 2, 2),       # a single quad, two channel blocks
    ("signs", 3, 64, 32, 64, 64),
    # 128 channels in two passes of 64 (layer 4 of base_model: one workgroup per CU, 147 KB of weight slices each)
    ("signs", 2, 128, 128, 193, 258),
    ("signs", 1, 128, 128, 20, 22),
    ("mask", 2, 128, 128, 37, 41),
    ("plain", 3, 128, 96, 50, 66),    # 96 real channels: the second pass stores two of its four block pairs
    ("signs", 40, 128, 128, 30, 30),  # every wavefront walks several tiles, image changes at the seams
    ("mask", 5, 128, 72, 3, 2),
]


@pytest.mark.parametrize("kind,B,K,M,OH,OW", S2D_THIN_CASES)
def test_direct_stride2_dgrad_against_the_tiled_kernel_and_cpu(kind, B, K, M, OH, OW):
    import torch.nn.functional as F

    from yogo_amd import _hip as Hh

    d_old, log_old, _ = _run_s2d_thin(False, kind, B, K, M, OH, OW, seed=53)
    d_new, log_new, (wf, dy8, msk, sg) = _run_s2d_thin(True, kind, B, K, M, OH, OW, seed=53)
    assert any(ln.startswith("conv_bf16_kernel<") and "s2d=1" in ln for ln in log_old), log_old
    assert any(ln.startswith("conv_bf16_s2d_direct_kernel<") for ln in log_new), log_new
    # the direct kernel steps the contraction 16 channels at a time, the tiled plan of these shapes 32 or 64 (tap-major inside a chunk): another
    # fp32 summation order -- the bf16 results may differ by one rounding step on a few values
    a, b = d_old.float(), d_new.float()
    ulp = 2.0 ** -7 * torch.maximum(a.abs(), b.abs()) + 1e-6 * a.abs().max()
    assert bool(((a - b).abs() <= ulp).all()), f"{((a - b).abs() - ulp).max().item()} beyond one bf16 step"
    assert (a != b).float().mean().item() < 5e-3
    # ... and an independent reference: torch's CPU conv_transpose2d (float64) on the same bf16-rounded operands
    if B * OH * OW <= 40000:
        st = Hh.stream_ptr()
        IH, IW = (OH + 1) // 2, (OW + 1) // 2
        dy = torch.empty(B, K, IH, IW, device="cuda")
        Hh.call("yogo_bf16_8c_to_nchw_f32", dy8, dy, B, K, IH * IW, st)
        got = torch.empty(B, M, OH, OW, device="cuda")
        Hh.call("yogo_bf16_8c_to_nchw_f32", d_new, got, B, M, OH * OW, st)
        w = wf.to(torch.bfloat16).double().cpu()
        want = F.conv_transpose2d(dy.double().cpu(), w, stride=2, padding=1, output_padding=(OH - ((IH - 1) * 2 + 1), OW - ((IW - 1) * 2 + 1)))
        if kind == "signs":
            sq = 2 if M <= 32 else 8
            s = sg.cpu().view(B, 2, OH, OW, sq).long()   # byte (h, pixel, q), bit i + 4e = (channel 16 q + 8 e + 4 h + i > 0)
            pos = torch.zeros(B, 16 * sq, OH, OW, dtype=torch.bool)
            for h in range(2):
                for q in range(sq):
                    for e in range(2):
                        for i in range(4):
                            pos[:, 16 * q + 8 * e + 4 * h + i] = ((s[:, h, :, :, q] >> (i + 4 * e)) & 1).bool()
            want = want * torch.where(pos[:, :M], 1.0, 0.01)
        if kind in ("signs", "mask"):
            want = want * msk.double().cpu()[:, :, None, None]
        got = got.cpu().double()
        tol = 2.0 ** -8 * want.abs() + 2e-5 * float(want.abs().max())
        assert bool(((got - want).abs() <= tol).all()), float(((got - want).abs() - tol).max())
        padc = d_new.float().permute(0, 1, 4, 2, 3).reshape(B, -1, OH, OW)[:, M:]
        assert padc.numel() == 0 or float(padc.abs().max()) == 0.0


# ---- the independent-wavefront kernel of the thin forward-type layers: conv_bf16_staged_kernel against conv_bf16_kernel ----
def _run_thin(staged, kind, B, Cin, Cout, H, W, s, seed):
    """y [B][Cout <= 64][OH][OW] = [mask x] [LeakyReLU] (conv3x3 stride s (x [B][Cin = 16 / 32][H][W]) + bias) [+ the sign map of y]:
    yogo/model_defns.py:36-46; kind "dgrad": dx = conv3x3(dy, flipped transposed weights), the stride-1 data gradient of model_defns.py:36-40"""
    import contextlib

    from _util import hooks_library
    from yogo_amd import _hip as Hh

    with (contextlib.nullcontext() if staged else hooks_library()):
        st = Hh.stream_ptr()
        g = torch.Generator(device="cuda").manual_seed(seed)
        OH, OW = (H - 1) // s + 1, (W - 1) // s + 1
        x8 = torch.randn(B, _blocks(Cin), H, W, 8, device="cuda", generator=g).to(torch.bfloat16)
        y8 = torch.full((B, _blocks(Cout), OH, OW, 8), 7.0, device="cuda").to(torch.bfloat16)   # poisoned: every unit must be written
        bias = torch.randn(Cout, device="cuda", generator=g)
        msk = (torch.rand(B, Cout, device="cuda", generator=g) > 0.2).float() / 0.8
        sg = torch.full((Hh.query_size("yogo_bf16_signs_bytes", B, Cout, OH, OW),), 0xA5, dtype=torch.uint8, device="cuda")
        if kind == "dgrad":   # the layer is Cout -> Cin (its weight [Cin][Cout][3][3]); x8 plays dy, y8 plays dx
            w = torch.randn(Cin, Cout, 3, 3, device="cuda", generator=g) * 0.05
            packed = torch.empty(Hh.query_size("yogo_conv_bf16_packed_bytes", Cout, Cin, 3, 1), dtype=torch.uint8, device="cuda")
            Hh.call("yogo_conv_bf16_pack", w, None, packed, Cout, Cin, 3, 1, st)
        else:
            w = torch.randn(Cout, Cin, 3, 3, device="cuda", generator=g) * 0.05
            packed = torch.empty(Hh.query_size("yogo_conv_bf16_packed_bytes", Cin, Cout, 3, 0), dtype=torch.uint8, device="cuda")
            Hh.call("yogo_conv_bf16_pack", w, None, packed, Cin, Cout, 3, 0, st)
        if not staged:
            Hh.call("yogo_hook_conv_bf16_staged", 0)
        Hh.launch_log(True)
        try:
            if kind == "signs":
                Hh.call("yogo_conv2d_fwd_bf16_signs", x8, packed, bias, y8, sg, msk, B, Cin, Cout, H, W, 3, s, 1, st)
            elif kind == "dgrad":
                Hh.call("yogo_conv2d_dgrad_bf16", x8, packed, y8, None, 0, None, B, Cout, Cin, H, W, 3, 1, st)
            else:
                Hh.call("yogo_conv2d_fwd_bf16", x8, packed, None if kind == "nobias" else bias, y8, None, msk if kind == "mask" else None, None, B, Cin, Cout, H, W, 3, s,
                        1 if kind in ("leaky", "mask") else 0, st)
            torch.cuda.synchronize()
            log = Hh.read_launch_log()
        finally:
            Hh.launch_log(False)
        return y8, sg, log, (w, x8, bias, msk)


THIN_CASES = [
    # (kind, B, Cin, Cout, H, W, stride)
    ("signs", 2, 16, 32, 386, 516, 1),    # layer 1 of base_model at 772x1032
    ("signs", 2, 32, 64, 386, 516, 2),    # layer 2
    ("signs", 1, 16, 32, 20, 22, 1),
    ("signs", 1, 32, 64, 20, 22, 2),
    ("mask", 2, 32, 64, 37, 41, 2),       # odd sizes
    ("plain", 3, 32, 32, 50, 66, 2),
    ("leaky", 5, 16, 24, 33, 29, 2),      # 24 real channels: the padding channels of the last block come out as zeros
    ("leaky", 5, 16, 24, 33, 29, 1),      # (8-row tiles: the last tile row has one row)
    ("nobias", 4, 32, 48, 2, 2, 2),
    ("signs", 3, 32, 64, 64, 64, 2),      # even sizes: the last column's right tap is the padding
    ("signs", 40, 16, 64, 30, 30, 2),     # every wavefront walks several tiles, image changes at the seams
    ("signs", 40, 16, 32, 30, 30, 1),
    ("mask", 1, 16, 8, 5, 300, 2),        # one channel block pair, short and wide
    ("leaky", 2, 32, 64, 1, 70, 2),       # a single row
    ("leaky", 2, 16, 32, 1, 70, 1),
    ("nobias", 4, 16, 20, 2, 2, 1),
    ("plain", 3, 16, 32, 50, 66, 1),
    ("signs", 2, 32, 40, 70, 1, 2),       # a single column
    ("signs", 2, 16, 32, 70, 1, 1),
    ("plain", 2, 16, 32, 193, 258, 2),
]


@pytest.mark.parametrize("kind,B,Cin,Cout,H,W,s", THIN_CASES)
def test_staged_thin_convolution_against_the_tiled_kernel_and_cpu(kind, B, Cin, Cout, H, W, s):
    import torch.nn.functional as F

    from yogo_amd import _hip as Hh

    y_old, sg_old, log_old, _ = _run_thin(False, kind, B, Cin, Cout, H, W, s, seed=61)
    y_new, sg_new, log_new, (w, x8, bias, msk) = _run_thin(True, kind, B, Cin, Cout, H, W, s, seed=61)
    assert any(ln.startswith("conv_bf16_kernel<") for ln in log_old), log_old
    assert any(ln.startswith("conv_bf16_staged_kernel<") for ln in log_new), log_new
    plan = next(ln for ln in log_old if ln.startswith("conv_bf16_kernel<"))
    a, b = y_old.float(), y_new.float()
    if " CKb=2 " in plan:   # the tiled kernel stepped through K in 16-channel chunks too: the same MFMA sequence per accumulator
        assert torch.equal(y_old.view(torch.int16), y_new.view(torch.int16)), f"{(a - b).abs().max().item()} max abs difference, {(a != b).float().mean().item()} differ"
        if kind == "signs":
            assert torch.equal(sg_old, sg_new)
    else:
        ulp = 2.0 ** -7 * torch.maximum(a.abs(), b.abs()) + 1e-6 * a.abs().max()
        assert bool(((a - b).abs() <= ulp).all()), f"{((a - b).abs() - ulp).max().item()} beyond one bf16 step"
        assert (a != b).float().mean().item() < 5e-3
    OH, OW = (H - 1) // s + 1, (W - 1) // s + 1
    if kind == "signs":   # the sign map is the sign of what was stored: byte (h, pixel, q), bit i + 4e = (channel 16 q + 8 e + 4 h + i > 0)
        sq = 2 if Cout <= 32 else 4
        sb = sg_new.view(B, 2, OH, OW, sq).long()
        yv = y_new.float().permute(0, 1, 4, 2, 3).reshape(B, -1, OH, OW)   # [B][channels padded to 16][OH][OW]
        for ch in range(0, Cout, 5):
            q, e, h, i = ch // 16, (ch % 16) // 8, (ch % 8) // 4, ch % 4
            bit = ((sb[:, h, :, :, q] >> (i + 4 * e)) & 1).bool()
            assert torch.equal(bit, yv[:, ch] > 0), ch
    padc = y_new.float().permute(0, 1, 4, 2, 3).reshape(B, -1, OH, OW)[:, Cout:]
    assert padc.numel() == 0 or float(padc.abs().max()) == 0.0
    # ... and an independent reference: torch's CPU conv2d (float64) on the same bf16-rounded operands, one bf16 rounding of the result
    if B * OH * OW <= 40000:
        st = Hh.stream_ptr()
        x = torch.empty(B, Cin, H, W, device="cuda")
        Hh.call("yogo_bf16_8c_to_nchw_f32", x8, x, B, Cin, H * W, st)
        got = torch.empty(B, Cout, OH, OW, device="cuda")
        Hh.call("yogo_bf16_8c_to_nchw_f32", y_new, got, B, Cout, OH * OW, st)
        wd = w.to(torch.bfloat16).double().cpu()
        if kind == "dgrad":
            want = F.conv_transpose2d(x.double().cpu(), wd, stride=1, padding=1)
        else:
            want = F.conv2d(x.double().cpu(), wd, None if kind == "nobias" else bias.double().cpu(), stride=s, padding=1)
        if kind in ("leaky", "mask", "signs"):
            want = F.leaky_relu(want, 0.01)
        if kind in ("mask", "signs"):
            want = want * msk.double().cpu()[:, :, None, None]
        got = got.cpu().double()
        tol = 2.0 ** -8 * want.abs() + 2e-5 * float(want.abs().max())
        assert bool(((got - want).abs() <= tol).all()), float(((got - want).abs() - tol).max())


# ---- the 1x1 head's forward with its weights in registers: conv_bf16_1x1_f32_kernel against conv_bf16_kernel ----
def _run_head(new, kind, B, Cin, Cout, H, W, seed):
    """forward: pred fp32 [B][Cout <= 32][H][W] = conv1x1(x bf16 [B][Cin][H][W]) + bias (yogo/model_defns.py:66);
    data gradient: dx bf16 [B][Cin][H][W] = conv1x1^T(dpred bf16 [B][Cout <= 16]) x LeakyReLU'(sign map of x) [x channel mask]"""
    import contextlib

    from _util import hooks_library
    from yogo_amd import _hip as Hh

    with (contextlib.nullcontext() if new else hooks_library()):
        st = Hh.stream_ptr()
        g = torch.Generator(device="cuda").manual_seed(seed)
        w = torch.randn(Cout, Cin, 1, 1, device="cuda", generator=g) * 0.1
        bias = torch.randn(Cout, device="cuda", generator=g)
        if not new:
            Hh.call("yogo_hook_conv_bf16_head", 0)
        Hh.launch_log(True)
        try:
            if kind == "fwd":
                x8 = torch.randn(B, _blocks(Cin), H, W, 8, device="cuda", generator=g).to(torch.bfloat16)
                out = torch.full((B, Cout, H, W), 7.0, device="cuda")
                packed = torch.empty(Hh.query_size("yogo_conv_bf16_packed_bytes", Cin, Cout, 1, 0), dtype=torch.uint8, device="cuda")
                Hh.call("yogo_conv_bf16_pack", w, None, packed, Cin, Cout, 1, 0, st)
                Hh.call("yogo_conv2d_fwd_bf16", x8, packed, bias, None, out, None, None, B, Cin, Cout, H, W, 1, 1, 0, st)
                aux = (w, x8, bias, None, None)
            else:
                dy8 = torch.randn(B, _blocks(Cout), H, W, 8, device="cuda", generator=g).to(torch.bfloat16)
                if Cout % 16:   # the padding channels of a gradient tensor are zeros (what decode_loss_bwd_bf16_kernel writes)
                    dyf = dy8.float().permute(0, 1, 4, 2, 3).reshape(B, -1, H, W)
                    dyf[:, Cout:] = 0
                    dy8 = dyf.reshape(B, _blocks(Cout), 8, H, W).permute(0, 1, 3, 4, 2).contiguous().to(torch.bfloat16)
                out = torch.full((B, _blocks(Cin), H, W, 8), 7.0, device="cuda").to(torch.bfloat16)
                msk = (torch.rand(B, Cin, device="cuda", generator=g) > 0.2).float() / 0.8
                sg = torch.randint(0, 256, (Hh.query_size("yogo_bf16_signs_bytes", B, Cin, H, W),), dtype=torch.uint8, device="cuda", generator=g)
                packed = torch.empty(Hh.query_size("yogo_conv_bf16_packed_bytes", Cin, Cout, 1, 1), dtype=torch.uint8, device="cuda")
                Hh.call("yogo_conv_bf16_pack", w, None, packed, Cin, Cout, 1, 1, st)
                if kind == "dgrad_signs":
                    Hh.call("yogo_conv2d_dgrad_bf16_signs", dy8, packed, out, sg, None, B, Cin, Cout, H, W, 1, 1, st)
                elif kind == "dgrad_signs_mask":
                    Hh.call("yogo_conv2d_dgrad_bf16_signs", dy8, packed, out, sg, msk, B, Cin, Cout, H, W, 1, 1, st)
                else:
                    Hh.call("yogo_conv2d_dgrad_bf16", dy8, packed, out, None, 0, msk if kind == "dgrad_mask" else None, B, Cin, Cout, H, W, 1, 1, st)
                aux = (w, dy8, None, msk, sg)
            torch.cuda.synchronize()
            log = Hh.read_launch_log()
        finally:
            Hh.launch_log(False)
        return out, log, aux


HEAD_CASES = [
    # (kind, B, Cin, Cout, H, W)
    ("fwd", 2, 128, 12, 97, 129),            # the head of base_model at 772x1032 (7 classes)
    ("fwd", 3, 128, 12, 5, 7),
    ("fwd", 2, 64, 9, 31, 33),               # 4 classes, 4 steps
    ("fwd", 1, 32, 25, 20, 22),              # more than 16 output channels
    ("fwd", 40, 128, 12, 12, 13),            # every wavefront walks several tiles, image changes at the seams
    ("fwd", 2, 16, 32, 9, 11),               # one step, all 32 output channels
]


@pytest.mark.parametrize("kind,B,Cin,Cout,H,W", HEAD_CASES)
def test_head_kernels_are_bit_identical_to_the_tiled_kernel_and_match_cpu(kind, B, Cin, Cout, H, W):
    from yogo_amd import _hip as Hh

    o_old, log_old, _ = _run_head(False, kind, B, Cin, Cout, H, W, seed=71)
    o_new, log_new, (w, x8, bias, msk, sg) = _run_head(True, kind, B, Cin, Cout, H, W, seed=71)
    assert any(ln.startswith("conv_bf16_kernel<") for ln in log_old), log_old
    name = "conv_bf16_1x1_f32_kernel<" if kind == "fwd" else "conv_bf16_1x1_dgrad_kernel<"
    assert any(ln.startswith(name) for ln in log_new), log_new
    plan = next(ln for ln in log_old if ln.startswith("conv_bf16_kernel<"))
    if kind == "fwd":
        if " CKb=2 " in plan or Cin == 16:
            assert torch.equal(o_old, o_new), float((o_old - o_new).abs().max())
        else:
            assert float((o_old - o_new).abs().max()) <= 2e-5 * float(o_old.abs().max())
    else:
        assert torch.equal(o_old.view(torch.int16), o_new.view(torch.int16)), float((o_old.float() - o_new.float()).abs().max())
    # ... and an independent CPU reference (float64 on the same bf16-rounded operands)
    st = Hh.stream_ptr()
    wd = w.to(torch.bfloat16).double().cpu()[:, :, 0, 0]
    if kind == "fwd":
        x = torch.empty(B, Cin, H, W, device="cuda")
        Hh.call("yogo_bf16_8c_to_nchw_f32", x8, x, B, Cin, H * W, st)
        want = torch.einsum("oc,bchw->bohw", wd, x.double().cpu()) + bias.double().cpu()[None, :, None, None]
        got = o_new.double().cpu()
        assert float((got - want).abs().max()) <= 2e-5 * float(want.abs().max())
    else:
        dy = torch.empty(B, Cout, H, W, device="cuda")
        Hh.call("yogo_bf16_8c_to_nchw_f32", x8, dy, B, Cout, H * W, st)
        want = torch.einsum("oc,bohw->bchw", wd, dy.double().cpu())
        if "signs" in kind:
            sq = 2 if Cin <= 32 else (4 if Cin <= 64 else 8)
            s = sg.cpu().view(B, 2, H, W, sq).long()   # byte (h, pixel, q), bit i + 4e = (channel 16 q + 8 e + 4 h + i > 0)
            pos = torch.zeros(B, 16 * sq, H, W, dtype=torch.bool)
            for h in range(2):
                for q in range(sq):
                    for e in range(2):
                        for i in range(4):
                            pos[:, 16 * q + 8 * e + 4 * h + i] = ((s[:, h, :, :, q] >> (i + 4 * e)) & 1).bool()
            want = want * torch.where(pos[:, :Cin], 1.0, 0.01)
        if "mask" in kind:
            want = want * msk.double().cpu()[:, :, None, None]
        got = torch.empty(B, Cin, H, W, device="cuda")
        Hh.call("yogo_bf16_8c_to_nchw_f32", o_new, got, B, Cin, H * W, st)
        got = got.cpu().double()
        tol = 2.0 ** -8 * want.abs() + 2e-5 * float(want.abs().max())
        assert bool(((got - want).abs() <= tol).all()), float(((got - want).abs() - tol).max())
        padc = o_new.float().permute(0, 1, 4, 2, 3).reshape(B, -1, H, W)[:, Cin:]
        assert padc.numel() == 0 or float(padc.abs().max()) == 0.0
