"""Parity at PRODUCTION shapes (BASELINE configs[2]: base_model, 772x1032x1, per-GPU batch 128, bf16).

The small-shape tests of test_gpu_bf16.py never reach the kernel instantiations and tilings a 772x1032 batch selects (e.g.
conv_bf16_kernel<2,4,8,...> needs OH*OW >= 4096; the planner picks other band widths / chunk depths / slot counts).  Here
  (A) every layer of base_model runs every direction (forward, data gradient, weight gradient -- with the sign-map / bias /
      fp32-head variants the training step uses) at 772x1032 against F.conv2d + autograd in fp32 ON THE CPU on bf16-rounded
      inputs, with the per-kernel bf16 tolerances of test_gpu_bf16.py (8e-3 of the output range for one bf16 rounding, 1e-4
      of max|g| for fp32-accumulated weight gradients); weight gradients also at B = 128 (the split-K plan depends on B);
  (B) one full bf16 HipTrainer.step at 772x1032 is compared with the CPU oracle's bf16-storage emulation of the step
      (O.bf16_train_step rounds where the HIP path stores bf16): end to end loss 1e-3, every gradient tensor cosine >= 0.995,
      running statistics 1e-3 -- and teacher-forced, every kernel of the step given its actual inputs: stored tensors to one
      bf16 ulp, parameter gradients to 2e-4 of max|g| (tests/_util.py says why the elementwise bound is applied per kernel);
  (C) the production batch 128 -- exactly bench.py's step -- with the launch log proving which instantiations / planner
      parameters ran: they must include every conv_bf16_kernel<...> and wgrad_bf16_kernel<...> row of the committed
      rocprofv3 summary (profiles/r*_kernel_stats.txt) and nothing the per-layer tests did not launch too.  Numerics through
      a size-independent property: a batch of 64 copies of (B)'s two images has the same batch statistics, loss and
      (mean) gradients as the two images, so the B = 128 step must reproduce the oracle's B = 2 step.
(torch's own GPU convolutions are deliberately not used as a reference: MIOpen has no precompiled gfx950 kernels in this
image and would spend minutes compiling.)
Reference rows: yogo/model_defns.py:30-77 (blocks), yogo/train.py:309-325 (step)."""
import glob
import os
import re

import pytest
import torch
import torch.nn.functional as F

import yogo_oracle as O

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HI, WI, C = 772, 1032, 7
# (cin, cout, k, stride, IH, IW, bias, fwd variant, dgrad variant) of layers 1..7 as HipTrainer(half=True) calls them:
#   fwd "signs" = LeakyReLU block without BatchNorm (writes the sign map), "plain" = conv only (BatchNorm follows), "head" = fp32 out
#   dgrad "signs" = previous block is LeakyReLU without BatchNorm (reads its sign map), "plain" = previous block has BatchNorm
LAYERS = {
    1: (16, 32, 3, 1, 386, 516, True, "signs", "plain"),
    2: (32, 64, 3, 2, 386, 516, True, "signs", "signs"),
    3: (64, 128, 3, 1, 193, 258, True, "signs", "signs"),
    4: (128, 128, 3, 2, 193, 258, False, "plain", "signs"),
    5: (128, 128, 3, 1, 97, 129, True, "plain", "plain"),
    6: (128, 128, 3, 1, 97, 129, True, "signs", "plain"),
    7: (128, 5 + C, 1, 1, 97, 129, True, "head", "signs"),
}
SEEN = set()   # kernel instantiations launched by the per-layer tests (A)


def H():
    from yogo_amd import _hip

    return _hip


def bf(t):
    return t.to(torch.bfloat16).float()


def to8c(t):
    h = H()
    B, Cc, Hh, W = t.shape
    out = torch.empty(B, h.lib().yogo_bf16_channel_blocks(Cc), Hh, W, 8, dtype=torch.bfloat16, device="cuda")
    h.call("yogo_nchw_f32_to_bf16_8c", t.contiguous().float(), out, B, Cc, Hh * W, h.stream_ptr())
    return out


def from8c(t, Cc):
    h = H()
    B, cb, Hh, W, _ = t.shape
    out = torch.empty(B, Cc, Hh, W, device="cuda")
    h.call("yogo_bf16_8c_to_nchw_f32", t, out, B, Cc, Hh * W, h.stream_ptr())
    return out


def rel(a, b):
    """max |a - b| / max |b| on the device"""
    return float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30))


def kernels_of(lines):
    return {re.sub(r"\s+", "", ln.split("|")[0]) for ln in lines}


def sign_map_gpu(y, Cc):
    """include/yogo_hip.h, yogo_bf16_signs_bytes: [B][2][H][W][Cpad/16] bytes; byte (h, pixel, q), bit i + 4e = (channel 4h + i of
    channel block 2q + e > 0).  y: fp32 NCHW on the device."""
    B, _, Hh, W = y.shape
    cpad = 32 if Cc <= 32 else (64 if Cc <= 64 else (Cc + 127) // 128 * 128)
    pos = torch.zeros(B, cpad, Hh, W, dtype=torch.int32, device="cuda")
    pos[:, :Cc] = (y > 0)
    # channel = 16 q + 8 e + 4 h + i
    pos = pos.view(B, cpad // 16, 2, 2, 4, Hh, W).permute(0, 3, 5, 6, 1, 2, 4)   # [B][h][H][W][q][e][i]
    wts = torch.tensor([[1, 2, 4, 8], [16, 32, 64, 128]], dtype=torch.int32, device="cuda")
    return (pos * wts).sum((-1, -2)).to(torch.uint8).reshape(-1)


@pytest.mark.parametrize("layer", sorted(LAYERS))
def test_layer_every_direction_at_772x1032(layer):
    h = H()
    cin, cout, k, s, IH, IW, has_bias, fvar, dvar = LAYERS[layer]
    B = 2
    pad = 1 if k == 3 else 0
    g = torch.Generator(device="cuda").manual_seed(100 + layer)
    xd = bf(torch.randn(B, cin, IH, IW, generator=g, device="cuda"))          # device copies feed the kernels,
    w = torch.randn(cout, cin, k, k, generator=g, device="cuda") / (cin * k * k) ** 0.5
    bdev = torch.randn(cout, generator=g, device="cuda") if has_bias else None
    x = xd.cpu().requires_grad_(True)                                         # CPU copies feed the fp32 reference
    wb = bf(w).cpu().requires_grad_(True)
    b = bdev.cpu().requires_grad_(True) if has_bias else None
    ref_pre = F.conv2d(x, wb, b, stride=s, padding=pad)
    OH, OW = ref_pre.shape[2:]
    st = h.stream_ptr()
    # (blocks 2..4 of the reference carry Dropout2d: a channel mask; block 7 -- layer 6 here -- is LeakyReLU without one)
    mask = ((torch.rand(B, cout, generator=g, device="cuda") > 0.1).float() / 0.9) if (fvar == "signs" and layer <= 3) else None
    x8 = to8c(xd)
    h.launch_log(True)
    # ---- forward ---------------------------------------------------------------------------------------------------------
    packed = torch.empty(h.query_size("yogo_conv_bf16_packed_bytes", cin, cout, k, 0), dtype=torch.uint8, device="cuda")
    h.call("yogo_conv_bf16_pack", w, None, packed, cin, cout, k, 0, st)
    bd = bdev
    if fvar == "head":
        o32 = torch.full((B, cout, OH, OW), float("nan"), device="cuda")
        h.call("yogo_conv2d_fwd_bf16", x8, packed, bd, None, o32, None, None, B, cin, cout, IH, IW, k, s, 0, st)
        assert rel(o32.cpu(), ref_pre.detach()) < 2e-5 * max(1.0, (cin * k * k) ** 0.5 / 8), layer
    else:
        out = torch.full((B, h.lib().yogo_bf16_channel_blocks(cout), OH, OW, 8), float("nan"), dtype=torch.bfloat16, device="cuda")
        if fvar == "signs":
            sg = torch.full((h.query_size("yogo_bf16_signs_bytes", B, cout, OH, OW),), 0xA5, dtype=torch.uint8, device="cuda")
            h.call("yogo_conv2d_fwd_bf16_signs", x8, packed, bd, out, sg, mask, B, cin, cout, IH, IW, k, s, 1, st)
            want = F.leaky_relu(ref_pre.detach(), 0.01) * (mask.cpu()[:, :, None, None] if mask is not None else 1.0)
            got = from8c(out, cout)
            assert rel(got.cpu(), want) < 8e-3, layer
            # the sign map is the map of the STORED output (bit-exact), channel blocks that exist only
            cpad = 32 if cout <= 32 else (64 if cout <= 64 else (cout + 127) // 128 * 128)
            used = (cout + 15) // 16
            assert torch.equal(sg.view(-1, cpad // 16)[:, :used], sign_map_gpu(got, cout).view(-1, cpad // 16)[:, :used]), layer
        else:
            h.call("yogo_conv2d_fwd_bf16", x8, packed, bd, out, None, None, None, B, cin, cout, IH, IW, k, s, 0, st)
            assert rel(from8c(out, cout).cpu(), ref_pre.detach()) < 8e-3, layer
    # ---- backward ----------------------------------------------------------------------------------------------------------
    gy = bf(torch.randn(B, cout, OH, OW, generator=g, device="cuda"))
    ref_pre.backward(gy.cpu())
    gy8 = to8c(gy)
    dmode = 2 if (s == 2 and k == 3) else 1
    pd = torch.empty(h.query_size("yogo_conv_bf16_packed_bytes", cin, cout, k, dmode), dtype=torch.uint8, device="cuda")
    h.call("yogo_conv_bf16_pack", w, None, pd, cin, cout, k, dmode, st)
    dx = torch.full((B, h.lib().yogo_bf16_channel_blocks(cin), IH, IW, 8), float("nan"), dtype=torch.bfloat16, device="cuda")
    if dvar == "signs":
        refy = torch.randn(B, cin, IH, IW, generator=g, device="cuda")
        cmask = (torch.rand(B, cin, generator=g, device="cuda") > 0.1).float() / 0.9
        h.call("yogo_conv2d_dgrad_bf16_signs", gy8, pd, dx, sign_map_gpu(refy, cin), cmask, B, cin, cout, IH, IW, k, s, st)
        want = x.grad * torch.where(refy.cpu() > 0, 1.0, 0.01) * cmask.cpu()[:, :, None, None]
    else:
        h.call("yogo_conv2d_dgrad_bf16", gy8, pd, dx, None, 0, None, B, cin, cout, IH, IW, k, s, st)
        want = x.grad
    assert rel(from8c(dx, cin).cpu(), want) < 8e-3, layer
    # ---- weight gradient: B = 2 and the production batch 128 (split-K plan) -----------------------------------------------------
    for Bw in (B, 128):
        if Bw == B:
            xw, gw, wg, bg = x8, gy8, wb.grad, (b.grad if b is not None else None)
        else:
            del ref_pre, want
            xs = bf(torch.randn(Bw, cin, IH, IW, generator=g, device="cuda"))
            gs = bf(torch.randn(Bw, cout, OH, OW, generator=g, device="cuda"))
            xw, gw = to8c(xs), to8c(gs)
            xs, gs = xs.cpu(), gs.cpu()
            wg = torch.zeros(w.shape, dtype=torch.float64)
            for i0 in range(0, Bw, 16):   # chunked fp32 reference on the CPU, summed in fp64
                wg += torch.nn.grad.conv2d_weight(xs[i0:i0 + 16], w.shape, gs[i0:i0 + 16], stride=s, padding=pad).double()
            wg = wg.float()
            bg = gs.double().sum((0, 2, 3)).float() if b is not None else None
            del xs, gs
        ws = torch.empty(h.query_size("yogo_conv2d_wgrad_bf16_workspace_bytes", Bw, cin, cout, IH, IW, k, s) // 4, device="cuda")
        dw = torch.full((cout, cin, k, k), float("nan"), device="cuda")
        db = torch.full((cout,), float("nan"), device="cuda") if b is not None else None
        h.call("yogo_conv2d_wgrad_bf16", xw, gw, dw, db, ws, Bw, cin, cout, IH, IW, k, s, 0.0, st)
        assert float((dw.cpu() - wg).abs().max()) < 1e-4 * float(wg.abs().max()), (layer, Bw)
        if b is not None:
            assert float((db.cpu() - bg).abs().max()) < 1e-4 * float(bg.abs().max()), (layer, Bw)
    h.launch_log(False)
    SEEN.update(kernels_of(h.read_launch_log()))


def _model(B, seed=0):
    from yogo_amd.model import YOGO

    torch.manual_seed(seed)
    m = YOGO((HI, WI), 0.0425, 0.0555, C, clip_value=1e9).cuda()   # unclamped: the raw gradients are compared
    m.train()
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout2d):
            mod.p = 0.0
    return m


def _oracle_step(sd, x, lab, device):
    """the oracle's bf16-storage emulation of the step (rounds where the HIP path stores bf16): loss, gradients, running stats"""
    spec = O.arch("base_model", C)
    loss, _, grads, ns = O.bf16_train_step(x, sd, spec, lab, 0.0425, 0.0555)
    return loss, grads, ns


def _compare_step(tr, model, loss_ref, grads_ref, stats_ref, what):
    from _util import BF16_STEP_LOSS_RTOL, assert_grads_match_bf16_oracle

    got = tr.loss_components()
    assert abs(got["loss"] - loss_ref) < BF16_STEP_LOSS_RTOL * abs(loss_ref), (what, got, loss_ref)
    off = 0
    flat = tr.flat.grad
    mine = {}
    for name, p in model.named_parameters():
        n = p.numel()
        mine[name] = flat[off:off + n].view(p.shape).cpu()
        off += n
    worst = assert_grads_match_bf16_oracle(mine, grads_ref, what)
    sd = model.state_dict()
    for k, v in stats_ref.items():
        if "num_batches" in k:
            assert int(sd[k]) == int(v)
        else:
            torch.testing.assert_close(sd[k].cpu(), v.cpu(), rtol=1e-3, atol=1e-4)
    return worst


def test_bf16_training_step_at_772x1032_vs_cpu_oracle():
    """(B): one bf16 step at the production image size, B = 2, against the oracle's bf16-storage emulation of the same step on
    the CPU (O.bf16_train_step): end to end loss 1e-3, every gradient tensor cosine >= 0.995, running statistics 1e-3; then
    TEACHER-FORCED (tests/_util.py): every stored tensor, statistic and parameter gradient of the step against the per-block
    emulation fed with the step's own tensors -- one bf16 ulp / 2e-4 of max|g|"""
    from yogo_amd.train import HipTrainer
    from yogo_amd.yogo_loss import YOGOLoss

    B = 2
    m = _model(B, seed=21)
    sd0 = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    x = O.synthetic_images(B, HI, WI, seed=22)
    lab = O.synthetic_labels(B, m.Sx, m.Sy, K=64, num_classes=C, seed=23)
    tr = HipTrainer(m, YOGOLoss().cuda(), total_steps=10, half=True)
    tr.trace = {}
    tr.step(x.cuda(), lab.cuda())
    torch.cuda.synchronize()
    from _util import teacher_forced_bf16_step_check

    teacher_forced_bf16_step_check(O, tr, m, x, lab, O.arch("base_model", C), sd0, "772x1032 B=2")
    loss_ref, grads_ref, ns = _oracle_step(sd0, x, lab, "cpu")
    _compare_step(tr, m, loss_ref, grads_ref, ns, "B=2 vs bf16-emulating CPU oracle")


def test_production_batch_step_and_kernel_set():
    """(C): bench.py's step (B = 128, bf16) on 64 copies of (B)'s two images: the launch log must contain every
    conv_bf16_kernel / wgrad_bf16_kernel instantiation of the committed rocprofv3 summary and -- when the per-layer tests (A)
    ran in this session -- nothing the per-layer tests did not launch too; loss, gradients and batch statistics must
    reproduce the oracle's bf16-storage emulation of the step on the two images (same batch statistics, mean-reduced loss)"""
    from yogo_amd.train import HipTrainer
    from yogo_amd.yogo_loss import YOGOLoss

    h = H()
    B, rep = 128, 64
    m = _model(B, seed=21)
    sd0 = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    x2 = O.synthetic_images(2, HI, WI, seed=22)
    lab2 = O.synthetic_labels(2, m.Sx, m.Sy, K=64, num_classes=C, seed=23)
    x = x2.cuda().repeat(rep, 1, 1, 1)
    lab = lab2.cuda().repeat(rep, 1, 1, 1)
    tr = HipTrainer(m, YOGOLoss().cuda(), total_steps=10, half=True)
    h.launch_log(True)
    tr.step(x, lab)
    torch.cuda.synchronize()
    h.launch_log(False)
    lines = h.read_launch_log()
    # the parity step above runs with Dropout2d off (torch's mask stream cannot be reproduced); the bench's step has it on, which
    # selects the channel-mask epilogues: one more step with the reference's rates, for the kernel set only (on a copy: the parity
    # step's gradients and statistics are compared below)
    import copy

    m2 = copy.deepcopy(m)
    for mod, pr in zip([mm for mm in m2.modules() if isinstance(mm, torch.nn.Dropout2d)], (0.05, 0.1, 0.15)):
        mod.p = pr
    tr2 = HipTrainer(m2, YOGOLoss().cuda(), total_steps=10, half=True)
    h.launch_log(True)
    tr2.step(x, lab)
    torch.cuda.synchronize()
    h.launch_log(False)
    lines = lines + h.read_launch_log()
    del tr2, m2
    launched = kernels_of(lines)
    print("\n".join(sorted(set(lines))))
    profs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_kernel_stats.txt")))
    assert profs, "no committed rocprofv3 summary"
    want = set()
    for ln in open(profs[-1]):
        mm = re.search(r"((?:conv_bf16_kernel|conv_bf16_ws\d?_kernel|conv_bf16_s2d_direct_kernel|conv_bf16_staged_kernel|conv_bf16_1x1_f32_kernel|wgrad_bf16_kernel)<[^>]*>)", ln)
        if mm:
            want.add(re.sub(r"\s+", "", mm.group(1)))
    assert want, profs[-1]
    missing = want - launched
    assert not missing, f"instantiations of {os.path.basename(profs[-1])} that this step did not launch: {sorted(missing)}"
    if SEEN:
        conv = {k for k in launched if k.startswith(("conv_bf16_kernel", "conv_bf16_ws_kernel", "conv_bf16_ws3_kernel", "conv_bf16_s2d_direct_kernel", "conv_bf16_staged_kernel", "conv_bf16_1x1_f32_kernel", "wgrad_bf16_kernel"))}
        assert conv <= SEEN, f"launched at B=128 but not covered by the per-layer parity tests: {sorted(conv - SEEN)}"
    loss_ref, grads_ref, ns = _oracle_step(sd0, x2, lab2, "cpu")
    _compare_step(tr, m, loss_ref, grads_ref, ns, "B=128 (64 x 2 images) vs bf16-emulating CPU oracle on the 2 images")


# ---- BASELINE configs[1]: fp32 forward + loss at 772x1032, batch 64 ------------------------------------------------------------------
def _fp32_kernels(lines):
    out = set()
    for ln in lines:
        m = re.match(r"\s*((?:conv_igemm_f32_kernel|conv_first_kernel)[^|]*)\|", ln)
        if m:
            out.add(re.sub(r"\s+", "", m.group(1)))
    return out


def test_fp32_train_forward_and_loss_at_772x1032():
    """configs[1] (what bench.py::fp32_forward_loss times): the fp32 TRAIN-mode forward (BatchNorm batch statistics, decode) + fused
    loss kernel at the production image size.  B = 2 against the oracle's fp32 forward + loss on the CPU (yogo/model.py:267-313,
    yogo/yogo_loss.py:38-129; prediction rtol 1e-4 / atol 1e-4 of its range, loss + components 1e-4, running statistics 2e-4);
    then batch 64 = 32 copies of the two images: same batch statistics, so the same prediction per copy and the same mean loss --
    with the launch log showing the conv_igemm_f32_kernel<...> instantiations of the batch-64 run (incl. <4, 2, .>, the kernel of
    the bench line's fp32 roofline): the kernels bench.py measures are the kernels held to the oracle here"""
    from yogo_amd.model import YOGO
    from yogo_amd.yogo_loss import YOGOLoss

    h = H()
    torch.manual_seed(31)
    m = YOGO((HI, WI), 0.0425, 0.0555, C).cuda()
    m.train()
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout2d):
            mod.p = 0.0
    sd0 = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    x2 = O.synthetic_images(2, HI, WI, seed=32)
    lab2 = O.synthetic_labels(2, m.Sx, m.Sy, K=64, num_classes=C, seed=33)
    L = YOGOLoss().cuda()
    spec = O.arch("base_model", C)
    ns = {}
    with torch.no_grad():
        want = O.yogo_forward(x2, sd0, spec, 0.0425, 0.0555, train=True, new_stats=ns)
        wloss, wcomps = O.yogo_loss(want, lab2)
    seen = {}
    for B in (2, 64):
        m.load_state_dict(sd0)
        rep = B // 2
        x, lab = x2.cuda().repeat(rep, 1, 1, 1), lab2.cuda().repeat(rep, 1, 1, 1)
        h.launch_log(True)
        with torch.no_grad():
            out = m(x)
            loss, comps = L(out, lab)
        torch.cuda.synchronize()
        h.launch_log(False)
        seen[B] = _fp32_kernels(h.read_launch_log())
        scale = float(want.abs().max())
        for r in (0, rep - 1):   # first and last copy
            d = float((out[2 * r:2 * r + 2].cpu() - want).abs().max())
            assert d < 1e-4 * scale + 1e-4, (B, r, d, scale)
        assert abs(float(loss) - float(wloss)) < 1e-4 * abs(float(wloss)), (B, float(loss), float(wloss))
        for k in wcomps:
            assert abs(comps[k] - wcomps[k]) < 1e-4 * abs(wcomps[k]) + 1e-6, (B, k, comps[k], wcomps[k])
        sd = m.state_dict()
        for k, v in ns.items():
            if "num_batches" in k:
                assert int(sd[k]) == int(v)
            elif "running_mean" in k or B == 2:
                torch.testing.assert_close(sd[k].cpu(), v, rtol=2e-4, atol=2e-4)
            # (running_var at B = 64: the unbiased n / (n - 1) factor differs by < 1e-6 at these element counts -- same bound)
            else:
                torch.testing.assert_close(sd[k].cpu(), v, rtol=2e-4, atol=2e-4)
    assert any(k.startswith("conv_igemm_f32_kernel<4,2") for k in seen[64]), seen[64]   # the kernel of bench.py's fp32 roofline line
    print("fp32 kernels at B=64:", sorted(seen[64]))
