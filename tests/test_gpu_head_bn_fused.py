"""BatchNorm backward of the block under the 1x1 detection head with the head's data gradient computed inside (yogo_bn_bwd_bf16_head, bn.hip)
against (a) the unfused pair yogo_conv2d_dgrad_bf16 + yogo_bn_bwd_bf16 and (b) fp64, and the training step with and without it.
Reference: autograd of yogo/model_defns.py:58-65 + the head (yogo/model.py:150-155)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _to8c(t):
    B, C, H, W = t.shape
    Cp = ((C + 15) // 16) * 16
    tp = torch.zeros(B, Cp, H, W, dtype=t.dtype)
    tp[:, :C] = t
    return tp.reshape(B, Cp // 8, 8, H, W).permute(0, 1, 3, 4, 2).contiguous().to(torch.bfloat16)


def _from8c(t, C):
    B, Cb, H, W, _ = t.shape
    return t.float().permute(0, 1, 4, 2, 3).reshape(B, Cb * 8, H, W)[:, :C]


@pytest.mark.parametrize("B,C,P,H,W,act,training", [(3, 128, 12, 13, 17, 1, 1), (2, 32, 7, 9, 33, 1, 1), (2, 64, 16, 24, 32, 0, 1), (2, 128, 12, 97, 129, 1, 0)])
def test_head_bn_backward_against_the_unfused_pair_and_fp64(B, C, P, H, W, act, training):
    from yogo_amd import _hip as h

    gen = torch.Generator().manual_seed(100 * H + W)
    HW = H * W
    gh = (torch.randn(B, P, H, W, generator=gen) * 0.3).to(torch.bfloat16).float()
    w = torch.randn(P, C, generator=gen) * 0.2
    z = torch.randn(B, C, H, W, generator=gen).to(torch.bfloat16).float()
    mean, var = z.mean((0, 2, 3)), z.var((0, 2, 3), unbiased=False)
    invstd = 1.0 / torch.sqrt(var + 1e-5)
    gamma, beta = torch.rand(C, generator=gen) + 0.5, torch.randn(C, generator=gen) * 0.3
    st = h.stream_ptr()
    gh8, z8 = _to8c(gh).cuda(), _to8c(z).cuda()
    dev = lambda t: t.cuda()
    rows = h.query_ints("yogo_bn_bwd_bf16_rows", 1, B, HW)[0]

    def run(fused):
        part = torch.full((rows * C * 2,), float("nan"), dtype=torch.float32, device="cuda")
        sums = torch.empty(2 * C, dtype=torch.float32, device="cuda")
        dgam, dbet = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
        if fused:
            dz = torch.full((B, C // 8, H, W, 8), float("nan"), dtype=torch.bfloat16, device="cuda")
            h.call("yogo_bn_bwd_bf16_head", gh8, dev(w), P, z8, dz, dev(mean), dev(invstd), dev(gamma), dev(beta), act, dgam, dbet, part, sums, B, C, HW,
                   training, 0.0, st)
        else:
            pk = torch.empty(h.query_size("yogo_conv_bf16_packed_bytes", C, P, 1, 1), dtype=torch.uint8, device="cuda")
            h.call("yogo_conv_bf16_pack", dev(w.reshape(P, C, 1, 1).contiguous()), None, pk, C, P, 1, 1, st)
            dz = torch.full((B, C // 8, H, W, 8), float("nan"), dtype=torch.bfloat16, device="cuda")
            h.call("yogo_conv2d_dgrad_bf16", gh8, pk, dz, None, 0, None, B, C, P, H, W, 1, 1, st)
            h.call("yogo_bn_bwd_bf16", dz, z8, dz, dev(mean), dev(invstd), dev(gamma), dev(beta), act, dgam, dbet, part, sums, B, C, HW, training, 0.0, st)
        torch.cuda.synchronize()
        return _from8c(dz.cpu(), C).double(), dgam.cpu().double(), dbet.cpu().double()

    dzf, dgf, dbf = run(True)
    dzu, dgu, dbu = run(False)
    # fp64 of the same definition: g = bf16(sum_k gh[k] bf16(w[k][c])), then BatchNorm + activation backward
    wb = w.to(torch.bfloat16).double()
    g = torch.einsum("bkhw,kc->bchw", gh.double(), wb).float().to(torch.bfloat16).double()
    xh = (z.double() - mean.double()[None, :, None, None]) * invstd.double()[None, :, None, None]
    y = xh * gamma.double()[None, :, None, None] + beta.double()[None, :, None, None]
    ge = g * (torch.where(y > 0, 1.0, 0.01) if act == 1 else 1.0)
    S1, S2 = ge.sum((0, 2, 3)), (ge * xh).sum((0, 2, 3))
    n = B * HW
    mg, mgx = (S1 / n, S2 / n) if training else (torch.zeros(C, dtype=torch.float64),) * 2
    dz64 = (invstd * gamma).double()[None, :, None, None] * (ge - mg[None, :, None, None] - xh * mgx[None, :, None, None])
    tol = 2.0 ** -7   # one bf16 rounding of dz (2^-9) + the rare element whose g rounds the other way (2^-8 of g)
    for name, dz in (("fused", dzf), ("unfused", dzu)):
        err = float((dz - dz64).abs().max()) / float(dz64.abs().max())
        assert err < tol, (name, err)
    assert float((dgf - S2).abs().max()) < 2e-3 * float(S2.abs().max()) + 1e-3 and float((dbf - S1).abs().max()) < 2e-3 * float(S1.abs().max()) + 1e-3
    assert float((dgf - dgu).abs().max()) < 2e-3 * float(dgu.abs().max()) + 1e-3 and float((dbf - dbu).abs().max()) < 2e-3 * float(dbu.abs().max()) + 1e-3
    # the two paths round g identically except where the fp32 sums differ in their last bits: almost every element of dz is the same bf16
    same = float((dzf == dzu).double().mean())
    assert same > 0.99, same


def test_training_step_with_and_without_the_head_fusion():
    from yogo_amd import _hip as h
    from yogo_amd import engine as E
    from yogo_amd.model import YOGO
    from yogo_amd.model_defns import get_model_func
    from yogo_amd.train import HipTrainer
    from yogo_amd.yogo_loss import YOGOLoss
    import yogo_oracle as O

    # depth_ver_0 (yogo/model_defns.py) is the ModelDefn whose last block before the head has a BatchNorm; base_model's has none and keeps
    # the head's data-gradient launch
    for Himg, Wimg, B in ((96, 128, 4), (136, 72, 3)):
        x = O.synthetic_images(B, Himg, Wimg, seed=43).cuda()
        out = {}
        old = E._HEAD_BN_FUSE
        try:
            for fused in (False, True):
                E._HEAD_BN_FUSE = fused
                torch.manual_seed(6)
                model = YOGO((Himg, Wimg), 0.0425, 0.0555, 7, clip_value=1e9, model_func=get_model_func("depth_ver_0")).cuda()
                model.train()
                lab = O.synthetic_labels(B, model.Sx, model.Sy, K=4, num_classes=7, seed=44).cuda()
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
            E._HEAD_BN_FUSE = old
        g0, names, sizes, log0 = out[False]
        g1, log1 = out[True][0], out[True][3]
        # the head's data gradient launch (M = 128 outputs of a 1x1 "convolution" with K = 16) is gone
        l0 = [l for l in log0.split("\n") if "K=12 M=128" in l]
        l1 = [l for l in log1.split("\n") if "K=12 M=128" in l]
        assert len(l0) == 1 and len(l1) == 0, (l0, l1)
        off = 0
        for n, sz in zip(names, sizes):
            a, b_ = g1[off:off + sz].double(), g0[off:off + sz].double()
            off += sz
            cos = float((a * b_).sum() / (a.norm() * b_.norm() + 1e-30))
            assert cos > 0.9995, (n, cos)   # (a handful of elements of the head's data gradient round the other way: bf16 noise downstream)
