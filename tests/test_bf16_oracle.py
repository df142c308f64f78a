"""The oracle's bf16-storage emulation of the training step (O.bf16_train_step) is itself pinned here, on the CPU:
with the rounding switched off it must reproduce the step of the SAME oracle functions (run in float64, so that the comparison is free of fp32 summation noise) that the reference fixtures pin
(yogo_forward + yogo_loss + autograd: tests/test_oracle_golden.py) -- i.e. its hand-written backward chain (BatchNorm
backward, the fused layer-0 sums incl. the Gram form, activation derivatives taken from stored outputs / pre-activations,
dropout masks) is the derivative of the reference's forward (yogo/model_defns.py:30-77, yogo/yogo_loss.py:38-129,
yogo/train.py:309-322).  With rounding on it must stay a bf16-sized perturbation of that step."""
import pytest
import torch

import yogo_oracle as O


def _fp32_step(x, sd, spec, lab, drop_masks=None, dtype=torch.float32):
    """the pinned oracle functions + autograd; dtype float64 gives the derivative free of fp32 summation noise (torch's fp32
    BatchNorm backward on the CPU is ~1e-3 noisy below a BatchNorm layer: it subtracts batch means of the gradient)"""
    names = [k for k, v in sd.items() if k.startswith("model.") and v.is_floating_point() and "running" not in k]
    sdd = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()}
    leaf = {k: sdd[k].clone().requires_grad_(True) for k in names}
    sdl = dict(sdd)
    sdl.update(leaf)
    ns = {}
    dm = None if drop_masks is None else {i: v.to(dtype) for i, v in drop_masks.items()}
    pred = O.yogo_forward(x.to(dtype), sdl, spec, 0.0425, 0.0555, train=True, drop_masks=dm, new_stats=ns)
    loss, comps = O.yogo_loss(pred, lab.to(dtype))
    loss.backward()
    return float(loss.detach()), comps, {k: v.grad.float() for k, v in leaf.items()}, ns


CASES = [("base_model", 64, 96, False, 0), ("base_model", 49, 67, False, 0), ("silu_model", 64, 96, False, 0),
         ("quarter_filters", 66, 50, True, 0), ("depth_ver_3", 64, 96, False, 0), ("triple_filters", 48, 64, False, 0),
         ("base_model", 64, 96, False, 1)]


@pytest.mark.parametrize("name,H,W,rgb,with_drop", CASES)
def test_unrounded_emulation_is_the_fp32_step(monkeypatch, name, H, W, rgb, with_drop):
    monkeypatch.setattr(O, "_rb", lambda t: t)
    C = 5
    spec = O.arch(name, C)
    sd = O.init_state(spec, 3 if rgb else 1, seed=11)
    g = torch.Generator().manual_seed(5)
    for k in list(sd):   # non-trivial BatchNorm affine parameters and biases
        if k.endswith(".1.weight"):
            sd[k] = 1 + 0.3 * torch.randn(sd[k].shape, generator=g)
        elif k.endswith("bias") and sd[k].is_floating_point():
            sd[k] = 0.1 * torch.randn(sd[k].shape, generator=g)
    x = torch.randint(0, 256, (3, 3 if rgb else 1, H, W), dtype=torch.uint8, generator=g)
    Sx, Sy = O.grid_size(spec, H, W)
    lab = O.synthetic_labels(3, Sx, Sy, K=5, num_classes=C, seed=6)
    masks = None
    if with_drop:
        masks = {i: (torch.rand(3, e[0], generator=g) >= e[6]).float() / (1 - e[6]) for i, e in enumerate(spec) if e[6] > 0}
    l0, c0, g0, ns0 = _fp32_step(x, sd, spec, lab, masks, dtype=torch.float64)
    l1, c1, g1, ns1 = O.bf16_train_step(x, sd, spec, lab, 0.0425, 0.0555, drop_masks=masks)
    assert abs(l0 - l1) < 1e-5 * abs(l0)
    assert set(g0) == set(g1)
    for k in g0:
        gmax = float(g0[k].abs().max())
        if k.endswith(".0.bias") and k.replace(".0.bias", ".1.weight") in g0:   # conv bias in front of BatchNorm: zero + noise
            wmax = float(g0[k.replace(".bias", ".weight")].abs().max())
            assert float(g1[k].abs().max()) < 1e-3 * wmax and gmax < 1e-3 * wmax, k
            continue
        err = float((g0[k] - g1[k]).abs().max())
        assert err < 5e-5 * gmax + 1e-7, (name, k, err, gmax)   # fp32 convolutions of the emulation against a float64 derivative
    for k in ns0:
        torch.testing.assert_close(ns1[k].float(), ns0[k].float(), rtol=1e-4, atol=1e-5)


def test_rounded_emulation_is_a_bf16_sized_perturbation():
    C = 7
    spec = O.arch("base_model", C)
    sd = O.init_state(spec, 1, seed=3)
    x = O.synthetic_images(2, 96, 128, seed=1)
    Sx, Sy = O.grid_size(spec, 96, 128)
    lab = O.synthetic_labels(2, Sx, Sy, K=6, num_classes=C, seed=2)
    l0, _, g0, _ = _fp32_step(x, sd, spec, lab)
    l1, _, g1, _ = O.bf16_train_step(x, sd, spec, lab, 0.0425, 0.0555, clip=1.0)
    assert abs(l0 - l1) < 5e-3 * abs(l0)
    for k in g0:
        if k == "model.5.0.bias":
            continue
        a, b = g1[k].double().reshape(-1), g0[k].clamp(-1, 1).double().reshape(-1)
        cos = float((a * b).sum() / (a.norm() * b.norm() + 1e-30))
        assert 0.9 < cos < 1.0 - 1e-9, (k, cos)          # close to, but NOT, the fp32 step
        assert float(a.abs().max()) <= 1.0                # clamped (yogo/model.py:76-77)
    # determinism
    l2, _, g2, _ = O.bf16_train_step(x, sd, spec, lab, 0.0425, 0.0555, clip=1.0)
    assert l1 == l2 and all(torch.equal(g1[k], g2[k]) for k in g1)


# ---- the teacher-forced checker itself (tests/_util.py), on the CPU: fed with a mock "step" assembled from the emulation's own
# ---- tensors it must pass with zero differences; with ONE stored value moved by two bf16 ulps, or one gradient element
# ---- moved by 1e-3 of its tensor's range, it must fail.  (On the GPU box the same function checks the real HipTrainer step.)
def _to8c(t):
    B, C, H, W = t.shape
    cp = (C + 15) // 16 * 16
    p = torch.zeros(B, cp, H, W)
    p[:, :C] = t
    return p.view(B, cp // 8, 8, H, W).permute(0, 1, 3, 4, 2).contiguous().to(torch.bfloat16)


class _Obj:
    def __init__(self, **kw):
        self.__dict__.update(kw)


def _mock_step(name, H, W, C=5, B=2):
    spec = O.arch(name, C)
    sd = O.init_state(spec, 1, seed=4)
    x = O.synthetic_images(B, H, W, seed=8)
    Sx, Sy = O.grid_size(spec, H, W)
    lab = O.synthetic_labels(B, Sx, Sy, K=5, num_classes=C, seed=9)
    taps = {}
    loss, comps, grads, _ = O.bf16_train_step(x, sd, spec, lab, 0.0425, 0.0555, taps=taps)
    n = len(spec)
    saved = []
    for i, S in enumerate(taps["saved"]):
        last = i == n - 1
        no_z = i == 0 and O.l0_keeps_no_z(spec, O.l0_on_matrix_cores(spec, x))   # (the engine then keeps the sign map instead of z)
        saved.append(_Obj(x_in=x if i == 0 else _to8c(S["x"]), y=S["y"] if last else _to8c(S["y"]),
                          z=_to8c(S["z"]) if S["bn"] and not no_z else None, signs0=O.l0_sign_map(S) if no_z else None,
                          mean=S.get("mean"), invstd=S.get("invstd"), mask=None,
                          pre=_to8c(S["pre"]) if S.get("pre") is not None else None))
    trace = {"saved": saved, "raw": taps[f"y{n - 1}"]}
    for i in range(n):
        trace[("g", i)] = _to8c(taps[f"g{i}"])
        if f"dz{i}" in taps:
            trace[("dz", i)] = _to8c(taps[f"dz{i}"])
    order = [k for k in sd if k in grads]   # state-dict order == named_parameters order
    params = [(k, _Obj(numel=(lambda t: (lambda: t.numel()))(grads[k]), shape=grads[k].shape)) for k in order]
    flat = torch.cat([grads[k].reshape(-1) for k in order])
    model = _Obj(_clip=0.0, anchor_w=0.0425, anchor_h=0.0555, named_parameters=lambda: params)
    tr = _Obj(trace=trace, loss=_Obj(no_obj_weight=0.5, iou_weight=5.0, classify_weight=1.0, label_smoothing=0.01),
              loss_components=lambda: {"loss": loss}, flat=_Obj(grad=flat))
    return tr, model, x, lab, spec, sd


@pytest.mark.parametrize("name,H,W", [("base_model", 64, 96), ("silu_model", 48, 64), ("depth_ver_3", 49, 67)])
def test_teacher_forced_checker_accepts_the_emulation(name, H, W):
    from _util import teacher_forced_bf16_step_check

    tr, model, x, lab, spec, sd = _mock_step(name, H, W)
    teacher_forced_bf16_step_check(O, tr, model, x, lab, spec, sd, name)


@pytest.mark.parametrize("what", ["y3", "dz4", "g2", "grad", "mean"])
def test_teacher_forced_checker_rejects_a_wrong_value(what):
    from _util import teacher_forced_bf16_step_check

    tr, model, x, lab, spec, sd = _mock_step("base_model", 64, 96)
    tc = tr.trace

    def bump(t):   # one element, two bf16 ulps (a large one, so that it is no cancellation value)
        f = t.float()
        idx = int(f.abs().reshape(-1).argmax())
        v = f.reshape(-1)[idx]
        t.view(-1)[idx] = (v * (1 + 2.0 ** -6)).to(t.dtype)

    if what == "y3":
        bump(tc["saved"][3].y)
    elif what == "dz4":
        bump(tc[("dz", 4)])
    elif what == "g2":
        bump(tc[("g", 2)])
    elif what == "mean":
        tc["saved"][4].mean = tc["saved"][4].mean * (1 + 1e-3)
    else:
        g = tr.flat.grad
        g[1000] += 1e-3 * float(g.abs().max())
    with pytest.raises(AssertionError):
        teacher_forced_bf16_step_check(O, tr, model, x, lab, spec, sd, what)
