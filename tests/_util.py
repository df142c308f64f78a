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


# ---- a bf16 training step of the HIP path against the oracle's bf16-storage emulation (O.bf16_train_step) ----------------
# Tolerances (stated once, used by every bf16 whole-step test): the emulation rounds where the HIP path stores bf16, so what
# is left between the two is fp32 summation order plus the rare bf16 value that sits on a rounding boundary:
#   loss 1e-3 relative; per gradient tensor max|d| <= 1e-2 max|g| and cosine >= 0.999; running statistics 1e-3.
BF16_STEP_LOSS_RTOL = 1e-3
BF16_STEP_GRAD_RTOL = 1e-2
BF16_STEP_COS_MIN = 0.999


def is_conv_bias_before_bn(name, names):
    """model.{i}.0.bias of a block that also holds a BatchNorm (model.{i}.1.weight): its gradient is mathematically zero"""
    parts = name.split(".")
    return name.endswith(".0.bias") and f"model.{parts[1]}.1.weight" in names


def assert_grads_match_bf16_oracle(got, want, what, grad_rtol=BF16_STEP_GRAD_RTOL, cos_min=BF16_STEP_COS_MIN, verbose=True):
    """got / want: dict name -> gradient tensor (CPU).  Returns (worst cosine, its name)."""
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
        assert err <= grad_rtol * gmax + 1e-9, (what, name, err, gmax)
        assert cos >= cos_min, (what, name, cos)
    return worst
