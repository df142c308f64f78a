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
