"""numerical check of a wgrad variant library against the product library on the production layer shapes (same inputs, dw / db equal
up to fp32 summation order).  usage: wgrad_variant_check.py TAG"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
from yogo_amd import _hip as H
import ab_variants as AV
import ab_wgrad_bf16 as BW
tag = sys.argv[1]
libs = {t: AV.load(t) for t in ("base", tag)}
B = 4
for name, (Cin, Cout, IH, IW, k, s) in BW.LAYERS.items():
    pad = 1 if k == 3 else 0
    OH, OW = (IH + 2 * pad - k) // s + 1, (IW + 2 * pad - k) // s + 1
    g = torch.Generator(device="cuda").manual_seed(1)
    x8 = torch.randn(B, BW.blocks(Cin), IH, IW, 8, device="cuda", generator=g).to(torch.bfloat16)
    g8 = torch.randn(B, BW.blocks(Cout), OH, OW, 8, device="cuda", generator=g).to(torch.bfloat16)
    res = {}
    for t, L in libs.items():
        H._lib = L
        ws = torch.empty(H.query_size("yogo_conv2d_wgrad_bf16_workspace_bytes", B, Cin, Cout, IH, IW, k, s) // 4, device="cuda")
        dw, db = torch.full((Cout, Cin, k, k), float("nan"), device="cuda"), torch.full((Cout,), float("nan"), device="cuda")
        H.call("yogo_conv2d_wgrad_bf16", x8, g8, dw, db, ws, B, Cin, Cout, IH, IW, k, s, 0.0, H.stream_ptr())
        torch.cuda.synchronize()
        res[t] = (dw.cpu(), db.cpu())
    e1 = float((res["base"][0] - res[tag][0]).abs().max() / res["base"][0].abs().max())
    e2 = float((res["base"][1] - res[tag][1]).abs().max() / res["base"][1].abs().max())
    print(name, "dw rel", e1, "db rel", e2, flush=True)
