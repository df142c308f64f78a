"""Micro-benchmark of layer 1's data gradient + layer 0's backward sweep at the production shape, fused and unfused, across library variants:
    python tools/bench_first_fused.py [TAG1,TAG2,...] [rounds] [B]      (TAG "base" = the product library)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from yogo_amd import _hip as H


def load(tag):
    H._lib, H.LIB_PATH = None, os.path.join(ROOT, "yogo_amd", "lib", "libyogo_hip.so" if tag == "base" else f"libyogo_hip_{tag}.so")
    return H.lib()


def timed(fn, reps=10):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


if __name__ == "__main__":
    tags = (sys.argv[1] if len(sys.argv) > 1 else "base").split(",")
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 128
    Hh, Ww = 386, 516
    libs = {t: load(t) for t in tags}
    H._lib = libs[tags[0]]
    st = H.stream_ptr()
    g8 = (torch.randn(B, 4, Hh, Ww, 8, device="cuda") * 0.5).to(torch.bfloat16)
    w = torch.randn(32, 16, 3, 3, device="cuda") * 0.1
    img = torch.randint(0, 256, (B, 1, 2 * Hh, 2 * Ww), dtype=torch.uint8, device="cuda")
    signs = torch.randint(0, 256, (B, Hh * Ww * 2), dtype=torch.uint8, device="cuda")
    pk = torch.empty(H.query_size("yogo_conv_bf16_packed_bytes", 16, 32, 3, 1), dtype=torch.uint8, device="cuda")
    H.call("yogo_conv_bf16_pack", w, None, pk, 16, 32, 3, 1, st)
    cols = H.query_ints("yogo_conv_first_bn_wgrad_cols", 1, 1, 16)[0]
    dx = torch.empty(B, 2, Hh, Ww, 8, dtype=torch.bfloat16, device="cuda")
    x8 = torch.randn(B, 2, Hh, Ww, 8, device="cuda").to(torch.bfloat16)
    dw, db = torch.empty(32, 16, 3, 3, device="cuda"), torch.empty(32, device="cuda")
    ws_u = torch.empty(H.query_size("yogo_conv2d_wgrad_bf16_workspace_bytes", B, 16, 32, Hh, Ww, 3, 1) // 4, dtype=torch.float32, device="cuda")
    one = torch.ones(16, device="cuda")
    res = {}
    for r in range(rounds + 1):
        for t in tags:
            H._lib = libs[t]
            rows_f = H.query_ints("yogo_conv2d_dgrad_first_bwd_rows", 1, B, Hh, Ww, 0)[0]
            rows_w = H.query_ints("yogo_conv2d_dgrad_first_bwd_rows", 1, B, Hh, Ww, 1)[0]
            part_w = torch.empty(rows_w * cols, dtype=torch.float32, device="cuda")
            ws_w = torch.empty(H.query_size("yogo_conv2d_dgrad_wgrad_first_bwd_workspace_bytes", B, Hh, Ww) // 4, dtype=torch.float32, device="cuda")
            rows_u = H.query_ints("yogo_conv_first_wgrad_rows", 1, B, 2 * Hh, 2 * Ww, 2)[0]
            part_f = torch.empty(rows_f * cols, dtype=torch.float32, device="cuda")
            part_u = torch.empty(rows_u * cols, dtype=torch.float32, device="cuda")
            tf = timed(lambda: H.call("yogo_conv2d_dgrad_bf16_first_bwd", g8, pk, img, signs, part_f, B, 16, 32, Hh, Ww, 1, st))
            tfw = timed(lambda: H.call("yogo_conv2d_dgrad_wgrad_bf16_first_bwd", g8, pk, x8, img, signs, part_w, dw, db, ws_w, B, 16, 32, Hh, Ww, 1, 1.0, None, st))
            if t == tags[0]:
                tg = timed(lambda: H.call("yogo_conv2d_wgrad_bf16", x8, g8, dw, db, ws_u, B, 16, 32, Hh, Ww, 3, 1, 1.0, st))
                td = timed(lambda: H.call("yogo_conv2d_dgrad_bf16", g8, pk, dx, None, 0, None, B, 16, 32, Hh, Ww, 3, 1, st))
                tw = timed(lambda: H.call("yogo_conv_first_bn_wgrad_bf16_xs", img, 0, dx, signs, one, one, one, one, part_u, B, 1, 16, 2 * Hh, 2 * Ww, 2, 1, st))
            if r > 0:
                res.setdefault(t, []).append(tf)
                res.setdefault(t + " +wgrad", []).append(tfw)
                if t == tags[0]:
                    res.setdefault("unfused wgrad", []).append(tg)
                    res.setdefault("unfused dgrad", []).append(td)
                    res.setdefault("unfused sweep", []).append(tw)
    gb = B * (4 * Hh * Ww * 16 + 4 * Hh * Ww + 2 * Hh * Ww) / 1e9
    for k, v in res.items():
        med = sorted(v)[len(v) // 2]
        print(f"{k:16s} " + " ".join(f"{x:7.1f}" for x in v) + f"   median {med:7.1f} us" + (f"   {gb / med * 1e3:.2f} TB/s of algorithmic bytes" if k in libs else ""))
