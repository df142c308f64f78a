"""In-process A/B of compile-time variants of the layer-0 training forward (conv_first_mfma2_kernel: y + sign map, no z) and backward
sweep (conv_first_bn_wgrad_pk2_kernel: image + gradient + sign map -> sums), base_model's first convolution at 772x1032, batch 128.
    python tools/ab_l0.py TAG1,TAG2,... [rounds]      (TAG "base" = the product library; bash yogo_amd/csrc/build.sh variant TAG FILE -D...)"""
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from yogo_amd import _hip as H
from ab_variants import load

if __name__ == "__main__":
    tags = sys.argv[1].split(",")
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    B, IH, IW, Cout = 128, 772, 1032, 16
    OH, OW = IH // 2, IW // 2
    libs = {t: load(t) for t in tags}
    x = torch.randint(0, 256, (B, 1, IH, IW), dtype=torch.uint8, device="cuda")
    w = (torch.randn(Cout, 1, 3, 3, device="cuda") * 0.02).to(torch.bfloat16).float()
    bias = torch.randn(Cout, device="cuda")
    y = torch.empty(B, 2, OH, OW, 8, dtype=torch.bfloat16, device="cuda")
    sg = torch.empty(B * OH * OW * 2, dtype=torch.uint8, device="cuda")
    mean, invstd, gamma, beta = (torch.randn(Cout, device="cuda") for _ in range(4))
    g = torch.randn(B, 2, OH, OW, 8, device="cuda").to(torch.bfloat16)
    cols = H.query_ints("yogo_conv_first_bn_wgrad_cols", 1, 1, Cout)[0]
    rows = H.query_ints("yogo_conv_first_wgrad_rows", 1, B, IH, IW, 2)[0]
    part = torch.empty(rows * cols, device="cuda")
    res = collections.defaultdict(list)
    for r in range(rounds + 1):
        for t in tags:
            H._lib = libs[t]
            st = H.stream_ptr()
            fs = {"fwd": lambda: H.call("yogo_conv_first_mfma_signs", x, w, bias, None, y, sg, mean, invstd, gamma, beta, B, Cout, IH, IW, 1, st),
                  "bwd": lambda: H.call("yogo_conv_first_bn_wgrad_bf16_xs", x, 0, g, sg, mean, invstd, gamma, beta, part, B, 1, Cout, IH, IW, 2, 1, st)}
            for name, f in fs.items():
                f(); f()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    f()
                e1.record()
                torch.cuda.synchronize()
                if r > 0:
                    res[(t, name)].append(e0.elapsed_time(e1) / 10)
    for name in ("fwd", "bwd"):
        for t in tags:
            print(f"{name} {t:8s} " + "/".join(f"{v:.3f}" for v in res[(t, name)]) + " ms")
