"""micro-benchmark of single bf16 conv launches (experiments; not part of the product).
usage: bench_conv_bf16.py [B] [which]   which = comma list of layer names (l1..l7) x kind (f = fwd with BatchNorm sums, a = fwd with bias + leaky + channel mask,
s = as a plus the sign map, d = dgrad, r = dgrad with act'(ref) and channel mask, m = as r with the sign map as reference),
e.g. l5f,l4r"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yogo_amd import _hip as H

LAYERS = {"l1": (16, 32, 386, 516, 3, 1), "l2": (32, 64, 386, 516, 3, 2), "l3": (64, 128, 193, 258, 3, 1),
          "l4": (128, 128, 193, 258, 3, 2), "l5": (128, 128, 97, 129, 3, 1), "l6": (128, 128, 97, 129, 3, 1),
          "l7": (128, 12, 97, 129, 1, 1)}


def blocks(c):
    return ((c + 15) // 16) * 2


def bench(name, B, Cin, Cout, IH, IW, k, s, kind, reps=10):
    pad = 1 if k == 3 else 0
    OH, OW = (IH + 2 * pad - k) // s + 1, (IW + 2 * pad - k) // s + 1
    st = H.stream_ptr()
    w = torch.randn(Cout, Cin, k, k, device="cuda") * 0.05
    x8 = torch.randn(B, blocks(Cin), IH, IW, 8, device="cuda").to(torch.bfloat16)
    y8 = torch.randn(B, blocks(Cout), OH, OW, 8, device="cuda").to(torch.bfloat16)
    mode = 0 if kind in "fasph" else (2 if (s == 2 and k == 3) else 1)
    packed = torch.empty(H.query_size("yogo_conv_bf16_packed_bytes", Cin, Cout, k, mode), dtype=torch.uint8, device="cuda")
    H.call("yogo_conv_bf16_pack", w, None, packed, Cin, Cout, k, mode, st)
    if kind == "s":
        bias = torch.randn(Cout, device="cuda")
        msk = (torch.rand(B, Cout, device="cuda") > 0.1).float()
        sg = torch.empty(H.query_size("yogo_bf16_signs_bytes", B, Cout, OH, OW), dtype=torch.uint8, device="cuda")
        f = lambda: H.call("yogo_conv2d_fwd_bf16_signs", x8, packed, bias, y8, sg, msk, B, Cin, Cout, IH, IW, k, s, 1, st)
        nbytes = B * (16 * blocks(Cin) * IH * IW + 17 * blocks(Cout) * OH * OW)
    elif kind in "mn":   # n: sign map without the channel mask
        sg = torch.randint(0, 256, (H.query_size("yogo_bf16_signs_bytes", B, Cin, IH, IW),), dtype=torch.uint8, device="cuda")
        msk = (torch.rand(B, Cin, device="cuda") > 0.1).float() if kind == "m" else None
        f = lambda: H.call("yogo_conv2d_dgrad_bf16_signs", y8, packed, x8, sg, msk, B, Cin, Cout, IH, IW, k, s, st)
        nbytes = B * (17 * blocks(Cin) * IH * IW + 16 * blocks(Cout) * OH * OW)
    elif kind == "h":   # the head's forward: conv + bias into an fp32 NCHW tensor
        bias = torch.randn(Cout, device="cuda")
        o32 = torch.empty(B, Cout, OH, OW, device="cuda")
        f = lambda: H.call("yogo_conv2d_fwd_bf16", x8, packed, bias, None, o32, None, None, B, Cin, Cout, IH, IW, k, s, 0, st)
        nbytes = B * (16 * blocks(Cin) * IH * IW + 4 * Cout * OH * OW)
    elif kind == "p":   # plain forward: conv + bias, no activation, no BatchNorm sums (the training step's layers in front of a BatchNorm)
        bias = torch.randn(Cout, device="cuda")
        f = lambda: H.call("yogo_conv2d_fwd_bf16", x8, packed, bias, y8, None, None, None, B, Cin, Cout, IH, IW, k, s, 0, st)
        nbytes = B * 16 * (blocks(Cin) * IH * IW + blocks(Cout) * OH * OW)
    elif kind in "fa":
        rows, mpad = H.query_ints("yogo_conv2d_fwd_bf16_stats_shape", 2, B, Cin, Cout, IH, IW, k, s)
        stats = torch.empty(rows * mpad * 2, device="cuda")
        bias = torch.randn(Cout, device="cuda")
        msk = (torch.rand(B, Cout, device="cuda") > 0.1).float() if kind == "a" else None
        f = lambda: H.call("yogo_conv2d_fwd_bf16", x8, packed, bias, y8, None, msk, None if kind == "a" else stats, B, Cin, Cout, IH, IW,
                           k, s, 1 if kind == "a" else 0, st)
        nbytes = B * 16 * (blocks(Cin) * IH * IW + blocks(Cout) * OH * OW)
    else:
        ref = torch.randn(B, blocks(Cin), IH, IW, 8, device="cuda").to(torch.bfloat16) if kind == "r" else None
        msk = (torch.rand(B, Cin, device="cuda") > 0.1).float() if kind in "rc" else None   # c: channel mask only
        f = lambda: H.call("yogo_conv2d_dgrad_bf16", y8, packed, x8, ref, 1 if kind == "r" else 0, msk, B, Cin, Cout, IH, IW, k, s, st)
        nbytes = B * 16 * (blocks(Cin) * IH * IW * (2 if kind == "r" else 1) + blocks(Cout) * OH * OW)
    for _ in range(2):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    fl = 2.0 * B * Cout * Cin * k * k * OH * OW
    print(f"{name}{kind} B={B} {Cin}->{Cout} {IH}x{IW} k{k} s{s}: {ms:.3f} ms  {fl / ms / 1e9:.1f} TF  {nbytes / ms / 1e6:.0f} GB/s", flush=True)


if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    which = sys.argv[2].split(",") if len(sys.argv) > 2 else [n + k for n in LAYERS for k in "fd"]
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    for wname in which:
        bench(wname[:-1], B, *LAYERS[wname[:-1]], wname[-1], reps=reps)
