"""A/B of conv_bf16_kernel main loops in ONE process (diagnostic library; experiments only, not part of the product):
interleaved rounds of the ping-pong (PP) and the interleaved main loop on the production layer shapes.
    bash yogo_amd/csrc/build.sh diag && python tools/ab_conv_bf16.py [B] [which] [rounds]"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from yogo_amd import _hip as H

H.LIB_PATH = os.path.join(ROOT, "yogo_amd", "lib", "libyogo_hip_diag.so")
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_conv_bf16 as BC   # noqa: E402  (uses the same _hip module -> the diagnostic library)

if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    which = sys.argv[2].split(",") if len(sys.argv) > 2 else ["l5a", "l5d", "l3s", "l3m", "l4f"]
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    lib = H.lib()
    lib.yogo_diag_conv_bf16_pp.argtypes = [ctypes.c_int]
    for r in range(rounds):
        for pp in (1, 0):
            lib.yogo_diag_conv_bf16_pp(pp)
            print(f"--- round {r} pp={pp}", flush=True)
            for w in which:
                BC.bench(w[:-1], B, *BC.LAYERS[w[:-1]], w[-1], reps=10)
