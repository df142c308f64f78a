"""In-process A/B of conv_bf16_ws16_kernel (v_mfma_f32_16x16x32_bf16) against conv_bf16_ws_kernel<0> (32x32x16) on the launches it takes
(yogo_hook_conv_bf16_ws16 1 / 0 in libyogo_hip_hooks.so -- the product library has no plan switch), alternating, one device, one process.
    python tools/ab_ws16.py [rounds] [B] [which] [reps]"""
import collections
import contextlib
import ctypes
import io
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from yogo_amd import _hip as H
H.LIB_PATH = os.path.join(ROOT, "yogo_amd", "lib", "libyogo_hip_hooks.so")
_f = H.lib().yogo_hook_conv_bf16_ws16
_f.restype, _f.argtypes = ctypes.c_int, [ctypes.c_int]
import bench_conv_bf16 as BC

if __name__ == "__main__":
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    which = sys.argv[3].split(",") if len(sys.argv) > 3 else ["l5p", "l5d"]   # p = conv + bias (layer 5's forward), d = data gradient (layers 5 / 6)
    reps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
    res = collections.defaultdict(list)
    for r in range(rounds + 1):   # round 0 warms the clocks and is dropped
        for mode in (0, 1):
            H.call("yogo_hook_conv_bf16_ws16", 2 * mode)   # (0 = the 32x32x16 kernel, 2 = the 16x16x32 member on every eligible launch)
            for w in which:
                buf = io.StringIO()
                with contextlib.redirect_stdout(buf):
                    BC.bench(w[:-1], B, *BC.LAYERS[w[:-1]], w[-1], reps=reps)
                m = re.search(r": ([\d.]+) ms", buf.getvalue())
                if r > 0 and m:
                    res[(w, mode)].append(float(m.group(1)))
    H.call("yogo_hook_conv_bf16_ws16", 1)
    print("kernel   32x32x16 ws<0> (ms)               16x16x32 ws16 (ms)")
    for w in which:
        a, b = res[(w, 0)], res[(w, 1)]
        print(f"{w:6s}  {'/'.join(f'{x:.3f}' for x in a):>32s}  {'/'.join(f'{x:.3f}' for x in b):>32s}   {sorted(b)[len(b)//2] / sorted(a)[len(a)//2] - 1:+.1%}")
