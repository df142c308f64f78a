"""In-process A/B of the persistent wavefront-specialised kernel against the tiled 8-wavefront kernel (yogo_hook_conv_bf16_persistent 1 / 0 in libyogo_hip_hooks.so -- the product library has no plan switch):
the launches of the training step the persistent kernel takes, alternating, one device, one process.
    python tools/ab_ws.py [rounds] [B] [which]"""
import collections
import contextlib
import io
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from yogo_amd import _hip as H
H.LIB_PATH = os.path.join(ROOT, "yogo_amd", "lib", "libyogo_hip_hooks.so")   # the product's objects + the yogo_hook_* switches
import ctypes
for _n in ("yogo_hook_conv_bf16_persistent", "yogo_hook_conv_bf16_direct", "yogo_hook_conv_bf16_staged", "yogo_hook_conv_bf16_head"):
    _f = getattr(H.lib(), _n)
    _f.restype, _f.argtypes = ctypes.c_int, [ctypes.c_int]
import bench_conv_bf16 as BC

if __name__ == "__main__":
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    which = sys.argv[3].split(",") if len(sys.argv) > 3 else ["l3s", "l5f", "l5d", "l6f", "l6d"]
    # (kinds: l3s = layer 3 forward with bias + LeakyReLU + mask + sign map; l5a/l6a would take the mask; the step's layer 5 / 6
    #  forwards are bias only = kind "f" WITHOUT BatchNorm sums here is not expressible in bench_conv_bf16, so "d" (no bias) stands
    #  in for them: same kernel, same epilogue order)
    res = collections.defaultdict(list)
    for r in range(rounds + 1):
        for mode in (0, 1):
            H.call("yogo_hook_conv_bf16_persistent", mode)
            H.call("yogo_hook_conv_bf16_head", mode)     # (the 1x1 head kernels)
            H.call("yogo_hook_conv_bf16_staged", mode)   # (the independent-wavefront kernel of the thin layers)
            H.call("yogo_hook_conv_bf16_direct", mode)   # (the direct stride-2 data gradients)
            for w in which:
                kind = w[-1]
                buf = io.StringIO()
                with contextlib.redirect_stdout(buf):
                    BC.bench(w[:-1], B, *BC.LAYERS[w[:-1]], "d" if kind == "f" else kind, reps=10)
                m = re.search(r": ([\d.]+) ms", buf.getvalue())
                if r > 0 and m:
                    res[(w, mode)].append(float(m.group(1)))
    H.call("yogo_hook_conv_bf16_persistent", 1)
    print("kernel        tiled (ms)                         persistent (ms)")
    for w in which:
        a, b = res[(w, 0)], res[(w, 1)]
        print(f"{w:6s}  {'/'.join(f'{x:.3f}' for x in a):>32s}  {'/'.join(f'{x:.3f}' for x in b):>32s}   {sorted(b)[len(b)//2] / sorted(a)[len(a)//2] - 1:+.1%}")
