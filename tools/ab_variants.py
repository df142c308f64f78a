"""In-process A/B of compile-time variants: several builds of the library (bash yogo_amd/csrc/build.sh variant TAG FILE -D...)
loaded side by side, the same micro-benchmarks alternating between them (rounds x variants x kernels, one device, one process).
    python tools/ab_variants.py conv TAG1,TAG2,... [which] [rounds] [B]      (TAG "base" = the product library)
    python tools/ab_variants.py wgrad TAG1,TAG2,... [which] [rounds] [B]"""
import collections
import ctypes
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from yogo_amd import _hip as H


def load(tag):
    path = os.path.join(ROOT, "yogo_amd", "lib", "libyogo_hip.so" if tag == "base" else f"libyogo_hip_{tag}.so")
    H._lib, H.LIB_PATH = None, path
    return H.lib()


if __name__ == "__main__":
    kind = sys.argv[1]
    tags = sys.argv[2].split(",")
    which = sys.argv[3].split(",") if len(sys.argv) > 3 and sys.argv[3] != "-" else None
    rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 3
    B = int(sys.argv[5]) if len(sys.argv) > 5 else 128
    libs = {t: load(t) for t in tags}
    if kind == "conv":
        import bench_conv_bf16 as BC
        which = which or ["l5a", "l5d", "l3s", "l3m", "l4f"]
        run = lambda w: BC.bench(w[:-1], B, *BC.LAYERS[w[:-1]], w[-1], reps=int(os.environ.get("AB_REPS", "10")))
    else:
        import ab_wgrad_bf16 as BW
        which = which or list(BW.LAYERS)
        run = lambda w: BW.bench(w, B, *BW.LAYERS[w])
    import io, contextlib
    res = collections.defaultdict(list)
    for r in range(rounds + 1):   # round 0 warms the clocks and is dropped
        for t in tags:
            H._lib = libs[t]
            for w in which:
                buf = io.StringIO()
                with contextlib.redirect_stdout(buf):
                    run(w)
                m = re.search(r": ([\d.]+) ms", buf.getvalue())
                if r > 0 and m:
                    res[(w, t)].append(float(m.group(1)))
    print("kernel  " + "  ".join(f"{t:>22s}" for t in tags))
    for w in which:
        print(f"{w:6s}  " + "  ".join(f"{'/'.join(f'{x:.3f}' for x in res[(w, t)]):>22s}" for t in tags))
    print("median  " + "  ".join(f"{sum(sorted(res[(w, t)])[len(res[(w, t)]) // 2] for w in which):22.3f}" for t in tags))
