"""phase stamps of conv_bf16_kernel (diagnostic library; experiments only): mean s_memtime ticks per workgroup for prologue /
main loop / epilogue and, for the ping-pong loop, the sums of fetch / barrier / MFMA / barrier time of waves 0 and 4.
    bash yogo_amd/csrc/build.sh diag && python tools/stamps_conv_bf16.py [B] [which] [pp]"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from yogo_amd import _hip as H

H.LIB_PATH = os.path.join(ROOT, "yogo_amd", "lib", "libyogo_hip_diag.so")
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_conv_bf16 as BC   # noqa: E402

if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    which = sys.argv[2].split(",") if len(sys.argv) > 2 else ["l5a"]
    pps = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1, 0]
    dbg = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    lib = H.lib()
    lib.yogo_diag_conv_bf16_pp.argtypes = [ctypes.c_int]
    lib.yogo_diag_conv_bf16.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t]
    nmax = 1 << 16
    st = torch.zeros(nmax * 16, dtype=torch.int64, device="cuda")
    for w in which:
        for pp in pps:
            lib.yogo_diag_conv_bf16_pp(pp)
            lib.yogo_diag_conv_bf16(dbg, 0, None, 0)
            BC.bench(w[:-1], B, *BC.LAYERS[w[:-1]], w[-1], reps=5)          # warm clocks, un-stamped timing
            st.zero_()
            lib.yogo_diag_conv_bf16(dbg, 0, st.data_ptr(), st.numel() * 8)
            BC.bench(w[:-1], B, *BC.LAYERS[w[:-1]], w[-1], reps=1)
            torch.cuda.synchronize()
            h = st.view(nmax, 16).cpu().double()
            h = h[h[:, 3] != 0]
            if h.numel() == 0:
                print("  (no stamps)")
                continue
            pro, loop, epi = (h[:, 1] - h[:, 0]).mean(), (h[:, 2] - h[:, 1]).mean(), (h[:, 3] - h[:, 2]).mean()
            print(f"  {w} pp={pp}: wgs={h.shape[0]} prologue={pro:.0f} loop={loop:.0f} epilogue={epi:.0f} ticks; span={(h[:, 3].max() - h[:, 0].min()):.0f}")
            if not pp and float(h[:, 4].abs().sum()) > 0:   # single- / two-buffer loop: decode, first request, first landing, second chunk
                t = lambda k: (h[:, k] - h[:, 1]).clamp(min=0).mean()
                print(f"     after the loop stamp: offsets decoded +{t(4):.0f}, chunk 0 requested +{t(5):.0f}, landed (barrier) +{t(6):.0f}, chunk 1 landed +{t(7):.0f}")
            if pp and float(h[:, 12].abs().sum()) > 0:
                t = lambda k: (h[:, k] - h[:, 1]).clamp(min=0).mean()
                print(f"     after the loop stamp: decoded +{t(12):.0f}, chunk 0 requested and accumulators zeroed +{t(13):.0f}, landed +{t(14):.0f}, barrier passed +{t(15):.0f}")
            if pp:
                g0, g1 = h[:, 4:8].mean(0), h[:, 8:12].mean(0)
                print(f"     wave0 fetch={g0[0]:.0f} barrier={g0[1]:.0f} mfma={g0[2]:.0f} barrier={g0[3]:.0f} | wave4 fetch={g1[0]:.0f} barrier={g1[1]:.0f} mfma={g1[2]:.0f} barrier={g1[3]:.0f}")
