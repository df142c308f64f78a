"""stamps of the persistent wavefront-specialised kernel (diagnostic library): per workgroup -- lifetime, ticks inside the chunk bodies, the
tile seams (epilogue arithmetic), the next-tile decode, and the wait + barrier statement of step 8; with ablation bits
(1 = no output stores, 2 = no seam arithmetic, 4 = no LDS-DMA).
    bash yogo_amd/csrc/build.sh diag && python tools/stamps_ws.py [B] [which] [dbg,dbg,...]"""
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
    which = sys.argv[2].split(",") if len(sys.argv) > 2 else ["l5d", "l3s"]
    dbgs = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0, 1, 2, 3, 4]
    lib = H.lib()
    lib.yogo_diag_conv_bf16.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t]
    st = torch.zeros(512 * 16, dtype=torch.int64, device="cuda")
    for w in which:
        for dbg in dbgs:
            lib.yogo_diag_conv_bf16(dbg, 0, None, 0)
            BC.bench(w[:-1], B, *BC.LAYERS[w[:-1]], w[-1], reps=5)
            st.zero_()
            lib.yogo_diag_conv_bf16(dbg, 0, st.data_ptr(), st.numel() * 8)
            BC.bench(w[:-1], B, *BC.LAYERS[w[:-1]], w[-1], reps=1)
            torch.cuda.synchronize()
            hh = st.view(512, 16).cpu().double()
            h, lp = hh[:256], hh[256:]
            h = h[h[:, 1] != 0]
            if h.numel() == 0:
                print("  (no stamps)")
                continue
            life = h[:, 1] - h[:, 0]
            nt = h[:, 5]
            print(f"  {w} dbg={dbg}: wgs={h.shape[0]} tiles/wg={nt.mean():.1f} life={life.mean():.0f} (max {life.max():.0f}) | compute per tile: life {(life / nt).mean():.0f} "
                  f"chunks {(h[:, 2] / nt).mean():.0f} (first {(h[:, 10] / nt).mean():.0f}, c1 {(h[:, 12] / nt).mean():.0f}, c2 {(h[:, 13] / nt).mean():.0f}, last {(h[:, 11] / nt).mean():.0f}) seam {(h[:, 3] / nt).mean():.0f} x8 {(h[:, 6] / nt).mean():.0f} | loader per tile: work {(h[:, 8] / nt).mean():.0f} "
                  f"wait+barrier {(h[:, 9] / nt).mean():.0f}; loader work in periods 0..3: " + " ".join(f"{(lp[:h.shape[0], k] / nt).mean():.0f}" for k in range(4))
                  + "; their vmcnt waits: " + " ".join(f"{(lp[:h.shape[0], 4 + k] / nt).mean():.0f}" for k in range(4)))
