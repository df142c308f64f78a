"""stamps of the persistent stride-2 data-gradient kernel (conv_bf16_ws2_kernel, diagnostic library): per workgroup and tile -- ticks of
the compute wavefront in pass A / its epilogue / pass B / its epilogue, inside the barrier statements, and of the loader inside its
vmcnt waits and barriers; ablation bits 1 = no output stores, 2 = no epilogue arithmetic, 4 = no LDS-DMA.
    bash yogo_amd/csrc/build.sh diag && python tools/stamps_ws2.py [B] [which] [dbg,dbg,...]"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from yogo_amd import _hip as H

H.LIB_PATH = os.environ.get("YOGO_DIAG_LIB", os.path.join(ROOT, "yogo_amd", "lib", "libyogo_hip_diag.so"))   # (an ablation build: YOGO_DIAG_LIB=...)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_conv_bf16 as BC   # noqa: E402

if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    which = sys.argv[2].split(",") if len(sys.argv) > 2 else ["l4m"]
    dbgs = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0, 1, 2, 4, 7]
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
            h = st.view(512, 16).cpu().double()[:256]
            h = h[h[:, 1] != 0]
            if h.numel() == 0:
                print("  (no stamps)")
                continue
            life = h[:, 1] - h[:, 0]
            nt = h[:, 6]
            f = lambda c: (h[:, c] / nt).mean()
            print(f"  {w} dbg={dbg}: wgs={h.shape[0]} tiles/wg={nt.mean():.1f} life={life.mean():.0f} | per tile: {(life / nt).mean():.0f} = pass A {f(2):.0f} + epi A {f(3):.0f} "
                  f"+ pass B {f(4):.0f} + epi B {f(5):.0f}")
