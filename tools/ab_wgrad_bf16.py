"""A/B of wgrad_bf16_kernel variants in ONE process (diagnostic library; experiments only).
    bash yogo_amd/csrc/build.sh diag && python tools/ab_wgrad_bf16.py [B] [rounds]"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from yogo_amd import _hip as H

H.LIB_PATH = os.path.join(ROOT, "yogo_amd", "lib", "libyogo_hip_diag.so")
LAYERS = {"l1": (16, 32, 386, 516, 3, 1), "l2": (32, 64, 386, 516, 3, 2), "l3": (64, 128, 193, 258, 3, 1),
          "l4": (128, 128, 193, 258, 3, 2), "l5": (128, 128, 97, 129, 3, 1), "l7": (128, 12, 97, 129, 1, 1)}


def blocks(c):
    return ((c + 15) // 16) * 2


def bench(name, B, Cin, Cout, IH, IW, k, s, reps=10):
    pad = 1 if k == 3 else 0
    OH, OW = (IH + 2 * pad - k) // s + 1, (IW + 2 * pad - k) // s + 1
    st = H.stream_ptr()
    x8 = torch.randn(B, blocks(Cin), IH, IW, 8, device="cuda").to(torch.bfloat16)
    g8 = torch.randn(B, blocks(Cout), OH, OW, 8, device="cuda").to(torch.bfloat16)
    ws = torch.empty(H.query_size("yogo_conv2d_wgrad_bf16_workspace_bytes", B, Cin, Cout, IH, IW, k, s) // 4, device="cuda")
    dw, db = torch.empty(Cout, Cin, k, k, device="cuda"), torch.empty(Cout, device="cuda")
    f = lambda: H.call("yogo_conv2d_wgrad_bf16", x8, g8, dw, db, ws, B, Cin, Cout, IH, IW, k, s, 1.0, st)
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
    print(f"{name} wgrad B={B} {Cin}->{Cout} {IH}x{IW} k{k} s{s}: {ms:.3f} ms  {2.0 * B * Cout * Cin * k * k * OH * OW / ms / 1e9:.1f} TF", flush=True)


if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    which = sys.argv[3].split(",") if len(sys.argv) > 3 else ["l3", "l4", "l5"]
    bits = [int(v) for v in sys.argv[4].split(",")] if len(sys.argv) > 4 else [0, 1]
    lib = H.lib()
    lib.yogo_diag_wgrad_bf16.argtypes = [ctypes.c_int]
    if os.environ.get("WB_STAMPS"):   # per-workgroup time sums of wave 0 (s_memtime ticks): wait + barrier / DMA issue / step loop
        lib.yogo_diag_wgrad_bf16_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
        st = torch.zeros(1 << 16, dtype=torch.int64, device="cuda")
        for w in which:
          for dbits in bits:
            lib.yogo_diag_wgrad_bf16(dbits)
            lib.yogo_diag_wgrad_bf16_stamps(None, 0)
            bench(w, B, *LAYERS[w], reps=5)
            lib.yogo_diag_wgrad_bf16_stamps(st.data_ptr(), st.numel() * 8)
            bench(w, B, *LAYERS[w], reps=1)
            torch.cuda.synchronize()
            h = st.view(-1, 4).cpu().double()
            h = h[h[:, 3] != 0]
            print(f"  {w} diag={dbits}: workgroups {h.shape[0]}; per workgroup: wait+barrier {h[:, 0].mean():.0f}, DMA issue {h[:, 1].mean():.0f}, steps {h[:, 2].mean():.0f}, loop total {h[:, 3].mean():.0f} ticks")
        sys.exit(0)
    for r in range(rounds):
        for b in bits:
            lib.yogo_diag_wgrad_bf16(b)
            print(f"--- round {r} diag={b}", flush=True)
            for w in which:
                bench(w, B, *LAYERS[w])
