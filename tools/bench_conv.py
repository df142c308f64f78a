"""micro-benchmark of single conv launches (experiments; not part of the product)"""
import sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yogo_amd import _hip as H

def bench(B, Cin, Cout, IH, IW, k, s, reps=10):
    pad = 1 if k == 3 else 0
    OH, OW = (IH + 2*pad - k)//s + 1, (IW + 2*pad - k)//s + 1
    x = torch.randn(B, Cin, IH, IW, device="cuda")
    w = torch.randn(Cout, Cin, k, k, device="cuda") * 0.05
    b = torch.randn(Cout, device="cuda")
    st = H.stream_ptr()
    packed = torch.empty(H.query_size("yogo_conv_packed_bytes", Cin, Cout, k, s, 0)//4, device="cuda")
    H.call("yogo_conv_pack_f32", w, packed, Cin, Cout, k, s, 0, st)
    out = torch.empty(B, Cout, OH, OW, device="cuda")
    f = lambda: H.call("yogo_conv2d_fwd_f32", x, packed, b, out, None, None, None, B, Cin, Cout, IH, IW, k, s, 1, st)
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)/reps
    fl = 2.0*B*Cout*Cin*k*k*OH*OW
    print(f"fwd B={B} {Cin}->{Cout} {IH}x{IW} k{k} s{s}: {ms:.3f} ms  {fl/ms/1e9:.1f} TF  dbg={os.environ.get('YOGO_IGEMM_DBG','0')}")

if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    bench(B, 128, 128, 97, 129, 3, 1)
    bench(B, 64, 128, 193, 258, 3, 1)
    bench(B, 128, 128, 193, 258, 3, 2)
    bench(B, 32, 64, 386, 516, 3, 2)
    bench(B, 16, 32, 386, 516, 3, 1)
