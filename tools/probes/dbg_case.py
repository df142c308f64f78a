import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo/oracle")
import torch, numpy as np
import torch.nn.functional as F
from test_gpu_bf16 import to8c, from8c, bf, H
h = H()
B, Cin, Cout, IH, IW, k, s = (2, 16, 32, 20, 37, 3, 1)
g = torch.Generator().manual_seed(1)
x = bf(torch.randn(B, Cin, IH, IW, generator=g))
w = torch.randn(Cout, Cin, k, k, generator=g) / np.sqrt(Cin * k * k)
wb = bf(w)
ref = F.conv2d(x, wb, None, stride=s, padding=1)
OH, OW = ref.shape[2:]
st = h.stream_ptr()
packed = torch.empty(h.query_size("yogo_conv_bf16_packed_bytes", Cin, Cout, k, 0), dtype=torch.uint8, device="cuda")
h.call("yogo_conv_bf16_pack", w.cuda(), None, packed, Cin, Cout, k, 0, st)
x8 = to8c(x)
out = torch.full((B, h.lib().yogo_bf16_channel_blocks(Cout), OH, OW, 8), float("nan"), dtype=torch.bfloat16, device="cuda")
h.call("yogo_conv2d_fwd_bf16", x8, packed, None, out, None, None, None, B, Cin, Cout, IH, IW, k, s, 0, st)
got = from8c(out, Cout)
err = (got - ref).abs()
print("nan count", torch.isnan(got).sum().item(), "of", got.numel())
for c in range(Cout):
    e = err[:, c]
    print(c, float(torch.nan_to_num(e, nan=99).max()), int(torch.isnan(got[:, c]).sum()))
bad = torch.nonzero(torch.nan_to_num(err, nan=99) > 0.05)
print(bad[:20])
