"""where do the persistent and the tiled kernel differ?  (debug aid for tests/test_gpu_ws.py)
    python tools/probes/ws_mismatch.py kind B Cin H W"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_ws as T

kind, B, Cin, H, W = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
a, sa, la = T._run(False, kind, B, Cin, H, W, seed=11)
b, sb, lb = T._run(True, kind, B, Cin, H, W, seed=11)
print(lb[-1])
d = (a.view(torch.int16) != b.view(torch.int16))   # [B, Cb, H, W, 8]
print("differ:", d.float().mean().item())
print("per image:", d.float().mean(dim=(1, 2, 3, 4)).tolist())
print("per channel block:", [round(x, 4) for x in d.float().mean(dim=(0, 2, 3, 4)).tolist()])
pix = d.any(dim=4).any(dim=1).view(B, H * W)   # [B, pixels]
for i in range(B):
    idx = pix[i].nonzero().flatten().tolist()
    print(f"image {i}: {len(idx)} pixels differ; first {idx[:20]} last {idx[-10:]}")
print("per channel of block 0:", [round(x, 4) for x in d[:, 0].float().mean(dim=(0, 1, 2)).tolist()])
# band-local view (bands of TW columns): which tile-local pixel indices (row-major inside the band) differ in band 0 of image 0
import re
m = re.search(r"TW=(\d+)", lb[-1])
TW = int(m.group(1))
for band in range(0, (W + TW - 1) // TW):
    sub = d[0, 0, :, band * TW:(band + 1) * TW, :].any(dim=2)   # [H, bw]
    loc = sub.reshape(-1).nonzero().flatten().tolist()
    print(f"band {band}: {len(loc)} of {sub.numel()} tile-local pixels differ: {loc[:40]}")
x = (a.float() - b.float())[0, 0]
print("max abs diff per channel:", x.abs().amax(dim=(0, 1)).tolist())
print("sample old/new at first differing pixel:", a[0, 0].reshape(-1, 8)[pix[0].nonzero()[0, 0]].tolist(), b[0, 0].reshape(-1, 8)[pix[0].nonzero()[0, 0]].tolist())
