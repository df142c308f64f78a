"""where do conv_bf16_ws3_kernel and the tiled kernel differ?  (debug aid for tests/test_gpu_ws.py)"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import test_gpu_ws as T
from yogo_amd import _hip as _H
if os.environ.get("YOGO_LIB"):
    _H.LIB_PATH = os.environ["YOGO_LIB"]
kind, B, Cin, H, W = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
y0, l0, _ = T._run_s2f(False, kind, B, Cin, H, W, seed=41)
y1, l1, _ = T._run_s2f(True, kind, B, Cin, H, W, seed=41)
print(l1)
d = (y0.view(torch.int16) != y1.view(torch.int16))   # [B][16][OH][OW][8]
print("differ:", d.float().mean().item(), "nan:", torch.isnan(y1.float()).float().mean().item(), "still poisoned (7.0):", (y1.float() == 7.0).float().mean().item())
print("per channel block:", [round(x, 3) for x in d.float().mean((0, 2, 3, 4)).tolist()])
print("per image:", [round(x, 3) for x in d.float().mean((1, 2, 3, 4)).tolist()])
OH, OW = d.shape[2], d.shape[3]
pm = d.any(4).any(1)[0].float()   # [OH][OW] of image 0
print("rows with mismatches (image 0):", [i for i in range(OH) if pm[i].any()][:40])
r = [i for i in range(OH) if pm[i].any()]
if r:
    print("cols in first bad row:", [j for j in range(OW) if pm[r[0], j]][:60])
