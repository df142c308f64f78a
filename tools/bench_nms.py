"""batched threshold+NMS throughput on realistic (sparse) and dense prediction tensors, B images at 97x129 cells"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import yogo_oracle as O
from yogo_amd.utils import format_preds_batched

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
base = O.synthetic_predictions(16, 129, 97, num_classes=7, K=100, seed=50)
real = base.repeat(B // 16, 1, 1, 1).cuda()
g = torch.Generator().manual_seed(51)
d = O.decode(torch.randn(16, 12, 97, 129, generator=g) * 1.5, *O.make_grids(129, 97), 0.0425, 0.0555, inference=True)
d[:, 4] = torch.rand(16, 97, 129, generator=g) * 0.55 + 0.45
dense = d.repeat(B // 16, 1, 1, 1).cuda()
for name, pred in (("realistic", real), ("dense", dense)):
    for _ in range(2):
        rows, cells, counts = format_preds_batched(pred)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        rows, cells, counts = format_preds_batched(pred)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"nms {name}: B={B} {ms:.3f} ms  {B / ms * 1e3:.0f} img/s  mean kept {counts.float().mean().item():.1f}  "
          f"({B * 12 * 12513 * 4 / ms / 1e6:.1f} GB/s of prediction bytes)")
