"""the post-process on a random-init network's own predictions (the 'dense' workload of tools/infer_profile.py e2e): time and kept boxes per image
    python tools/bench_nms_net.py [B]        (through tools/ab_nms.py for a library A/B)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("NMS_LIB"):   # (a variant library, e.g. under rocprofv3 where tools/ab_nms.py's child processes are no option)
    from yogo_amd import _hip as _H
    _H.LIB_PATH = os.path.join(os.path.dirname(_H.LIB_PATH), f"libyogo_hip_{os.environ['NMS_LIB']}.so")
from yogo_amd.model import YOGO
from yogo_amd.synthetic import synthetic_images
from yogo_amd.utils import format_preds_batched

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = YOGO((772, 1032), 0.0425, 0.0555, 7, inference=True).to(dev).eval()
x = synthetic_images(B, 772, 1032, device=dev, seed=7)
with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
    rp = m.forward_raw(x)
    dec = rp.decoded()
fire = float((dec[:, 4] > 0.5).float().mean())
for name, inp in (("fused", rp), ("decoded", dec)):
    for _ in range(2):
        rows, cells, counts = format_preds_batched(inp)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        rows, cells, counts = format_preds_batched(inp)
    e1.record()
    torch.cuda.synchronize()
    print(f"nms net {name}: B={B} {e0.elapsed_time(e1) / 3:.3f} ms  fire {fire:.3f}  mean kept {counts.float().mean().item():.1f}  min {int(counts.min())} max {int(counts.max())}")
