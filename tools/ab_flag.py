"""In-process A/B of whole training steps between the two settings of an engine plan flag:
    python tools/ab_flag.py _L01_FUSE_BWD [rounds] [steps]
One trainer, alternating blocks of `steps` steps with the flag False / True; prints ms per step per block."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import yogo_amd.engine as E
from yogo_amd.model import YOGO
from yogo_amd.train import HipTrainer
from yogo_amd.yogo_loss import YOGOLoss
import yogo_oracle as O

if __name__ == "__main__":
    flag = sys.argv[1]
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    assert isinstance(getattr(E, flag), bool)
    B = 128
    x = torch.randint(0, 256, (B, 1, 772, 1032), dtype=torch.uint8, device="cuda")
    torch.manual_seed(0)
    model = YOGO((772, 1032), 0.0425, 0.0555, 7, clip_value=1.0).cuda()
    model.train()
    lab = O.synthetic_labels(B, model.Sx, model.Sy, K=30, num_classes=7, seed=1).cuda()
    tr = HipTrainer(model, YOGOLoss().cuda(), total_steps=100000, half=True)
    res = {False: [], True: []}
    for v in (False, True):
        setattr(E, flag, v)
        for _ in range(3):
            tr.step(x, lab)
    for _ in range(rounds):
        for v in (False, True):
            setattr(E, flag, v)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                tr.step(x, lab)
            torch.cuda.synchronize()
            res[v].append((time.perf_counter() - t0) * 1e3 / steps)
    for v in (False, True):
        print(f"{flag} = {str(v):5s} " + " ".join(f"{t:.3f}" for t in res[v]) + f"   median {sorted(res[v])[len(res[v]) // 2]:.3f} ms/step")
