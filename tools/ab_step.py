"""In-process A/B of whole training steps between library builds (bash yogo_amd/csrc/build.sh variant TAG FILE -D...):
    python tools/ab_step.py TAG1,TAG2,... [rounds] [steps]      (TAG "base" = the product library)
One trainer per library (same seed), alternating blocks of `steps` steps; prints ms per step per block."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from yogo_amd import _hip as H
import yogo_amd.engine as E
from yogo_amd.model import YOGO
from yogo_amd.train import HipTrainer
from yogo_amd.yogo_loss import YOGOLoss
import yogo_oracle as O

if __name__ == "__main__":
    tags = sys.argv[1].split(",")
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    libs = {}
    for t in tags:
        H._lib, H.LIB_PATH = None, os.path.join(ROOT, "yogo_amd", "lib", "libyogo_hip.so" if t == "base" else f"libyogo_hip_{t}.so")
        libs[t] = H.lib()
    B = 128
    x = torch.randint(0, 256, (B, 1, 772, 1032), dtype=torch.uint8, device="cuda")
    trs, res = {}, {}
    for t in tags:
        H._lib = libs[t]
        E._WGRAD_QUEUE = None   # (a queue belongs to the library that made it)
        torch.manual_seed(0)
        model = YOGO((772, 1032), 0.0425, 0.0555, 7, clip_value=1.0).cuda()
        model.train()
        lab = O.synthetic_labels(B, model.Sx, model.Sy, K=30, num_classes=7, seed=1).cuda()
        tr = HipTrainer(model, YOGOLoss().cuda(), total_steps=100000, half=True)
        for _ in range(3):
            tr.step(x, lab)
        trs[t] = (tr, lab, E._WGRAD_QUEUE)
    for _ in range(rounds):
        for t in tags:
            H._lib = libs[t]
            tr, lab, E._WGRAD_QUEUE = trs[t]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                tr.step(x, lab)
            torch.cuda.synchronize()
            res.setdefault(t, []).append((time.perf_counter() - t0) * 1e3 / steps)
    for t in tags:
        print(f"{t:10s} " + " ".join(f"{v:.3f}" for v in res[t]) + f"   median {sorted(res[t])[len(res[t]) // 2]:.3f} ms/step")
