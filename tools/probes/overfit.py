"""sanity probe: the loss of a fixed batch falls under repeated optimisation steps, bf16 next to fp32 (not part of the product)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from yogo_amd.model import YOGO
from yogo_amd.synthetic import synthetic_images, synthetic_labels
from yogo_amd.train import HipTrainer
from yogo_amd.yogo_loss import YOGOLoss

dev = torch.device("cuda", 0)
for half in (False, True):
    torch.manual_seed(0)
    m = YOGO((772, 1032), 0.0425, 0.0555, 7).to(dev)
    m.train()
    B = 16
    imgs = synthetic_images(B, 772, 1032, device=dev, seed=1)
    labels = synthetic_labels(B, m.Sx, m.Sy, K=64, num_classes=7, device=dev, seed=2)
    tr = HipTrainer(m, YOGOLoss().to(dev), total_steps=200, half=half)
    out = []
    for it in range(80):
        tr.step(imgs, labels)
        if it % 10 == 0 or it == 79:
            out.append(round(tr.loss_components()["loss"], 3))
    print("bf16" if half else "fp32", out, flush=True)
