"""A/B in one process: weight gradients on a second stream (all layers / layers <= 4 only) vs one stream (experiments only)"""
import sys, time, torch
sys.path.insert(0, "/root/repo")
import bench as B
import yogo_amd.engine as E
from yogo_amd.model import YOGO
from yogo_amd.synthetic import synthetic_images, synthetic_labels
from yogo_amd.train import HipTrainer
from yogo_amd.yogo_loss import YOGOLoss
dev = torch.device("cuda", 0)
torch.manual_seed(0)
m = YOGO((B.H, B.W), B.ANCHOR_W, B.ANCHOR_H, B.NUM_CLASSES).to(dev); m.train()
tr = HipTrainer(m, YOGOLoss().to(dev), total_steps=5000, half=True)
x = synthetic_images(128, B.H, B.W, device=dev, seed=1); lab = synthetic_labels(128, m.Sx, m.Sy, K=64, num_classes=B.NUM_CLASSES, device=dev, seed=2)
for _ in range(15): tr.step(x, lab)
for rnd in range(3):
    for side, mx in ((False, 99), (True, 99), (True, 4)):
        E._WGRAD_SIDE_STREAM = side; E._WGRAD_SIDE_MAX_LAYER = mx
        for _ in range(3): tr.step(x, lab)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): tr.step(x, lab)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        print(f"side={side} max_layer={mx}: {1e3*(t1-t0)/20:.3f} ms/step", flush=True)
