import sys, time, torch
sys.path.insert(0, "/root/repo")
import bench as B
from yogo_amd.model import YOGO
from yogo_amd.synthetic import synthetic_images, synthetic_labels
from yogo_amd.train import HipTrainer
from yogo_amd.yogo_loss import YOGOLoss
dev = torch.device("cuda", 0)
torch.manual_seed(0)
m = YOGO((B.H, B.W), B.ANCHOR_W, B.ANCHOR_H, B.NUM_CLASSES).to(dev); m.train()
tr = HipTrainer(m, YOGOLoss().to(dev), total_steps=500, half=True)
x = synthetic_images(128, B.H, B.W, device=dev, seed=1); lab = synthetic_labels(128, m.Sx, m.Sy, K=64, num_classes=B.NUM_CLASSES, device=dev, seed=2)
for _ in range(15): tr.step(x, lab)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(30): tr.step(x, lab)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e3*(t1-t0)/30:.2f} ms/step, wall {1e3*(t2-t0)/30:.2f} ms/step")
