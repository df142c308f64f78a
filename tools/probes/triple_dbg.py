"""which kernel variant makes the bf16 step of triple_filters at 193x258 differ from fp32 (diagnostic library; experiments only)"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
from yogo_amd import _hip as H
H.LIB_PATH = os.path.join(ROOT, "yogo_amd", "lib", "libyogo_hip_diag.so")
import yogo_oracle as O
from yogo_amd.model import YOGO
from yogo_amd.model_defns import MODELS
from yogo_amd.train import HipTrainer
from yogo_amd.yogo_loss import YOGOLoss
lib = H.lib()
name = sys.argv[1] if len(sys.argv) > 1 else "triple_filters"
Hh, W, B = 193, 258, 3


def run(half, steps=2):
    torch.manual_seed(1)
    m = YOGO((Hh, W), 0.0425, 0.0555, 5, model_func=MODELS[name]).cuda(); m.train()
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout2d): mod.p = 0.0
    x = torch.randint(0, 256, (B, 1, Hh, W), dtype=torch.uint8, generator=torch.Generator().manual_seed(2)).cuda()
    lab = O.synthetic_labels(B, m.Sx, m.Sy, K=4, num_classes=5, seed=3).cuda()
    tr = HipTrainer(m, YOGOLoss().cuda(), total_steps=4, half=half)
    out = []
    for _ in range(steps):
        tr.step(x, lab); out.append(round(tr.loss_components()["loss"], 4))
    return out


print("fp32", run(False))
for label, fn in (("all on", lambda: None), ("ring off", lambda: lib.yogo_diag_conv_bf16_ring(0)), ("lean4 off", lambda: lib.yogo_diag_conv_bf16_lean4(0)),
                  ("pp off", lambda: lib.yogo_diag_conv_bf16_pp(0)), ("wgrad LEAN2 off", lambda: lib.yogo_diag_wgrad_bf16(32))):
    lib.yogo_diag_conv_bf16_ring(1); lib.yogo_diag_conv_bf16_lean4(1); lib.yogo_diag_conv_bf16_pp(1); lib.yogo_diag_wgrad_bf16(0)
    fn()
    print(f"bf16 {label}:", run(True), flush=True)


def grads(half):
    torch.manual_seed(1)
    m = YOGO((Hh, W), 0.0425, 0.0555, 5, model_func=MODELS[name]).cuda(); m.train()
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout2d): mod.p = 0.0
    x = torch.randint(0, 256, (B, 1, Hh, W), dtype=torch.uint8, generator=torch.Generator().manual_seed(2)).cuda()
    lab = O.synthetic_labels(B, m.Sx, m.Sy, K=4, num_classes=5, seed=3).cuda()
    tr = HipTrainer(m, YOGOLoss().cuda(), total_steps=4, half=half)
    tr.step(x, lab)
    names = [n for n, p in m.named_parameters()]
    sizes = [p.numel() for n, p in m.named_parameters()]
    return names, sizes, tr.flat.grad.clone()


lib.yogo_diag_conv_bf16_ring(1); lib.yogo_diag_conv_bf16_lean4(1); lib.yogo_diag_conv_bf16_pp(1); lib.yogo_diag_wgrad_bf16(0)
n, sz, g32 = grads(False)
_, _, g16 = grads(True)
off = 0
for nm, k in zip(n, sz):
    a, b = g32[off:off + k], g16[off:off + k]
    cos = float((a * b).sum() / (a.norm() * b.norm() + 1e-30))
    print(f"  {nm:28s} n={k:8d} |g32|={float(a.norm()):.3e} |g16|={float(b.norm()):.3e} cos={cos:.4f} clamped32={float((a.abs() >= 0.0999).float().mean()):.3f}")
    off += k
