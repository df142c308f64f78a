"""Does bf16 training TRAIN like fp32 training?  (the step tests hold one or two steps; the reference's loop is
yogo/train.py:295-339.)  Thirty optimisation steps of base_model on three fixed synthetic batches with `half=True` (bf16 storage + bf16 MFMA) and with `half=False` (fp32), same initial weights, same
batches in the same order, Dropout2d off: the two loss trajectories have to fall together."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# stated bounds: both runs reduce the loss by at least 40 % over the 30 steps; the mean of the last five losses of the two runs
# agrees within 5 %; no single step's loss differs by more than 10 % of the fp32 value (bf16 storage is a perturbation of the
# forward pass at the 1e-3 level, Adam's first steps are sign-like: measured values are printed by the test)
MIN_DROP, FINAL_RTOL, STEP_RTOL = 0.40, 0.05, 0.10


def test_bf16_and_fp32_loss_trajectories_fall_together():
    from yogo_amd.model import YOGO
    from yogo_amd.train import HipTrainer
    from yogo_amd.yogo_loss import YOGOLoss
    import yogo_oracle as O

    H, W, C = 193, 258, 4
    steps, B = 30, 4
    # fixed synthetic batches with objects (the fake-data images are 772x1032 noise with 2-3 boxes; what is under test is the
    # optimiser trajectory of the two precisions, not the data loader -- tests/test_gpu_cli.py runs that path)
    xs = [O.synthetic_images(B, H, W, seed=500 + i).cuda() for i in range(3)]
    runs = {}
    for half in (False, True):
        torch.manual_seed(3)
        m = YOGO((H, W), 0.0425, 0.0555, C, clip_value=1.0).cuda()
        m.train()
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout2d):
                mod.p = 0.0
        labs = [O.synthetic_labels(B, m.Sx, m.Sy, K=12, num_classes=C, seed=600 + i).cuda() for i in range(3)]
        tr = HipTrainer(m, YOGOLoss().cuda(), learning_rate=3e-4, total_steps=steps, half=half)
        losses = []
        for s in range(steps):
            tr.step(xs[s % 3], labs[s % 3])
            losses.append(tr.loss_components()["loss"])
        runs[half] = losses
    f32, b16 = runs[False], runs[True]
    print("fp32:", [round(v, 3) for v in f32])
    print("bf16:", [round(v, 3) for v in b16])
    worst = max(abs(a - b) / abs(a) for a, b in zip(f32, b16))
    fin32, fin16 = sum(f32[-5:]) / 5, sum(b16[-5:]) / 5
    print(f"largest per-step difference {worst:.3%}; mean of the last five losses fp32 {fin32:.4f} bf16 {fin16:.4f} ({abs(fin16 - fin32) / fin32:.3%})")
    assert all(v == v and v < 1e6 for v in f32 + b16)
    assert f32[-1] < (1 - MIN_DROP) * f32[0] and b16[-1] < (1 - MIN_DROP) * b16[0]
    assert abs(fin16 - fin32) <= FINAL_RTOL * fin32
    assert worst <= STEP_RTOL
