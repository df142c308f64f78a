#!/usr/bin/env python3
"""The `yogo infer` hot path (BASELINE configs[4]: batch 256, bf16) for rocprofv3 -- tools/collect_profile.sh runs it as
    rocprofv3 --kernel-trace --stats ... -- python3 tools/infer_profile.py e2e|post [reps]
  e2e  : eval forward on the bf16 matrix cores + fused decode/threshold/NMS (YOGO.forward_raw -> format_preds_batched) on
         synthetic images through a random-init base_model: its predictions are 'dense' (93 % of the cells fire), the NMS worst case
  post : the post-process alone on 'realistic' head outputs (100 objects per image): the fused kernel, and the two passes it
         replaces (decode_fwd_kernel + nms_batched_kernel<false>) on the same input, so one stats table shows both.
Prints one JSON line with HIP-event times."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from yogo_amd.model import YOGO
from yogo_amd.synthetic import raw_from_predictions, synthetic_images, synthetic_predictions
from yogo_amd.utils import format_preds_batched
from yogo_amd.utils.prediction_formatting import RawPredictions

which = sys.argv[1] if len(sys.argv) > 1 else "e2e"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
B = int(sys.argv[3]) if len(sys.argv) > 3 else 256
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = YOGO((772, 1032), 0.0425, 0.0555, 7, inference=True).to(dev).eval()


def timed(fn, n):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


out = {"which": which, "batch": B, "reps": reps}
with torch.no_grad():
    if which == "e2e":
        x = synthetic_images(B, 772, 1032, device=dev, seed=7)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out["forward_raw_ms"] = round(timed(lambda: m.forward_raw(x), reps), 3)
            out["forward_raw_plus_fused_postprocess_ms"] = round(timed(lambda: format_preds_batched(m.forward_raw(x)), reps), 3)
            out["forward_decode_plus_postprocess_ms"] = round(timed(lambda: format_preds_batched(m(x)), reps), 3)
    else:
        pred = synthetic_predictions(B, m.Sx, m.Sy, 7, K=100, device=dev)
        raw = raw_from_predictions(pred, m._Cxs, m._Cys, *m._decode_scalars()[:2])
        rp = RawPredictions(raw, m._Cxs, m._Cys, *m._decode_scalars(), True)
        out["fused_ms"] = round(timed(lambda: format_preds_batched(rp), reps), 4)
        out["two_pass_ms"] = round(timed(lambda: format_preds_batched(rp.decoded()), reps), 4)
        out["kept_per_image"] = round(float(format_preds_batched(rp)[2].float().mean()), 1)
print(json.dumps(out))
