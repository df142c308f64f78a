"""prints the launch plans of the bf16 weight-gradient kernels of base_model at B = 128 (experiments only)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from yogo_amd import _hip as H   # noqa: E402
import ab_wgrad_bf16 as W        # noqa: E402

H.LIB_PATH = os.path.join(ROOT, "yogo_amd", "lib", "libyogo_hip.so")
H.launch_log(True)
for w in ["l1", "l2", "l3", "l4", "l5", "l7"]:
    W.bench(w, 128, *W.LAYERS[w], reps=2)
for line in sorted(set(H.read_launch_log())):
    if "wgrad_bf16_kernel" in line:
        print(line)
