import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
from yogo_amd import _hip as H
import bench_conv_bf16 as BC
H.launch_log(True)
for w in ["l4m", "l2m", "l5a", "l4a", "l3m", "l3a", "l1a", "l2a"]:
    BC.bench(w[:-1], 128, *BC.LAYERS[w[:-1]], w[-1], reps=3)
print(H.read_launch_log())
