"""does the row pitch of the NCHW8c tensors matter?  the 16 -> 32 layer forward at 386 x 516 (rows of 8256 B: every other row starts
in the middle of a 128-byte line) against 386 x 512 and 386 x 520 (whole lines), same kernel; GB/s of algorithmic bytes"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
import bench_conv_bf16 as BC
for r in range(3):
    for name, dims in (("w516", (16, 32, 386, 516, 3, 1)), ("w512", (16, 32, 386, 512, 3, 1)), ("w520", (16, 32, 386, 520, 3, 1)),
                       ("l2w516", (32, 64, 386, 516, 3, 2)), ("l2w512", (32, 64, 386, 512, 3, 2))):
        for kind in "ad":
            BC.bench(name, 128, *dims, kind, reps=10)
