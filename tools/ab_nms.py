"""A/B of nms.hip variant builds (bash yogo_amd/csrc/build.sh variant TAG nms -D...): a script once per library, each in its own process.
    python tools/ab_nms.py TAG1,TAG2,... [script] [args...]     (TAG "base" = the product library; default script tools/bench_nms.py 256)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tags = sys.argv[1].split(",") if len(sys.argv) > 1 else ["base"]
script = sys.argv[2] if len(sys.argv) > 2 else os.path.join("tools", "bench_nms.py")
args = sys.argv[3:] if len(sys.argv) > 3 else ["256"]
for tag in tags:
    code = f"""
import os, sys
sys.argv = [{script!r}] + {args!r}
sys.path.insert(0, {ROOT!r})
from yogo_amd import _hip as H
if {tag!r} != 'base':
    H.LIB_PATH = os.path.join({ROOT!r}, 'yogo_amd', 'lib', 'libyogo_hip_{tag}.so')
__file__ = os.path.join({ROOT!r}, {script!r})
exec(compile(open(__file__).read(), __file__, 'exec'))
"""
    print("==", tag, flush=True)
    subprocess.run([sys.executable, "-c", code])
