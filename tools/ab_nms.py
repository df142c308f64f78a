"""A/B of nms.hip variant builds (bash yogo_amd/csrc/build.sh variant nmsN nms -DNMS_VARIANT=N): tools/bench_nms.py once per library, each in its own process."""
import os, sys, subprocess
ROOT = "/root/repo" if os.path.exists("/root/repo/tools") else os.getcwd()
for tag in ("base", "nmso", "base", "nmso"):
    code = f"""
import os, sys
sys.argv = ['x', '256']
sys.path.insert(0, {ROOT!r})
from yogo_amd import _hip as H
if {tag!r} != 'base':
    H.LIB_PATH = os.path.join({ROOT!r}, 'yogo_amd', 'lib', 'libyogo_hip_{tag}.so')
__file__ = os.path.join({ROOT!r}, 'tools', 'bench_nms.py')
exec(open(__file__).read())
"""
    print("==", tag, flush=True)
    subprocess.run([sys.executable, "-c", code])
