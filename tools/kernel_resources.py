"""Register / LDS / scratch usage of the library's kernels as hipcc reports it (-Rpass-analysis=kernel-resource-usage).
usage: kernel_resources.py [name-filter ...] [-- -DDEFINE ...]"""
import re
import subprocess
import sys
import shutil

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from bourse_amd import _build

args = sys.argv[1:]
defs = [a[2:] for a in args if a.startswith("-D")]
filt = [a for a in args if not a.startswith("-")]
r = []
_build.build(out="/tmp/lib_kernel_resources.so", defines=defs, remarks=r)
filt_tool = shutil.which("c++filt") or shutil.which("llvm-cxxfilt")
for m in re.finditer(r"Function Name: (\S+).*?SGPRs: (\d+).*?VGPRs: (\d+).*?AGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+).*?"
                     r"Occupancy \[waves/SIMD\]: (\d+).*?LDS Size \[bytes/block\]: (\d+)", r[0], re.S):
    n = m.group(1)
    d = subprocess.run([filt_tool, n], capture_output=True, text=True).stdout.strip() if filt_tool else n
    d = d.split("(")[0].replace("void bkd::", "")
    if filt and not any(f in d for f in filt):
        continue
    print("%-44s SGPR %3s VGPR %3s AGPR %3s scratch %4s occ %2s LDS %6s" % (d[:44], m.group(2), m.group(3), m.group(4), m.group(5), m.group(6), m.group(7)))
