"""Register / spill report per kernel instantiation of one .hip source (hipcc -Rpass-analysis=kernel-resource-usage).
usage: python tools/kernel_resources.py openvis_amd/csrc/gemm_f16_pp.hip [name filter]"""
import re
import subprocess
import sys

src, flt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-c", src, "-o", "/dev/null",
                    "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
PATS = {"VGPR": r" VGPRs: (\d+)", "AGPR": r"AGPRs: (\d+)", "spill": r"VGPR Spill: (\d+)", "scratch": r"ScratchSize \[bytes/lane\]: (\d+)",
        "LDS": r"LDS Size \[bytes/block\]: (\d+)", "occ": r"Occupancy \[waves/SIMD\]: (\d+)"}
for b in re.split(r"(?=remark: [^\n]*Function Name)", r.stderr):
    m = re.search(r"Function Name: (\S+)", b)
    if not m:
        continue
    d = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
    if flt and flt not in d:
        continue
    t = d[d.index("<"):d.rindex(">") + 1] if "<" in d else ""
    vals = " ".join(f"{k} {(re.search(p, b) or [None, '?'])[1]}" for k, p in PATS.items())
    print(f"{d.split('<')[0].split('::')[-1]}{t}  {vals}")
