#!/usr/bin/env python3
"""VGPRs / scratch / LDS / occupancy of every fused kernel instantiation (hipcc -Rpass-analysis=kernel-resource-usage).
usage: python tools/kernel_resources.py [extra hipcc flags]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-function", "-Wno-pass-failed",
       "-c", os.path.join(ROOT, "zune-jpeg_amd/csrc/zj_kernels.hip"), "-o", "/tmp/zj_res.o", "-Rpass-analysis=kernel-resource-usage"] + sys.argv[1:]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = m.group(1); rows[cur] = {}
        continue
    m = re.search(r"remark:\s+(VGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).split()[0]] = int(m.group(2))
names = {0: "RGB", 1: "GRAY", 2: "YCBCR", 3: "RGBA", 4: "CHW"}
print(f"{'kernel <HS,VS,OUT,GEN,FAST,TS>':40s} VGPR scratch LDS   waves/SIMD")
for k, v in sorted(rows.items()):
    m = re.search(r"zj_fused_kernelILi(\d)ELi(\d)ELi(\d)ELi(\d)ELb(\d)ELb(\d)E", k)
    r = re.search(r"zj_fused_ragged_kernelILi(\d)ELi(\d)ELi(\d)ELb(\d)E", k)
    if r:  # the ragged family: packed generation, fast path + generic stores at the row ends
        hs, vs, o, t = map(int, r.groups())
        print(f"<{hs},{vs},{names[o]:5s},packed,rag ,{'staged' if t else 'direct'}>".ljust(40),
              f"{v.get('VGPRs', -1):4d} {v.get('ScratchSize', -1):7d} {v.get('LDS', -1):6d} {v.get('Occupancy', -1):4d}")
        continue
    sm = re.search(r"zj_fused_seam_kernelILi(\d)ELi(\d)ELi(\d)E", k)
    if sm:  # the seam family: aligned widths whose rows do not start on 128-byte boundaries (staged stores, shared lines written back)
        hs, vs, o = map(int, sm.groups())
        print(f"<{hs},{vs},{names[o]:5s},packed,seam,staged>".ljust(40),
              f"{v.get('VGPRs', -1):4d} {v.get('ScratchSize', -1):7d} {v.get('LDS', -1):6d} {v.get('Occupancy', -1):4d}")
        continue
    if not m:
        continue
    hs, vs, o, g, f, t = map(int, m.groups())
    print(f"<{hs},{vs},{names[o]:5s},{'packed' if g else 'wide  '},{'fast' if f else 'any '},{'staged' if t else 'direct'}>".ljust(40),
          f"{v.get('VGPRs', -1):4d} {v.get('ScratchSize', -1):7d} {v.get('LDS', -1):6d} {v.get('Occupancy', -1):4d}")
