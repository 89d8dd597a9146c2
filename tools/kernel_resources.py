#!/usr/bin/env python3
"""Per-kernel registers / scratch / occupancy / LDS of the gfx950 build (cross-compiles, no GPU needed).
usage: tools/kernel_resources.py [extra hipcc flags, e.g. -DPS_BS_G=3]"""
import os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "pypore_amd", "csrc", "poreseg.hip")
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(root, "include"),
       "-S", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", src, "-o", "/tmp/poreseg_gfx950.s"] + sys.argv[1:]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r"remark:\s+(.*?)\s+\[-Rpass", line)
    if not m:
        continue
    t = m.group(1)
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.rsplit(":", 1)
        cur[k.strip()] = v.strip()
names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows),
                       capture_output=True, text=True).stdout.splitlines()
print(f"{'kernel':72s} {'VGPR':>5s} {'AGPR':>5s} {'scr':>5s} {'occ':>4s} {'LDS':>7s}")
for r, n in zip(rows, names):
    n = re.sub(r"\(.*", "", n).replace("void ", "")
    print(f"{n[:72]:72s} {r.get('VGPRs','?'):>5s} {r.get('AGPRs','?'):>5s} {r.get('ScratchSize [bytes/lane]','?'):>5s} "
          f"{r.get('Occupancy [waves/SIMD]','?'):>4s} {r.get('LDS Size [bytes/block]','?'):>7s}")
