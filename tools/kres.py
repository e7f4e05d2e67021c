#!/usr/bin/env python3
"""Summarise hipcc -Rpass-analysis=kernel-resource-usage for one .hip file (dev tool)."""
import re, subprocess, sys
src = sys.argv[1]
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950",
       "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/tmp/kres.o"] + sys.argv[2:]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None; rows = []
for line in out.splitlines():
    m = re.search(r"remark:\s+(.*?)\s+\[-Rpass", line)
    if not m: 
        if "error" in line: print(line)
        continue
    t = m.group(1)
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":",1)[1].strip()}; rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":",1); cur[k.strip()] = v.strip()
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(anonymous namespace\)::", "", name)[:70]
    print(f"{name:70s} sgpr={r.get('TotalSGPRs','?'):>4} vgpr={r.get('VGPRs','?'):>4} agpr={r.get('AGPRs','?'):>4} "
          f"scratch={r.get('ScratchSize [bytes/lane]','?'):>4} occ={r.get('Occupancy [waves/SIMD]','?'):>2} "
          f"sspill={r.get('SGPRs Spill','?'):>3} vspill={r.get('VGPRs Spill','?'):>3} lds={r.get('LDS Size [bytes/block]','?'):>6}")
