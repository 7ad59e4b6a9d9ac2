"""Registers / scratch / LDS / occupancy of the kernels of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage).
usage: python tools/kernel_resources.py [source.hip] [name-filter]"""
import re, subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.abspath(sys.argv[1]) if len(sys.argv) > 1 else os.path.join(ROOT, "centrifuger_amd/csrc/cfr_device.hip")
flt = sys.argv[2] if len(sys.argv) > 2 else ""
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", "-o", "/dev/null", src,
                      "-Rpass-analysis=kernel-resource-usage"] + sys.argv[3:], stderr=subprocess.PIPE, stdout=subprocess.PIPE, cwd=os.path.dirname(src)).stderr.decode()
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"remark:\s+(Function Name|VGPRs|AGPRs|TotalSGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
    if not m: continue
    if m.group(1) == "Function Name":
        cur = subprocess.run(["c++filt", m.group(2)], stdout=subprocess.PIPE).stdout.decode().strip()
        cur = re.sub(r"\(.*", "", cur)
        rows[cur] = {}
    elif cur: rows[cur][m.group(1).split(" ")[0]] = m.group(2)
print(f"{'kernel':70s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'scratch':>8s} {'LDS':>7s} {'occ':>4s}")
for k, v in rows.items():
    if flt and flt not in k: continue
    print(f"{k[:70]:70s} {v.get('VGPRs','?'):>5s} {v.get('AGPRs','?'):>5s} {v.get('TotalSGPRs','?'):>5s} {v.get('ScratchSize','?'):>8s} {v.get('LDS','?'):>7s} {v.get('Occupancy','?'):>4s}")
