"""Compact register / scratch / LDS table of every kernel in one .hip file (hipcc -Rpass-analysis=kernel-resource-usage).
    python tools/kernel_resources.py neuronika_amd/csrc/nk_gemm.hip [filter] [-- extra hipcc flags]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
extra = []
if "--" in args:
    i = args.index("--"); extra = args[i + 1:]; args = args[:i]
src = args[0]; flt = args[1] if len(args) > 1 else ""
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"),
       "-I" + os.path.join(ROOT, "neuronika_amd/csrc"), "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage", *extra]
r = subprocess.run(cmd, capture_output=True, text=True)
cur = {}
rows = []
for line in r.stderr.splitlines():
    m = re.search(r"remark: [^:]*:\d+:\d+:\s+(\w[\w ]*): (\S+)", line) or re.search(r":\s+(Function Name|Name): (\S+)", line)
    m = re.search(r"(Function Name|Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
    if not m: continue
    k, v = m.group(1), m.group(2)
    if k in ("Function Name", "Name"):
        cur = {"name": v}; rows.append(cur)
    else:
        cur[k.split(" ")[0]] = v
for row in rows:
    name = subprocess.run(["c++filt", row["name"]], capture_output=True, text=True).stdout.strip()
    if flt and flt not in name: continue
    print(f"{name[:90]:90s} vgpr {row.get('VGPRs','?'):>4s} agpr {row.get('AGPRs','?'):>3s} sgpr {row.get('TotalSGPRs','?'):>4s} scratch {row.get('ScratchSize','?'):>4s} occ {row.get('Occupancy','?'):>2s} lds {row.get('LDS','?'):>6s}")
if r.returncode != 0:
    sys.stderr.write(r.stderr[-3000:])
