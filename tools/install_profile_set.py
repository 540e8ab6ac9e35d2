"""Copy a round's closing profile set from gpurun_out/ (written by tools/gpu_session_final.sh + tools/sessions/rNN_g.sh on the GPU box)
into profiles/ (tracked): bench lines, rocprofv3 kernel stats, traffic PMC (-> profiles/roofline_traffic.json, what bench.py's
`roofline.traffic` reads), tolerance margins, the PMC stall table.      python tools/install_profile_set.py r06 [pmc session dir]"""
import glob
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
pmc = sys.argv[2] if len(sys.argv) > 2 else f"gpurun_out/{tag}g"
out, prof = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
for f in glob.glob(os.path.join(out, f"{tag}z", f"{tag}_*")):
    if not f.endswith(".err"):
        shutil.copy(f, prof)
for name in (f"{tag}_bench_driver_invocation.json", f"{tag}_traffic_pmc.md", f"{tag}_tolerance_margins.json"):
    shutil.copy(os.path.join(out, name), prof)
shutil.copy(os.path.join(out, f"{tag}_roofline_traffic.json"), os.path.join(prof, "roofline_traffic.json"))
summary = os.path.join(ROOT, pmc, "pmc_summary.txt")
if os.path.exists(summary):
    shutil.copy(summary, os.path.join(prof, f"{tag}_pmc_summary.txt"))
    table = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_stalls_table.py"), summary], capture_output=True, text=True).stdout
    md = os.path.join(prof, f"{tag}_pmc_stalls.md")
    if os.path.exists(md):  # replace the table, keep the prose
        text = open(md).read()
        text = re.sub(r"\| kernel \| us under PMC [^\n]*\n(\|[^\n]*\n)+", table, text, count=1)
        open(md, "w").write(text)
print("installed", tag, "from", out)
