"""Which basic blocks of a kernel's ISA (hipcc -S --cuda-device-only) hold both MFMAs and scratch (spill) traffic.
    python tools/isa_scratch_in_loops.py file.s [substring of mangled kernel name]"""
import re, sys
s = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r'^(_Z\w+):\n', s, re.M):
    name = m.group(1)
    if flt not in name or 'sgemm' not in name and not flt: continue
    j = s.find('.Lfunc_end', m.end())
    body = s[m.end():j]
    hot = []
    for b in re.split(r'\n(?=\.LBB\d+_\d+:)', body):
        nm, ns = len(re.findall(r'v_mfma', b)), len(re.findall(r'scratch_', b))
        if nm >= 16: hot.append((b.split('\n')[0].strip()[:12], nm, ns, len(re.findall(r'v_add_f32|v_pk_add_f32', b))))
    tot = len(re.findall(r'scratch_', body))
    print(name[:70], 'scratch ops total', tot, 'mfma blocks (label, mfma, scratch, vadd):', hot)
