"""Per kernel of a device assembly file: VGPRs, AGPRs, SGPRs, LDS bytes, scratch bytes, instruction count and a hash of the
instruction stream (labels and comments stripped) - to check that a change to the SURROUNDING source left a kernel's code alone
(the GEMM k-loops move by a percent when it does not).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Ineuronika_amd/csrc -S --cuda-device-only neuronika_amd/csrc/nk_gemm.hip -o /tmp/gemm.s
    python tools/isa_kernel_hashes.py /tmp/gemm.s | sort > after.txt ; diff before.txt after.txt
"""
import re, sys, hashlib, subprocess
txt = open(sys.argv[1]).read()
out = {}
for m in re.finditer(r'^(_Z\w+): *;[^\n]*\n(.*?)^\t\.size\t\1', txt, re.S | re.M):
    name, body = m.group(1), m.group(2)
    ins = [l.strip() for l in body.split('\n') if l.startswith('\t') and not l.strip().startswith('.') and not l.strip().startswith(';')]
    ins = [re.sub(r'\.LBB\d+_', '.LBB_', re.sub(r'\s*;.*$', '', l)) for l in ins]
    tail = txt[m.end():m.end() + 3000]
    def g(k):
        r = re.search(re.escape(name) + r'\.' + k + r', (\d+)', tail); return int(r.group(1)) if r else -1
    l = re.search(r'\.amdhsa_group_segment_fixed_size (\d+)', body)
    sc = re.search(r'; ScratchSize: (\d+)', tail)
    h = hashlib.md5('\n'.join(ins).encode()).hexdigest()[:10]
    out[name] = (g('num_vgpr'), g('num_agpr'), g('numbered_sgpr'), int(l.group(1)) if l else -1, int(sc.group(1)) if sc else -1, len(ins), h)
dem = subprocess.run(['c++filt'], input='\n'.join(out), capture_output=True, text=True).stdout.split('\n')
for (k, v), d in zip(out.items(), dem):
    print(d[:100].ljust(100), *v)
