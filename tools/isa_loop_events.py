"""Order of global loads / waits / MFMA groups / barriers inside the loops of a kernel's ISA (hipcc -S output).
Usage: python tools/isa_loop_events.py file.s <substring of the mangled kernel name> ...
A `wait vmcnt(0)` between two loads of one k-tile means the loads are serialised (conditional loads do that)."""
import re
import sys


def events(body):
    out = []
    mf = 0
    for l in body.split('\n'):
        t = l.strip().split()
        if not t or t[0].startswith(';'):
            continue
        op = t[0]
        if op.startswith('v_mfma'):
            mf += 1
            continue
        if mf and (op.startswith('global_') or op.startswith('s_waitcnt') or op.startswith('s_barrier') or op.startswith('ds_') or op.endswith(':')):
            out.append(f"mfma x{mf}")
            mf = 0
        if re.match(r'^\.LBB\d+_\d+:', op):
            out.append(op)
        elif op.startswith('global_load'):
            out.append('LOAD' + op[len('global_load'):])
        elif op.startswith('global_store'):
            out.append('STORE')
        elif op.startswith('s_waitcnt'):
            out.append('wait ' + ' '.join(t[1:]))
        elif op.startswith('s_barrier'):
            out.append('BARRIER')
        elif op.startswith('s_cbranch') or op.startswith('s_branch'):
            out.append('br->' + t[-1])
    if mf:
        out.append(f"mfma x{mf}")
    return out


def main():
    s = open(sys.argv[1]).read()
    for pat in sys.argv[2:]:
        for nm in re.findall(r'^(_Z\w*):', s, re.M):
            if pat not in nm:
                continue
            a = s.index('\n' + nm + ':')
            b = s.index('.Lfunc_end', a)
            ev = events(s[a:b])
            # compress runs
            print('==', nm)
            comp = []
            for e in ev:
                if comp and comp[-1][0] == e:
                    comp[-1][1] += 1
                else:
                    comp.append([e, 1])
            print(' | '.join(e if n == 1 else f"{e} x{n}" for e, n in comp))


main()
