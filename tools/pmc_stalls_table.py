"""tools/pmc_profile.sh's summary text (tools/rocpd_pmc_summary.py output) -> the per-kernel table of profiles/rNN_pmc_stalls.md:
clock = GRBM_GUI_ACTIVE / 8 / duration (GRBM counts per XCD-sum), MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (cycles x 1024 SIMDs),
other VALU / MFMA = (SQ_INSTS_VALU - SQ_INSTS_MFMA) / SQ_INSTS_MFMA (SQ_INSTS_VALU counts the MFMAs too), L2 hit = TCC_HIT / (HIT + MISS).
    python tools/pmc_stalls_table.py gpurun_out/r06g/pmc_summary.txt [kernel-name substring ...]"""
import re
import sys


def main():
    text = open(sys.argv[1]).read()
    want = sys.argv[2:] or ["wino_kernel", "wino_dw_kernel", "sgemm_kernel", "s2dx_kernel"]
    rows = {}
    for block in re.split(r"^## ", text, flags=re.M)[1:]:
        head, *lines = block.split("\n")
        m = re.match(r"(.*?)\s+\(dispatches: (\d+), avg duration under PMC ([\d.]+) us\)", head)
        if not m or not any(w in m.group(1) for w in want):
            continue
        c = {}
        for l in lines:
            q = re.match(r"\s+(\w+)\s+avg\s+([\d.]+)", l)
            if q:
                c[q.group(1)] = float(q.group(2))
        rows.setdefault(m.group(1), (float(m.group(3)), c))
    print("| kernel | us under PMC | clock GHz | MFMA busy | SQ_INSTS_MFMA (M) | other VALU / MFMA | SALU / MFMA | LDS inst / MFMA | vector loads / MFMA | LDS conflict / active | WAIT_INST_ANY / WAVE_CYCLES | waves | L2 hit |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    for name, (us, c) in rows.items():
        mf = c.get("SQ_INSTS_MFMA", 0.0)
        if mf <= 0:
            continue
        cycles = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        g = lambda k: c.get(k, 0.0)
        hit = g("TCC_HIT_sum") / max(1.0, g("TCC_HIT_sum") + g("TCC_MISS_sum"))
        print(f"| `{name}` | {us:.1f} | {cycles / us / 1e3:.3f} | {g('SQ_VALU_MFMA_BUSY_CYCLES') / max(1.0, cycles * 1024):.3f} | {mf / 1e6:.2f} | "
              f"{(g('SQ_INSTS_VALU') - mf) / mf:.2f} | {g('SQ_INSTS_SALU') / mf:.2f} | {g('SQ_INSTS_LDS') / mf:.3f} | {g('SQ_INSTS_VMEM_RD') / mf:.3f} | "
              f"{g('SQ_LDS_BANK_CONFLICT') / max(1.0, g('SQ_LDS_IDX_ACTIVE')):.3f} | {g('SQ_WAIT_INST_ANY') / max(1.0, g('SQ_WAVE_CYCLES')):.3f} | {g('SQ_WAVES'):.0f} | {hit:.3f} |")


if __name__ == "__main__":
    main()
