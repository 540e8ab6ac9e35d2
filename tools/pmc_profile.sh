#!/bin/bash
# PMC passes over benchmarks/pmc_targets.py (one rocprofv3 run per target and counter group; --pmc is never combined with
# the sys / hip / hsa traces).  Usage on the GPU box:   bash tools/pmc_profile.sh OUTDIR target [target...]
# (every rocprofv3 run sits under its own `timeout`: a counter group the tool aborts on must not hang the box; TA_* / TD_* groups
# did exactly that on this image and are not used)
# then python tools/rocpd_pmc_summary.py OUTDIR/<target>/*/*_results.db
set -u
out=$1; shift
cd /tmp && export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-/root/repo}
declare -A G
G[a]="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE"
G[b]="SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_IFETCH SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT"
G[c]="TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"
G[d]="TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum"
G[e]="SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAVES"
groups=${PMC_GROUPS:-"a b c d"}
mkdir -p "$root/$out"
for tgt in "$@"; do
  for g in $groups; do
    timeout -k 5 90 rocprofv3 --pmc ${G[$g]} --kernel-trace -d "$root/$out/$tgt/$g" -o r -- python "$root/benchmarks/pmc_targets.py" $tgt 3 > "$root/$out/${tgt}_$g.log" 2>&1 || echo "FAILED $tgt $g" 
  done
  echo "=== $tgt"
  python "$root/tools/rocpd_pmc_summary.py" $(find "$root/$out/$tgt" -name "*_results.db") 2>&1 | grep -v "^## .*\(fill\|copy\|pad_fwd\|rand\)" 
done
