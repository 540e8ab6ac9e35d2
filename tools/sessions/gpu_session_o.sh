#!/bin/bash
# GPU session O (round 3): counters of the conv input gradient AS THE C3 STEP LAUNCHES IT (unpadded columns) next to the form on the padded tensor
set -u
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out; mkdir -p $out
cd $root
PMC_GROUPS="a b e" bash tools/pmc_profile.sh gpurun_out/o_pmc conv_bwd_input conv_bwd_input_on_padded conv_fwd conv_bwd_kernel > $out/o_pmc.txt 2>&1
find $out/o_pmc -name "*.db" -delete
grep -E "^## |SQ_INSTS_MFMA|SQ_INSTS_VALU|SQ_VALU_MFMA_BUSY|GRBM_GUI|SQ_INSTS_SALU|SQ_INSTS_LDS|SQ_INSTS_VMEM_RD|SQ_WAVE_CYCLES|SQ_WAIT_INST_ANY|SQ_WAVES" $out/o_pmc.txt | grep -v "fill\|rand\|pad_"
