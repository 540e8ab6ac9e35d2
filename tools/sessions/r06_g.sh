#!/bin/bash
# Round 6, session g: the round's profile set at the commit passed as $1 (tools/gpu_session_final.sh: parity suite, the driver's bench
# invocation, bench lines + rocprofv3 kernel stats of the four workloads, micro-benchmarks, layer shapes, traffic PMC), then the stall
# counters of the three Winograd kernels and the GEMM.
set -u
commit=${1:-unknown}
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out; mkdir -p $out/r06g
cd $root
bash tools/gpu_session_final.sh $commit r06 2>&1 | tail -40
PMC_GROUPS="a b c e" bash tools/pmc_profile.sh gpurun_out/r06g/pmc conv_fwd conv_bwd_input conv_bwd_kernel conv_s2_bwd_input gemm4k > $out/r06g/pmc_summary.txt 2>&1
find $out -name "*.db" -delete
