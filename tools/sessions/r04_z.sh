#!/bin/bash
# GPU session r04-z: streaming (nt) loads for the row-contiguous 128-row GEMM operand (variant benchmarks/_ab/ntA.so, built from a copy of csrc
# with nk_load_stream(.., true) in TileLoader<false, 128>::load): the dK / dV products of C5 (A read exactly once), TN 4096^3 and the C5
# weight-gradient shape (operands shared between blocks through L2)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do for v in main ntA; do lib=benchmarks/_ab/$v.so; [ $v = main ] && lib=neuronika_amd/lib/libneuronika_hip.so
echo "rep$rep $v dkdv: $(NK_GEMM_PAIR=0 NEURONIKA_HIP_LIB=$PWD/$lib timeout 60 python benchmarks/ab_pair.py 64 2>&1 | grep dkdv | cut -c1-140)"; done; done
