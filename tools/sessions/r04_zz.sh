#!/bin/bash
# GPU session r04-zz: the out-projection's weight-gradient GEMM BETWEEN the attention backward kernel and its dK / dV products (a variant library
# whose nk_attention_bwd leaves the two products to the caller: benchmarks/_ab/nodkdv.so), benchmarks/c5_order.py with C5_ORDER_SPLIT=1
cd ${GRAFT_REPO_ROOT:-/root/repo}
C5_ORDER_SPLIT=1 NEURONIKA_HIP_LIB=$PWD/benchmarks/_ab/nodkdv.so timeout 240 python benchmarks/c5_order.py 2>&1 | tail -2 | tee gpurun_out/zz_order.json
