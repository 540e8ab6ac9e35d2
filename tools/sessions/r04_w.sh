cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do for v in main bwdin_la1; do lib=benchmarks/_ab/$v.so; [ $v = main ] && lib=neuronika_amd/lib/libneuronika_hip.so; echo "rep$rep $v $(AB_N=1024 NEURONIKA_HIP_LIB=$PWD/$lib timeout 120 python benchmarks/ab_gemm.py 2>&1 | tail -1)"; done; done
