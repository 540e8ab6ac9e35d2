#!/bin/bash
# GPU session D (round 3): C4 bench over the paced replica communicator (contention numbers), conv stand-in chunk sweep,
# streaming-kernel A/B against the round-2 library
set -u
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out; mkdir -p $out
cd $root
for g in 240 120; do
  NK_BENCH_REPLICAS=8 NK_REPLICA_CHANNELS=32 NK_REPLICA_GBPS=$g NK_BENCH_NO_SUBRECORDS=1 timeout -k 5 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline > $out/d_bench_replica_$g.json 2> $out/d_bench_replica_$g.err
done
timeout -k 5 300 python benchmarks/ab_conv_standin.py > $out/d_conv_standin.jsonl 2>&1
for i in 1 2; do
  NEURONIKA_HIP_LIB=$root/benchmarks/_ab/r02.so timeout -k 5 200 python benchmarks/microbench.py stream > $out/d_stream_r02_$i.jsonl 2>&1
  timeout -k 5 200 python benchmarks/microbench.py stream > $out/d_stream_new_$i.jsonl 2>&1
  NEURONIKA_HIP_LIB=$root/benchmarks/_ab/r02.so timeout -k 5 200 python benchmarks/microbench.py softmax > $out/d_softmax_r02_$i.jsonl 2>&1
  timeout -k 5 200 python benchmarks/microbench.py softmax > $out/d_softmax_new_$i.jsonl 2>&1
done
timeout -k 5 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_tape.py -m gpu -x -q -k "dropout or paced" > $out/d_pytest.log 2>&1; tail -2 $out/d_pytest.log
