#!/bin/bash
out=gpurun_out/r5w; mkdir -p $out
timeout 600 bash tools/cli_small_bench.sh > $out/cli_small.log 2>&1; cat $out/cli_small.log | tail -8
D=/tmp/gdca_cli_batch
INFLIGHT="3" timeout 600 bash tools/cli_batch_bench.sh 128 > $out/cli_batch_128.log 2>&1; grep "steady" $out/cli_batch_128.log
for m in 4 8; do
  echo "E128 --inflight 3 --merge $m: $(gaussdca.jl_amd/gdca_cli --batch $D/in --out $D/out --gpus 1 --inflight 3 --merge $m 2>&1 | grep steady)"
  echo "E128 --inflight 2 --merge $m: $(gaussdca.jl_amd/gdca_cli --batch $D/in --out $D/out --gpus 1 --inflight 2 --merge $m 2>&1 | grep steady)"
done
