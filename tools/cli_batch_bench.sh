#!/bin/bash
# gdca_cli --batch on the first F families of BASELINE.json's batch configuration, FASTA files in, ranking files out
# (the callers either side of the hot path):  bash tools/cli_batch_bench.sh [F] [scratch dir]
F=${1:-48}
D=${2:-/tmp/gdca_cli_batch}
rm -rf $D; mkdir -p $D/in $D/out
python - "$F" "$D" <<'PY'
import sys, subprocess, os
sys.path.insert(0, os.getcwd())
from importlib import import_module
batch = import_module("gaussdca.jl_amd.batch")
F, D = int(sys.argv[1]), sys.argv[2]
for f, (N, M) in enumerate(batch.batch_sizes(256)[:F]):
    subprocess.run(["gaussdca.jl_amd/gdca_cli", "--synth", str(N), str(M), str(0xE000 + f), "%s/in/fam%03d.fasta" % (D, f)], check=True,
                   stdout=subprocess.DEVNULL)
print("generated", F, "families")
PY
du -sh $D/in | cut -f1
for k in ${INFLIGHT:-1 2}; do
  ${CLI:-gaussdca.jl_amd/gdca_cli} --batch $D/in --out $D/out --gpus 1 --inflight $k 2>&1 | tail -3
done
ls $D/out | wc -l
