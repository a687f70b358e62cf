#!/bin/bash
# Host-side feed rate of the batch driver, no GPU needed: gdca_cli --batch --parse-only over the first F families of
# BASELINE.json's batch configuration (plain FASTA and gzip) for a ladder of parser-thread counts.
#   bash tools/parse_bench.sh [F] [scratch dir] ["thread counts"]
F=${1:-48}
D=${2:-/tmp/gdca_parse_bench}
PS=${3:-"1 2 4 8 16 32 64"}
CLI=gaussdca.jl_amd/gdca_cli
rm -rf $D; mkdir -p $D/plain $D/gz
python - "$F" "$D" <<'PY' | xargs -P 16 -L 1 $CLI --synth
import sys, os
sys.path.insert(0, os.getcwd())
from importlib import import_module
batch = import_module("gaussdca.jl_amd.batch")
F, D = int(sys.argv[1]), sys.argv[2]
for f, (N, M) in enumerate(batch.batch_sizes(256)[:F]):
    for sub, ext in (("plain", ".fasta"), ("gz", ".fasta.gz")):
        print(N, M, 0xE000 + f, "%s/%s/fam%03d%s" % (D, sub, f, ext))
PY
echo "generated $F families; host threads: $(nproc)"
for sub in plain gz; do
  echo "== $sub: $(du -sh $D/$sub | cut -f1)"
  cat $D/$sub/* > /dev/null   # page cache warm: the files were just written, but be explicit about what is measured
  for p in $PS; do
    [ $p -le $((2 * $(nproc))) ] && $CLI --batch $D/$sub --parse-only --parsers $p --passes ${PASSES:-1} 2>&1 | tail -1
  done
done
rm -rf $D
