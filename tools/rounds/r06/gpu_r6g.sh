#!/bin/bash
# Round 6: gdca_cli --batch on the first 128 families of E (FASTA files in, ranking files out): the slots' pipeline against phase batches of everything
out=gpurun_out/r6g; mkdir -p $out
F=${1:-128}; D=/tmp/gdca_cli_batch
rm -rf $D; mkdir -p $D/in $D/out
python - "$F" "$D" <<'PY'
import sys, subprocess, os
sys.path.insert(0, os.getcwd())
from importlib import import_module
batch = import_module("gaussdca.jl_amd.batch")
F, D = int(sys.argv[1]), sys.argv[2]
from concurrent.futures import ThreadPoolExecutor
def mk(a):
    f, (N, M) = a
    subprocess.run(["gaussdca.jl_amd/gdca_cli", "--synth", str(N), str(M), str(0xE000 + f), "%s/in/fam%03d.fasta" % (D, f)], check=True, stdout=subprocess.DEVNULL)
with ThreadPoolExecutor(12) as ex:
    list(ex.map(mk, enumerate(batch.batch_sizes(256)[:F])))
print("generated", F, "families")
PY
du -sh $D/in | cut -f1
run() { echo "== gdca_cli --batch $*"; rm -rf $D/out; mkdir -p $D/out; gaussdca.jl_amd/gdca_cli --batch $D/in --out $D/out --gpus 1 "$@" 2>&1 | tail -3; }
{
run --merge 1 --inflight 3
run --merge 16 --merge-blocks 100000 --inflight 2
run --merge 8 --merge-blocks 100000 --inflight 2
run --merge 1 --inflight 3
run --merge 16 --merge-blocks 100000 --inflight 2
run --merge 32 --merge-blocks 100000 --inflight 2
} > $out/cli_batch.log 2>&1
cat $out/cli_batch.log
timeout 600 python bench.py --config E --families $F --pipeline 16 --phased --no-cpu-baseline --steps 1 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench.py device-resident phased16 on the same', d['config']['families_per_step'], 'families:', round(d['value'],2), 'families/s')" | tee -a $out/cli_batch.log
timeout 600 python bench.py --config E --families $F --pipeline 2 --no-cpu-baseline --steps 1 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench.py device-resident two streams on the same', d['config']['families_per_step'], 'families:', round(d['value'],2), 'families/s')" | tee -a $out/cli_batch.log
