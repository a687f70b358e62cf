#!/bin/bash
# Round 6: the standstill reproduced -- what the batch driver did until then (the batches' contexts, hence their hardware queues, come to
# life beside running sweeps: GDCA_CLI_LAZY_SETS=1), a launch ended by the watchdog is final (GDCA_SWEEP_RETRIES=0), short bound.
# Measured with the round's first tree: 1 to 8 failing runs of 100 ("SPD inverse aborted: a dependency wait inside the sweep kernel timed out").
# With the sweep's hole detection the failing launches end after ~0.1 s instead of 1.5 s; with retries (the default) no run fails: gpu_r6s.sh.
out=gpurun_out/r6q; mkdir -p $out
python -c "import torch" 2>/dev/null
D=/tmp/gdca_cli_mix; rm -rf $D; mkdir -p $D/in
python - "$D" <<'PY'
import sys, numpy as np
sys.path.insert(0, '.')
from gaussdca.jl_amd import synth
D = sys.argv[1]
rng = np.random.default_rng(8)
sizes = [(min(int(n), 281), int(m)) for n, m in zip(rng.integers(6, 288, size=20), rng.integers(300, 4000, size=20))] + [(420, 3000), (380, 2500)]
for f, (N, M) in enumerate(sizes):
    synth.write_fasta("%s/in/fam%03d.fasta" % (D, f), synth.synth_family(N, M, 21, 0xABC0 + f))
PY
bad=0
for i in $(seq 1 100); do
  rm -rf $D/out; mkdir -p $D/out
  GDCA_CLI_LAZY_SETS=1 GDCA_SWEEP_RETRIES=0 GDCA_SWEEP_TIMEOUT_MS=1500 gaussdca.jl_amd/gdca_cli --batch $D/in --out $D/out --parsers 4 --merge 8 --merge-blocks 57 > $out/last.log 2>&1 || { bad=$((bad+1)); cp $out/last.log $out/fail_$i.log; }
done
echo "lazy sets, no second attempt: $bad failures of 100"
