#!/bin/bash
# Round 6: the batch driver beside hardware queues that come to life while sweeps run (tools/rounds/r06/gpu_r6q.sh found the standstills).
#   early_sets: every context -- and its hardware queue -- is made before any work (the driver's way since): no launch is disturbed
#   lazy_sets : GDCA_CLI_LAZY_SETS=1, the old way (the batches' contexts are made when the first small family shows up, beside running
#               sweeps): launches are disturbed, the sweep sees the hole in its polling, ends early and is run again -- no run may fail
out=gpurun_out/r6s; mkdir -p $out
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
rm -rf $D/ref; mkdir -p $D/ref
gaussdca.jl_amd/gdca_cli --batch $D/in --out $D/ref --parsers 4 --merge 1 > $out/ref.log 2>&1 || echo "reference run failed"
try() { # name, runs, env...
  name=$1; runs=$2; shift; shift
  bad=0; again=0; differ=0; slow=0
  for i in $(seq 1 $runs); do
    rm -rf $D/out; mkdir -p $D/out
    t0=$(date +%s.%N)
    env "$@" gaussdca.jl_amd/gdca_cli --batch $D/in --out $D/out --parsers 4 --merge 8 --merge-blocks 57 > $out/last.log 2>&1 || { bad=$((bad+1)); cp $out/last.log $out/fail_${name}_$i.log; }
    t1=$(date +%s.%N)
    grep -q "run again" $out/last.log && { again=$((again+1)); cp $out/last.log $out/again_${name}_$i.log; }
    python - $t0 $t1 <<'PY' || slow=$((slow+1))
import sys; sys.exit(0 if float(sys.argv[2]) - float(sys.argv[1]) < 1.0 else 1)
PY
  done
  echo "$name: $bad failures of $runs; runs with an inverse run again: $again; runs of a second and more: $slow"
}
try early_sets 30
try lazy_sets 200 GDCA_CLI_LAZY_SETS=1
grep -h "run again" $out/again_*.log | awk '{print $4, $5}' | sort | uniq -c | sort -rn | head
