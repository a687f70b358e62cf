#!/bin/bash
out=gpurun_out/r6k; mkdir -p $out
python -c "import torch" 2>/dev/null
for w in 0 1 0 1; do echo "== GDCA_WARM=$w"; GDCA_WARM=$w python tools/e2e_profile.py C 6 2>&1 | tail -5; done | tee $out/e2e_C.log
for w in 0 1; do echo "== GDCA_WARM=$w"; GDCA_WARM=$w python tools/e2e_profile.py D 4 2>&1 | tail -3; done | tee $out/e2e_D.log
