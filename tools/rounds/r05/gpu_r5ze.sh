#!/bin/bash
# round 5, call ze: one chain worker per elected compute unit (GDCA_MCU_SOLO)
out=gpurun_out/r5ze; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "schedule or inverse" > $out/pytest.log 2>&1; echo "pytest rc $?"; tail -2 $out/pytest.log
timeout 900 python tools/option_probe.py 64,100,128,160,200,250,290 "MCU_SOLO=0;MCU_SOLO=1,MCUS=12;MCU_SOLO=1,MCUS=16;MCU_SOLO=1,MCUS=24;MCU_SOLO=1,MCUS=32;MCU_SOLO=0,MCUS=16" 9 > $out/solo_single.log 2>&1; cat $out/solo_single.log
timeout 900 python tools/option_probe.py 300,358,400,500 "MCU_SOLO=0;MCU_SOLO=1,MCUS=16;MCU_SOLO=1,MCUS=24;MCU_SOLO=1,MCUS=32" 5 > $out/solo_multi.log 2>&1; cat $out/solo_multi.log
