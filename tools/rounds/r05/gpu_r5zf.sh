#!/bin/bash
out=gpurun_out/r5zf; mkdir -p $out
NS=300,307,313,326,339,345,352,358,364,371,377,384,396,409,422,435,448,460,473,486
timeout 1500 python tools/option_probe.py $NS "MCU_SOLO=0;MCU_SOLO=1,MCUS=12;MCU_SOLO=1,MCUS=16;MCU_SOLO=1,MCUS=20;MCU_SOLO=1,MCUS=10" 5 > $out/solo_a.log 2>&1; cat $out/solo_a.log
timeout 1500 python tools/option_probe.py $NS "MCU_SOLO=1,MCUS=16;MCU_SOLO=0;MCU_SOLO=1,MCUS=12;MCU_SOLO=1,MCUS=8" 5 > $out/solo_b.log 2>&1; cat $out/solo_b.log
