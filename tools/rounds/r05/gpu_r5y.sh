#!/bin/bash
out=gpurun_out/r5y; mkdir -p $out
timeout 900 python tools/option_probe.py 470,500,530,560,600,650,700 "MCUS=4;MCUS=6;MCUS=8;MCUS=10;MCUS=12" 9 > $out/mcus_big.log 2>&1; cat $out/mcus_big.log
timeout 900 python tools/option_probe.py 500 "MCUS=6;MCUS=10;MCUS=6;MCUS=10;MCUS=8;MCUS=8" 15 > $out/mcus_C.log 2>&1; cat $out/mcus_C.log
