#!/bin/bash
out=gpurun_out/r5zj; mkdir -p $out
timeout 900 python tools/option_probe.py 288,300,313,326,332 "REM_TAIL=-1;REM_TAIL=128;REM_TAIL=256;REM_TAIL=512;PANEL_HALVES=0;MCUS=10;MCUS=14" 7 > $out/g2_knobs.log 2>&1; cat $out/g2_knobs.log
timeout 900 python tools/option_probe.py 339,352,371,384 "REM_TAIL=-1;REM_TAIL=128;REM_TAIL=256;REM_TAIL=1024;PANEL_HALVES=1;MCUS=12;MCUS=20" 7 > $out/g3_knobs.log 2>&1; cat $out/g3_knobs.log
