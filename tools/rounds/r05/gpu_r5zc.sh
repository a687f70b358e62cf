#!/bin/bash
out=gpurun_out/r5zc; mkdir -p $out
timeout 900 python tools/option_probe.py 358,400,450,500,600 "REM_TAIL=-1;REM_TAIL=0;REM_TAIL=256;REM_TAIL=1024;REM_TAIL=2048" 7 > $out/rem_tail.log 2>&1; cat $out/rem_tail.log
timeout 900 python tools/option_probe.py 300,358,400,500,600 "PANEL_HALVES=-1;PANEL_HALVES=0;PANEL_HALVES=1;RAMP=0;RAMP=1" 7 > $out/halves_ramp.log 2>&1; cat $out/halves_ramp.log
timeout 900 python tools/option_probe.py 128,200,250,290 "RING=8;RING=2;RING=4;RAGGED=0;RAGGED=1" 7 > $out/ring.log 2>&1; cat $out/ring.log
