#!/bin/bash
out=gpurun_out/r5zd; mkdir -p $out
timeout 900 python tools/option_probe.py 300,307,313,326,339 "PANEL_HALVES=1;PANEL_HALVES=0;PANEL_HALVES=1;PANEL_HALVES=0" 9 > $out/halves_g2.log 2>&1; cat $out/halves_g2.log
timeout 900 python tools/option_probe.py 345,358,371,384,420 "PANEL_HALVES=1;PANEL_HALVES=0;PANEL_HALVES=1;PANEL_HALVES=0" 7 > $out/halves_g34.log 2>&1; cat $out/halves_g34.log
