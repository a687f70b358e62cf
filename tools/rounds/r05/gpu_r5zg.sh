#!/bin/bash
out=gpurun_out/r5zg; mkdir -p $out
NS=243,249,256,262,268,275,281,288,294,300,307,313,320,326,332,339,345,352,358,364,371,377,384,390,396,403,409,416
timeout 1500 python tools/option_probe.py $NS "GROUP=1;GROUP=2;GROUP=3;GROUP=4" 5 > $out/groups_solo_a.log 2>&1; cat $out/groups_solo_a.log
timeout 1500 python tools/option_probe.py $NS "GROUP=4;GROUP=3;GROUP=2;GROUP=1" 5 > $out/groups_solo_b.log 2>&1; cat $out/groups_solo_b.log
