#!/bin/bash
out=gpurun_out/r5z; mkdir -p $out
timeout 900 python tools/option_probe.py 270,280,290,300,310 "GROUP=1;GROUP=2;GROUP=1;GROUP=2" 9 > $out/group_44_49.log 2>&1; cat $out/group_44_49.log
timeout 900 python tools/option_probe.py 340,350,358,365 "GROUP=2;GROUP=3;GROUP=4;GROUP=2;GROUP=3;GROUP=4" 9 > $out/group_54_58.log 2>&1; cat $out/group_54_58.log
