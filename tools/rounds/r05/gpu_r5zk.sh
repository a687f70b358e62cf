#!/bin/bash
out=gpurun_out/r5zk; mkdir -p $out
timeout 900 python tools/option_probe.py 64,128,200,300,500 "TALLY_TJ=0;TALLY_TJ=32;TALLY_TJ=0;TALLY_TJ=32" 7 ms_covariance 20000 > $out/tally_tj.log 2>&1; cat $out/tally_tj.log
timeout 900 python tools/option_probe.py 64,128,200,300,500 "HAMMING_MODE=auto;HAMMING_MODE=full;HAMMING_MODE=bound" 7 ms_weights 20000 > $out/ham_mode.log 2>&1; cat $out/ham_mode.log
