#!/bin/bash
out=gpurun_out/r6h; mkdir -p $out
timeout 300 tools/_bin/ubench_tile_feed > $out/ubench_tile_feed.log 2>&1; cat $out/ubench_tile_feed.log
timeout 300 tools/_bin/ubench_fp4 > $out/ubench_fp4.log 2>&1; cat $out/ubench_fp4.log
