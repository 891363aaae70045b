#!/bin/bash
# run the latency ubench in a loop in the background and sample the clocks
( for i in 1 2 3 4 5 6; do ./tools/ubench/lat > /dev/null; done ) &
BG=$!
sleep 1
for i in 1 2 3; do rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk\|fclk" | head -4; sleep 0.7; done
wait $BG
rocm-smi --showclocks 2>/dev/null | grep -i "sclk" | head -2
