#!/bin/bash
# sample GPU clock/power while a command runs
( while true; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Average Graphics Package Power|Current Socket" | tr '\n' ' '; echo; sleep 0.25; done ) > gpurun_out/r2_clk_$1.log &
SP=$!
shift
"$@"
kill $SP
