#!/bin/bash
# Samples shader clock and socket power while a command runs: is the kernel running at the boost clock, or is the chip power-limited?
# usage: tools/clock_probe.sh <out.txt> -- <command...>
out=$1; shift; shift
mkdir -p "$(dirname "$out")"
"$@" > "${out%.txt}.cmd.log" 2>&1 &
pid=$!
: > "$out"
while kill -0 $pid 2>/dev/null; do
    echo "t=$(date +%s.%N)" >> "$out"
    rocm-smi --showclocks --showpower --showuse 2>/dev/null | grep -E "sclk|mclk|fclk|Power|GPU use" >> "$out"
    sleep 0.25
done
wait $pid
tail -3 "${out%.txt}.cmd.log"
