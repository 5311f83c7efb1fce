#!/bin/bash
# Instruction mix of one kernel of a solver library: tools/kernel_mix.sh <lib.so> <kernel-name-substring>
lib=$1; pat=$2
tmp=$(mktemp -d)
L=/opt/rocm/lib/llvm/bin
$L/llvm-objcopy --dump-section .hip_fatbin=$tmp/fat.bin $lib $tmp/stripped
t=$($L/clang-offload-bundler --list --type=o --input=$tmp/fat.bin | grep gfx950)
$L/clang-offload-bundler --unbundle --type=o --input=$tmp/fat.bin --targets=$t --output=$tmp/co.o
$L/llvm-objdump -d --no-show-raw-insn $tmp/co.o | awk -v pat="$pat" '
  /^[0-9a-f]+ <[^>]*>:/ { if ($2 !~ /^<L/) on = index($0, pat) > 0 }
  on && /^[ \t]+[a-z]/ { n[$1]++; tot++ }
  END { for (k in n) printf "%6d %s\n", n[k], k; printf "%6d TOTAL\n", tot }' | sort -rn | head -${3:-40}
rm -rf $tmp
