#!/bin/bash
# kbench of every build/ab variant (and the shipped library), same box, one after another
cd "$(dirname "$0")/.."
for rep in 1 2; do
  timeout 120 python tools/kbench.py ${1:-C3} 2>/dev/null | tail -1
  for lib in build/ab/libprosstt_amd_*.so; do
    PROSSTT_AMD_LIB=$PWD/$lib timeout 120 python tools/kbench.py ${1:-C3} 2>/dev/null | tail -1
  done
done
