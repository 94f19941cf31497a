#!/bin/bash
# Builds timing-only variants of the library (K3_ABLATE=n) next to the real one.  Outputs of
# these builds are wrong by construction; only their kernel time is read.
cd "$(dirname "$0")/.."
mkdir -p build/ab
for n in ${ABLATIONS:-1 2 3 4 5}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -fvisibility=hidden \
     -DK3_ABLATE=$n -o build/ab/libprosstt_amd_ab$n.so prosstt_amd/csrc/prosstt_amd.hip || exit 1
done
