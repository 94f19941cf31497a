#!/bin/bash
# Copies what tools/profile_bench.sh <tag> left under gpurun_out/prof_<tag>/ into profiles/ (tracked).
TAG=${1:-r06}; cd "$(dirname "$0")/.."; O=gpurun_out/prof_$TAG
cp $O/summary.txt profiles/${TAG}_summary.txt
cp $O/bench_unprofiled.json profiles/${TAG}_bench_unprofiled.json
cp $O/bench_trace.json profiles/${TAG}_bench_under_rocprof.json
cp "$(ls -t $O/trace/*/*kernel_stats.csv | head -1)" profiles/${TAG}_kernel_stats.csv
for k in sq sq2 sq3 sq4 fetch write; do
  f="$(ls -t $O/pmc_$k/*/*counter_collection.csv | head -1)"
  (head -1 "$f"; grep "sample_counts" "$f") > profiles/${TAG}_pmc_$k.csv
done
ls -la profiles/
