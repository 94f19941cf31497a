#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/s6; mkdir -p $O; cd $R
bash tools/pmc_kbench.sh C3 c3 > /dev/null 2>&1; grep heavy gpurun_out/pmc_c3/summary.txt | tee $O/pmc_C3_k3h.txt
rm -rf gpurun_out/pmc_c3/[abcd]
