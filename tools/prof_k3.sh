#!/bin/bash
# Profiles the count sampler on the GPU box: kernel trace + two PMC passes.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof
mkdir -p $O
cd $R
./tools/microbench > $O/microbench.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 tools/quickbench.py > $O/trace.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU --output-format csv -d $O/pmc1 -- python3 tools/quickbench.py > $O/pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc2 -- python3 tools/quickbench.py > $O/pmc2.log 2>&1
find $O -name '*.csv' | head -30
