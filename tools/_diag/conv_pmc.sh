#!/bin/bash
# SQ counters of the split8 convolution launches: separate --pmc passes (kernel-trace only), CSVs under gpurun_out/pmc_conv/
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_conv
mkdir -p $OUT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU" "SQ_INST_CYCLES_SALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $set --kernel-trace -d $OUT/p$i -o p$i --output-format csv -- python3 $R/tools/_diag/conv_layers_run.py ${1:+$R/$1} > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
find $OUT -name "*counter_collection.csv" | head
