#!/bin/bash
# SQ activity counters of the fused CSP tail kernel alone (tools/tail_bench.py), summarised by tools/pmc_sq.py
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/tail_pmc
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/sq -- python3 $R/tools/tail_bench.py > $OUT/bench_sq.txt 2> $OUT/sq.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT/mf -- python3 $R/tools/tail_bench.py > $OUT/bench_mf.txt 2> $OUT/mf.err
cd $R
python tools/pmc_sq.py $(find $OUT/sq -name "*counter_collection.csv" | head -1) > $OUT/sq_activity.txt
python tools/pmc_mfma.py $(find $OUT/mf -name "*counter_collection.csv" | head -1) $OUT/mfma_busy.json > $OUT/mfma_busy.txt || true
find $OUT -name "*_kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*.db" -delete
cat $OUT/sq_activity.txt; cat $OUT/mfma_busy.txt | head -12
