#!/bin/bash
# Round profile on the GPU box (run through gpurun): kernel durations (serialized and with the in-flight contexts),
# MFMA-pipe utilisation and HBM traffic per kernel.  Since round 6 bench.py's default run includes the full-resolution and 3-D
# aggregation legs, so the kernel-stats passes below hold every kernel of DESIGN.md 4 of the shipped binary.  Outputs under gpurun_out/prof_$1; summaries are then copied to
# profiles/ by hand.  rocprofv3 gets the program itself after `--` (no wrapper), counters in their own passes.
set -e
TAG=${1:-r06}
OUT=gpurun_out/prof_$TAG
rm -rf $OUT      # stale traces of an earlier call must not be picked up below
mkdir -p $OUT
# the plan every entry point shares: the committed configs/tuning/mi355x.json (measured here when absent)
export ST_TUNE_CACHE=$PWD/configs/tuning/mi355x.json
COMMON="--steps 20 --warmup 5 --no-cpu-baseline --no-test-step --sustain-seconds 0"
python bench.py $COMMON > $OUT/bench_plain.json 2> $OUT/bench_plain.err          # also fills the tuning cache
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/stats_inflight1 -- python3 $R/bench.py $COMMON --inflight 1 > $R/$OUT/bench_rocprof_inflight1.json 2> $R/$OUT/rocprof1.err
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/stats_inflight -- python3 $R/bench.py $COMMON > $R/$OUT/bench_rocprof_inflight.json 2> $R/$OUT/rocprof3.err
# counter passes: the headline loop only (no secondary legs) - except FETCH / WRITE, which also cover the stereo module's
# full-resolution and 3-D legs (cv_agg3d_kernel, vol_agg3d_kernel, softargmin_reg_kernel, feat_upsample_kernel)
PM="--steps 4 --warmup 2 --no-cpu-baseline --no-test-step --sustain-seconds 0 --inflight 1 --no-secondary-legs"
PMLEGS="--steps 4 --warmup 2 --no-cpu-baseline --no-test-step --sustain-seconds 0 --inflight 1"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d $R/$OUT/pmc_mfma -- python3 $R/bench.py $PM > /dev/null 2> $R/$OUT/pmc_mfma.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$OUT/pmc_fetch -- python3 $R/bench.py $PMLEGS > /dev/null 2> $R/$OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/$OUT/pmc_write -- python3 $R/bench.py $PMLEGS > /dev/null 2> $R/$OUT/pmc_write.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_SALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/$OUT/pmc_valu1 -- python3 $R/bench.py $PM > /dev/null 2> $R/$OUT/pmc_valu1.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $R/$OUT/pmc_valu2 -- python3 $R/bench.py $PM > /dev/null 2> $R/$OUT/pmc_valu2.err
cd $R
python tools/pmc_valu.py $(find $OUT/pmc_valu1 -name "*counter_collection.csv" | head -1) $(find $OUT/pmc_valu2 -name "*counter_collection.csv" | head -1) > $OUT/valu_mfma.txt
find $OUT -name "*_kernel_stats.csv" | head
python tools/pmc_mfma.py $(find $OUT/pmc_mfma -name "*counter_collection.csv" | head -1) $OUT/mfma_busy.json > $OUT/mfma_busy.txt
python tools/pmc_summary.py $(find $OUT/pmc_fetch -name "*counter_collection.csv" | head -1) $(find $OUT/pmc_write -name "*counter_collection.csv" | head -1) $OUT/hbm_traffic.json > $OUT/hbm_traffic.txt
python tools/op_profile.py --out $OUT/op_table.txt > /dev/null
cp configs/tuning/mi355x.json $OUT/tune_cache.json
# keep the merged-back payload small: the raw traces are not needed
find $OUT -name "*_kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*.db" -delete
du -sh $OUT
