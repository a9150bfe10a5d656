#!/bin/bash
# Profiles of one round on the GPU box: rocprofv3 kernel statistics of the driver's bench command, then PMC passes
# (one counter group per pass: FETCH_SIZE and WRITE_SIZE do not fit together, MI355X_MICROARCH.md "rocprofv3 PMC slots";
# --pmc only with --kernel-trace, never with system / runtime tracing).  Run from the repository root:
#     HELLO_PROFILE_COMMIT=<git rev-parse --short HEAD, expanded where .git exists> bash tools/profile_round.sh r05
# hbm_traffic.json records the sha256 of the library that was profiled (+ that commit): bench.py reports roofline.traffic only
# for the same library and flags traffic_stale otherwise.
# Writes raw output under gpurun_out/prof_<tag>/ and the summaries to copy into profiles/ under gpurun_out/profiles_<tag>/.
set -e
TAG=${1:-r05}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
SUM=$ROOT/gpurun_out/profiles_$TAG
rm -rf $OUT $SUM
mkdir -p $OUT $SUM
export TMPDIR=/tmp
# 1. kernel statistics of the command the driver runs, with the secondary legs (latency, 256-site launches, ...)
#    switched off so that every readconv_kernel launch in the statistics is a headline launch of --sites sites:
#    its average must agree with roofline.launch_ms of the JSON line printed by the same process
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-secondary > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $SUM/${TAG}_rocprofv3_kernel_stats_bench.csv
cp $OUT/bench_under_rocprof.json $SUM/${TAG}_bench_under_rocprofv3.json
echo "stats done"
# 2. PMC passes on a shorter run of the same workload (headline loop only)
SHORT="bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary"
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_COEXEC_CYCLES"; do
  name=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc_$name -- python3 $SHORT > $OUT/pmc_$name.json 2> $OUT/pmc_$name.err || echo "pass $name failed"
  f=$(find $OUT/pmc_$name -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 tools/summarize_pmc.py $f > $SUM/${TAG}_pmc_$name.txt
  echo "pmc $name done"
done
python3 tools/summarize_pmc.py --traffic $SUM > $SUM/hbm_traffic.json || true
# 3. one row per kernel of the forward: (5 + 20) steps x 10 launches = 250 forwards in the statistics run
python3 tools/summarize_pmc.py --table $SUM $SUM/${TAG}_rocprofv3_kernel_stats_bench.csv 250 > $SUM/${TAG}_kernel_table.md || true
# 4. the 2x-channel ("wide") hybrid model: readconv_wide_kernel and its layer-by-layer allele stage (statistics + SQ pass)
WIDE="tools/config_sweep.py --only wide --sites 4096 --steps 6"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_wide -- python3 $WIDE > $OUT/wide_stats.txt 2> $OUT/wide_stats.err || echo "wide stats failed"
f=$(find $OUT/stats_wide -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $SUM/${TAG}_rocprofv3_kernel_stats_wide.csv
mkdir -p $SUM/wide
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  name=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc_wide_$name -- python3 $WIDE > $OUT/pmc_wide_$name.txt 2> $OUT/pmc_wide_$name.err || echo "wide pass $name failed"
  f=$(find $OUT/pmc_wide_$name -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 tools/summarize_pmc.py $f > $SUM/wide/${TAG}_wide_pmc_$name.txt
  echo "wide pmc $name done"
done
# config_sweep runs 2 warm-up + 6 timed forwards of the configuration
python3 tools/summarize_pmc.py --table $SUM/wide $SUM/${TAG}_rocprofv3_kernel_stats_wide.csv 8 > $SUM/${TAG}_kernel_table_wide.md || true
ls -la $SUM
