#!/bin/bash
# Run on the GPU box (gpurun): kernel-trace stats + HBM traffic counters of the bench command, written under
# gpurun_out/r03_prof_<workload>_<precision>/.  Counters are collected in their own passes (FETCH_SIZE and WRITE_SIZE do not fit one pass;
# never together with a trace domain), after a plain run has stored the GEMM tile choices, so the profiled
# processes contain no tuning launches.      bash tools/collect_profiles.sh [workload] [precision]
set -e
WL=${1:-e2e16}
PREC=${2:-default}
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_prof_${WL}_${PREC}
mkdir -p $OUT
export ODIC_TILE_CACHE=$OUT/tile_cache.json
cd $GRAFT_REPO_ROOT
CMD="bench.py --workload $WL --steps 3 --warmup 1 --no-roofline --no-parity --no-fp32 --no-exact --no-cpu-baseline"
if [ "$PREC" != "default" ]; then CMD="$CMD --precision $PREC"; fi
python3 $CMD > $OUT/plain.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/$CMD > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $GRAFT_REPO_ROOT/$CMD > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $GRAFT_REPO_ROOT/$CMD > $OUT/write.log 2>&1
cd $GRAFT_REPO_ROOT
F=$(find $OUT/fetch -name "*counter_collection.csv" | head -1)
W=$(find $OUT/write -name "*counter_collection.csv" | head -1)
S=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
python3 tools/summarize_pmc.py $F $W > $OUT/pmc_hbm_traffic.txt
python3 tools/pmc_traffic_json.py $F $W $OUT/pmc_traffic_${WL}_${PREC}.json "$CMD (tile choices preloaded: no tuning launches)"
cut -c1-220 $S | head -60 > $OUT/kernel_stats.csv
rm -rf $OUT/fetch $OUT/write $OUT/trace/*/*kernel_trace.csv
echo collected
