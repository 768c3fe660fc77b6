#!/bin/bash
# Image-chunked stage execution (ODIC_SWIN_CHUNKS, engine.SwinEngine.stage_chunks): interleaved A/B of the bench step.
#   bash tools/chunk_sweep.sh [workload] "1" "2,1" "4,2" ...
WL=${1:-e2e16}; shift
OUT=gpurun_out/chunk_sweep_$WL.txt
: > $OUT
for rep in 1 2; do
  for c in "$@"; do
    v=$(ODIC_SWIN_CHUNKS=$c timeout -k 10 300 python bench.py --workload $WL --no-roofline --no-parity --no-fp32 --no-exact --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print(j['value'], j['ms_per_step'])") || exit 1
    echo "chunks=$c rep=$rep captions/s,ms = $v" | tee -a $OUT
  done
done
