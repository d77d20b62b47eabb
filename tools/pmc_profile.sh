#!/bin/bash
# Collect hardware counters for the bench workload on the GPU box (run through gpurun).
# Each --pmc set is its own rocprofv3 run (SQ: 8 slots, TCC: FETCH_SIZE uses 3, WRITE_SIZE 2);
# never combined with sys/hip/hsa tracing.  Output: gpurun_out/pmc_<tag>/pass*/...csv
# usage: tools/pmc_profile.sh <tag> [bench args...]      (PMC_CMD="python tools/bench_decoder.py" profiles another workload)
set -u
TAG=${1:-r01}; shift || true
ARGS=${@:---steps 3 --warmup 1 --no-cpu-baseline --no-latency}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
i=0
for SET in \
  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU" \
  "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_TRANS_F32 SQ_THREAD_CYCLES_VALU" \
  "FETCH_SIZE GRBM_GUI_ACTIVE" \
  "WRITE_SIZE GRBM_GUI_ACTIVE" \
  "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" ; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/pass$i -o p -- ${PMC_CMD:-python bench.py $ARGS} > $OUT/pass$i.log 2>&1 || echo "pass $i failed: $(tail -2 $OUT/pass$i.log)"
done
find $OUT -name "*.csv" | head -20
