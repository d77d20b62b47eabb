#!/bin/bash
# Round-4 measurement call 2: binning phase traces, r03 library vs HEAD in the clock experiment (cold start), SQ counters of the
# front-half kernels, WRITE_SIZE calibration, the faster tile scan.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/c2; mkdir -p $O
LSR_LIB=build_variants/liblsr_trace.so LSR_TRACE_SCATTER=$O/tr_scatter.bin LSR_TRACE_SORT=$O/tr_sort.bin timeout 120 python tools/trace_binning.py > $O/trace_binning.txt 2>&1; echo "trace rc $?"
LSR_SORT_VARIANT=1 LSR_LIB=build_variants/liblsr_trace.so LSR_TRACE_SCATTER=$O/tr_scatter_v1.bin LSR_TRACE_SORT=$O/tr_sort_v1.bin timeout 120 python tools/trace_binning.py > $O/trace_binning_sort512.txt 2>&1
cat $O/trace_binning.txt | tail -30
timeout 120 python tools/clock_experiment.py --cold-only > $O/clock_head.jsonl 2> $O/clock_head.err; echo "clock head rc $?"
LSR_LIB=build_variants/liblsr_r03.so timeout 120 python tools/clock_experiment.py --cold-only > $O/clock_r03.jsonl 2> $O/clock_r03.err; echo "clock r03 rc $?"
timeout 300 python tools/ab_knobs.py --rounds 3 --workloads raster16 '{"LSR_FOLD_SCAN":0}' > $O/ab.jsonl 2> $O/ab.err; echo "ab rc $?"
PMC_CMD="python tools/bench_stages.py pmc --no-decoder" 
i=0
for SET in \
  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU" \
  "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VMEM_WR" \
  "SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_INSTS_GDS SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA" ; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $O/pmc$i -o p -- $PMC_CMD > $O/pmc$i.log 2>&1 || echo "pmc pass $i failed: $(tail -2 $O/pmc$i.log)"
done
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/gfw -o p -- ./tools/microbench/gather_fetch > $O/gfw.log 2>&1 || echo "write pass failed"
python tools/pmc_summary.py $O > $O/pmc.md 2>/dev/null; head -c 3000 $O/pmc.md
