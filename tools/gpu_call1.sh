#!/bin/bash
# Round-4 measurement call 1 (run through gpurun): parity of the restructured front half, the clock experiment,
# phase traces of scatter / sort, knob A/B, FETCH_SIZE calibration.  Outputs under gpurun_out/c1/.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/c1; mkdir -p $O
timeout 420 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" | tee $O/pytest.rc
tail -3 $O/pytest.log
timeout 120 python tools/clock_experiment.py > $O/clock.jsonl 2> $O/clock.err; echo "clock rc $?"
LSR_LIB=build_variants/liblsr_trace.so LSR_TRACE_SCATTER=$O/tr_scatter.bin LSR_TRACE_SORT=$O/tr_sort.bin timeout 120 python tools/trace_binning.py > $O/trace_binning.txt 2>&1; echo "trace rc $?"
LSR_SORT_VARIANT=1 LSR_LIB=build_variants/liblsr_trace.so LSR_TRACE_SCATTER=$O/tr_scatter_v1.bin LSR_TRACE_SORT=$O/tr_sort_v1.bin timeout 120 python tools/trace_binning.py > $O/trace_binning_sort512.txt 2>&1
timeout 500 python tools/ab_knobs.py --rounds 2 '{"LSR_FOLD_SCAN":0}' '{"LSR_HOST_POLL":0}' '{"LSR_FOLD_SCAN":0,"LSR_HOST_POLL":0}' '{"LSR_SORT_VARIANT":1}' '{"LSR_SORT_LPT":0}' '{"LSR_SH_PLACEMENT":1}' '{"LSR_SH_PLACEMENT":2}' '{"LSR_FWD8_VARIANT":1}' '{"LSR_FWD8_VARIANT":2}' > $O/ab.jsonl 2> $O/ab.err; echo "ab rc $?"
timeout 120 ./tools/microbench/gather_fetch > $O/gather_plain.txt 2>&1
i=0
for SET in "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum" "TCC_REQ_sum TCC_READ_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $O/gf$i -o p -- ./tools/microbench/gather_fetch > $O/gf$i.log 2>&1 || echo "gf pass $i failed: $(tail -2 $O/gf$i.log)"
done
find $O -name "*counter_collection.csv" | head
cat $O/gather_plain.txt
