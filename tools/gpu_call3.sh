#!/bin/bash
# Round-4 measurement call 3: parity with the persistent sort, sort variants, SH full-line stores, r03 vs HEAD backward.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/c3; mkdir -p $O
timeout 420 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" | tee $O/pytest.rc
tail -3 $O/pytest.log
timeout 500 python tools/ab_knobs.py --rounds 2 --workloads raster16 '{"LSR_SORT_PERSIST":0}' '{"LSR_SORT_PERSIST":2}' \
   '{"LSR_SORT_PERSIST":0,"LSR_SORT_TIER1":3072}' '{"LSR_SORT_PERSIST":1,"LSR_SORT_TIER1":3072}' '{"LSR_SORT_PERSIST":2,"LSR_SORT_TIER1":3072}' \
   '{"LSR_SORT_PERSIST":0,"LSR_SORT_TIER1":2048}' '{"LSR_SORT_PERSIST":1,"LSR_SORT_TIER1":2048}' '{"LSR_SORT_PERSIST":2,"LSR_SORT_TIER1":2048}' > $O/ab_sort.jsonl 2> $O/ab_sort.err; echo "ab sort rc $?"
timeout 500 python tools/ab_knobs.py --rounds 2 --workloads cfg3,cfg4 '{"LSR_SH_FULL_LINE":1}' '{"LSR_SH_PLACEMENT":1}' '{"LSR_SH_PLACEMENT":1,"LSR_SH_FULL_LINE":1}' '{"LSR_SH_PLACEMENT":2,"LSR_SH_FULL_LINE":1}' '{"LSR_SORT_PERSIST":0}' > $O/ab_sh.jsonl 2> $O/ab_sh.err; echo "ab sh rc $?"
for i in 1 2 3; do
  timeout 100 python tools/bench_stages.py head --no-decoder >> $O/bwd_ab.jsonl 2>/dev/null
  LSR_LIB=build_variants/liblsr_r03.so timeout 100 python tools/bench_stages.py r03 --no-decoder >> $O/bwd_ab.jsonl 2>/dev/null
done
cat $O/bwd_ab.jsonl
