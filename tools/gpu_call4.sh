#!/bin/bash
# Round-4 measurement call 4: parity with the fused projection + SH kernel, its A/B at configs[3]/[4], sort workgroup sizes on a
# scene whose longest list fits the 384-thread variant, cost classes of the folded scan.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/c4; mkdir -p $O
timeout 420 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" | tee $O/pytest.rc
tail -3 $O/pytest.log; grep -E "FAILED|Error|error" $O/pytest.log | head -20
timeout 500 python tools/ab_knobs.py --rounds 2 --workloads cfg3,cfg4 '{"LSR_FUSE_SH":0}' '{"LSR_FUSE_SH":0,"LSR_SH_PLACEMENT":0}' > $O/ab_sh.jsonl 2> $O/ab_sh.err; echo "ab sh rc $?"
timeout 300 python tools/ab_knobs.py --rounds 2 --workloads raster16 --gaussians 250000 '{"LSR_SORT_VARIANT":1}' > $O/ab_sort250.jsonl 2> $O/ab_sort250.err; echo "ab sort rc $?"
timeout 300 python tools/ab_knobs.py --rounds 3 --workloads raster16 '{"LSR_FOLD_SCAN":0}' > $O/ab_fold.jsonl 2> $O/ab_fold.err
for i in 1 2 3; do
  timeout 100 python tools/bench_stages.py head --no-decoder >> $O/bwd_ab.jsonl 2>/dev/null
  LSR_LIB=build_variants/liblsr_r03.so timeout 100 python tools/bench_stages.py r03 --no-decoder >> $O/bwd_ab.jsonl 2>/dev/null
done
cat $O/bwd_ab.jsonl
