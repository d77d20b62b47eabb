#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/c6; mkdir -p $O
timeout 300 python -m pytest tests/test_surface_gpu.py tests/test_latency_gpu.py -m gpu -x -q 2>&1 | tail -4
for v in 1 4 16; do timeout 100 python tools/bench_stages.py v$v --no-decoder --warm --views=$v >> $O/views.jsonl 2>/dev/null; done
cat $O/views.jsonl
for i in 1 2; do
  timeout 200 python tools/ab_knobs.py --rounds 1 --workloads cfg3,cfg4 >> $O/psh.jsonl 2>/dev/null
  LSR_LIB=build_variants/liblsr_psh4.so timeout 200 python tools/ab_knobs.py --rounds 1 --workloads cfg3,cfg4 >> $O/psh4.jsonl 2>/dev/null
done
python - <<'PY'
import json
for fn in ("gpurun_out/c6/psh.jsonl", "gpurun_out/c6/psh4.jsonl"):
    for l in open(fn):
        r = json.loads(l); print(fn[-10:], r["workload"], r["fwd_ms"], r["fwdbwd_ms"], r["kernels_fwd"].get("preprocess"))
PY
timeout 300 python bench.py --steps 100 --no-cpu-baseline > $O/bench.json 2>$O/bench.err; python -c "
import json; d=json.load(open('gpurun_out/bench_full.json')); print(json.dumps(d['latency'])[:1500]); print(d['pipelined']); print(d['decoder_step']['forward'], d['decoder_step']['forward_backward'], d['decoder_step']['batch4']['forward'], d['decoder_step']['batch4']['forward_backward'])"
