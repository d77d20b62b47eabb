#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/c7; mkdir -p $O
timeout 420 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -2 $O/pytest.log
for v in 1 4 16; do timeout 100 python tools/bench_stages.py v$v --no-decoder --warm --views=$v >> $O/views.jsonl 2>/dev/null; done
cat $O/views.jsonl
timeout 400 python bench.py --steps 100 --no-cpu-baseline > $O/bench.json 2>$O/bench.err; python -c "
import json; d=json.load(open('gpurun_out/bench_full.json')); print(d['ms_per_step'], d['fwdbwd']['ms_per_step']); print(json.dumps(d['latency'])[:1400]); print(d['pipelined']); print(d['decoder_step']['forward'], d['decoder_step']['forward_backward'], d['decoder_step']['batch4']['forward'], d['decoder_step']['batch4']['forward_backward'])"
