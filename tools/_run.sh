cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_latency_gpu.py -x -q -k "graph" 2>&1 | tail -15
