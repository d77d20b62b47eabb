cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -3
