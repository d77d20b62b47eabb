cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -3
cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r06f/stats -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-latency > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py gpurun_out/r06f/stats/b_results.db "x" 2>/dev/null | grep -i "order\|render_fwd<4, 12, true\|clear16\|render_bwd<4" ; rm -rf gpurun_out/r06f/stats
