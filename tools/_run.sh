cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r06f
show() { python - "$1" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    r=json.loads(l)
    print(r['round'], r['workload'], r['knobs'], 'fwd', r['fwd_ms'], 'fb', r['fwdbwd_ms'], 'k_bwd', r['kernels_fwdbwd'].get('render_backward'), 'k_fwd(rec)', r['kernels_fwdbwd'].get('render_forward'))
PY
}
timeout 900 python tools/ab_knobs.py --rounds 3 --workloads raster16,cfg4 '{"LSR_REORDER":0}' > gpurun_out/r06f/ab.jsonl 2> gpurun_out/r06f/ab.err; show gpurun_out/r06f/ab.jsonl
