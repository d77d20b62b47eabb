#!/bin/bash
# Round profile on the GPU box (run through gpurun): bench JSON + rocprofv3 kernel stats + PMC.
# usage: tools/collect_profile.sh <tag> [commit]      outputs under gpurun_out/<tag>/
# (`commit` = the HEAD the snapshot was taken from — there is no .git on the box — stamped into the traffic file)
set -u
TAG=${1:-r01}
export LSR_PROFILE_COMMIT=${2:-unknown}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err || tail -3 $OUT/bench.err
cp gpurun_out/bench_full.json $OUT/bench_full.json 2>/dev/null   # the side file of THIS run (later bench invocations overwrite it)
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/stats -o bench -- python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-latency > $OUT/bench_under_rocprof.json 2> $OUT/rocprof.err || tail -3 $OUT/rocprof.err
bash tools/pmc_profile.sh $TAG --steps 3 --warmup 1 --clock-warmup 0 --no-cpu-baseline --no-latency > /dev/null 2>&1
python tools/rocpd_summary.py $OUT/stats/bench_results.db "$TAG ($LSR_PROFILE_COMMIT): python bench.py --steps 40 (16 views x 300k Gaussians, 256x256; fwd then fwd+bwd, decoder legs)" > $OUT/kernel_stats.md
python tools/pmc_summary.py gpurun_out/pmc_$TAG --traffic-json $OUT/traffic_render_forward.json > $OUT/pmc.md
# the headline workload ALONE (forward, 16 views x 300 k: one launch shape per kernel), so that per-kernel rates quoted in
# DESIGN.md (LDS bank conflicts, VALU busy) can be recomputed from a file under profiles/
bash tools/pmc_profile.sh ${TAG}_headline --steps 3 --warmup 1 --clock-warmup 0 --no-cpu-baseline --no-latency --no-bwd > /dev/null 2>&1
python tools/pmc_summary.py gpurun_out/pmc_${TAG}_headline > $OUT/pmc_headline.md
# the reference-shaped training step alone (configs[4]'s per-GPU batch: 4 scenes x 4 views, colour SH 4 + latent SH 2): the SH
# kernels' rates (VALU busy, HBM bytes) quoted in DESIGN.md come from this file
PMC_CMD="python tools/bench_decoder.py --scenes 4 --views 4 --steps 3" bash tools/pmc_profile.sh ${TAG}_cfg4 > /dev/null 2>&1
python tools/pmc_summary.py gpurun_out/pmc_${TAG}_cfg4 > $OUT/pmc_cfg4.md
# gpurun merges at most 64 MiB back: keep the summaries, drop the raw databases / traces they were made from
rm -rf $OUT/stats gpurun_out/pmc_$TAG gpurun_out/pmc_${TAG}_headline gpurun_out/pmc_${TAG}_cfg4
du -sh gpurun_out | tail -1
cat $OUT/bench.json
