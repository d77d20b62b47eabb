mkdir -p gpurun_out
for S in 611 612 613 614 615 616; do
timeout 400 python -m tests.fuzz_parity --seconds 240 --seed $S --out gpurun_out/fuzz_$S.json > gpurun_out/fuzz_$S.log 2>&1; tail -3 gpurun_out/fuzz_$S.log
done
