#!/bin/bash
# round 5, GPU call 1: L2-intake probe + the configs[4] N = 1 leg on the round-4 dispatch (baseline before the fp8 M = 129-256 work)
mkdir -p gpurun_out
timeout 300 ./tools/l2_intake > gpurun_out/r05_l2_intake.txt 2>&1; echo "l2_intake rc=$?"
tail -5 gpurun_out/r05_l2_intake.txt
timeout 900 python bench.py --gpus 1 --fp8 --image-size 336 --global-batch 256 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r05_c4_gb256_before.json 2> gpurun_out/r05_c4_gb256_before.err; echo "bench rc=$?"
tail -c 1500 gpurun_out/r05_c4_gb256_before.json; tail -5 gpurun_out/r05_c4_gb256_before.err
