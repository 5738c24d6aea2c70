cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
timeout 900 python bench.py --fp8 --image-size 336 --batch 32 --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r02_bench_336px_fp8_b32.json; cut -c1-300 gpurun_out/r02_bench_336px_fp8_b32.json
rm -rf gpurun_out/pv
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/pv -o t -- python $R/tools/prof_vision.py 64 5 2>&1 | grep "vision stack" | tee $R/gpurun_out/r02_vision_b64_by_grid.txt)
f=$(find gpurun_out/pv -name "*kernel_trace.csv" | head -1)
python tools/prof_by_grid.py $f 20 | tee -a gpurun_out/r02_vision_b64_by_grid.txt
rm -rf gpurun_out/pv
