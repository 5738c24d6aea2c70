cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/profvis -o vis -- python $GRAFT_REPO_ROOT/tools/prof_vision.py 64 5 > $GRAFT_REPO_ROOT/gpurun_out/profvis.log 2>&1)
grep "vision stack" gpurun_out/profvis.log
f=$(find gpurun_out/profvis -name "*.db" | head -1)
python tools/prof_stats.py $f 22 | cut -c1-170 | tee gpurun_out/r02_vision_b64_kernel_stats.csv
rm -rf gpurun_out/profvis
