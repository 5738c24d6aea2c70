cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(for m in default bind spread node0 node0_bind; do python tools/debug/r04_cpu_gemv_probe.py $m 2>&1 | grep -v amdgpu.ids; done; lscpu | grep -E "Model name|Socket|NUMA|Thread|Core") | tee gpurun_out/r04_cpu_baseline_sweep.txt
