cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(for b in 64 16 256; do VCLA_LIB=$GRAFT_REPO_ROOT/tools/libvcla_vit_timeline.so python tools/debug/vit_attn_timeline.py $b 2>&1 | grep -v amdgpu.ids; done) | tee gpurun_out/r04_vit_attn_timeline.txt
