cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/parity_report.txt
echo "== tests: loss kernel, edge cases, fp8 modes"
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -q -x -p no:cacheprovider -k "causal_lm_loss or edge or fp8_modes or fp8_mfma_prefill or fp8_kv" 2>&1 | tail -5
grep -n "fp8 mode\|fp8 MFMA prefill\|fp8 K/V" gpurun_out/parity_report.txt | cut -c1-250
cp gpurun_out/parity_report.txt gpurun_out/r04f_parity_report.txt
echo "== bench default (HF cpu baseline, config4 both modes)"
nproc; free -g | head -2
timeout 1500 python bench.py --steps 5 --warmup 2 2>&1 | tail -1 > gpurun_out/r04f_bench.json
python - <<'P'
import json
d=json.loads(open('gpurun_out/r04f_bench.json').read())
print('value',d['value'],'img/s',d['images_per_sec'],d['breakdown_ms'])
print('config2',d['config2']['tokens_per_sec'],d['config2']['breakdown_ms'])
c=d['config4']; print('config4',c['mode'],c['tokens_per_sec'],c['images_per_sec'],c['breakdown_ms'],c['accuracy'])
w=c['w8a8']; print('   w8a8',w['tokens_per_sec'],w['images_per_sec'],w['breakdown_ms'],w['accuracy'])
print('cpu',json.dumps(d['cpu_baseline'])[:1500])
print('roofline',json.dumps(d['roofline'])[:800])
P
