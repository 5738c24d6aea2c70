cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() { echo "-- $1"; env $1 timeout 600 python bench.py --batch 64 --steps 3 --warmup 1 --no-cpu-baseline --steps-c4 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['images_per_sec'], d['breakdown_ms'])"; }
(run "VCLA_X=0"; run "VCLA_DS_QKV_SPLIT=0"; run "VCLA_DS_SPLITK=2"; run "VCLA_DS_SPLITK=8"; run "VCLA_X=0"
echo "== B=32 bf16 (strong-scaling leg at 8 GPUs)"; timeout 600 python bench.py --batch 32 --steps 3 --warmup 1 --no-cpu-baseline --steps-c4 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['images_per_sec'], d['breakdown_ms'])") | tee gpurun_out/r04_b64_ab.txt
