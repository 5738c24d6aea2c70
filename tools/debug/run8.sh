cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 2400 python -m pytest tests/test_gpu_model.py -x -q -k "7b_prefill or same or left_padded or no_worse" 2>&1 | tail -15 ) 2>&1
( time timeout 2400 python -m pytest tests/test_gpu_dropin.py tests/test_gpu_sampling.py -x -q 2>&1 | tail -15 ) 2>&1
grep -n "same-dtype\|7B B=\|left-padded\|7B forward" gpurun_out/parity_report.txt | cut -c1-250
