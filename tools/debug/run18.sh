cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests/test_gpu_model.py tests/test_gpu_dropin.py -x -q 2>&1 | tail -4
timeout 600 python tools/prof_vision.py 64 5 2>&1 | grep "vision stack"
timeout 600 python tools/prof_vision.py 1 20 2>&1 | grep "vision stack"
