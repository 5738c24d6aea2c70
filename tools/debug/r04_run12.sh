cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_env_switches.py -m gpu -q -x -p no:cacheprovider -k macro 2>&1 | tail -30
