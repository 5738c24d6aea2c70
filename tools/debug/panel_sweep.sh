#!/bin/bash
run() { env "$@" python tools/bench_kernels.py panel 2>&1 | grep -v amdgpu.ids; }
run VCLA_PANEL_P=0
run VCLA_PANEL_P=512
run VCLA_PANEL_P=512 VCLA_PANEL_MINT=4
run VCLA_PANEL_P=256
run VCLA_PANEL_P=1024 VCLA_PANEL_MINT=4
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "panel or skinny or splitk or ragged or fp8" -p no:cacheprovider 2>&1 | tail -5
