#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-r03aa}
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py tests/test_events.py -m gpu -x -q 2>&1 | tail -4
bash tools/r03s.sh $T "c2_mixed c2_events c2_dieoff c5"
