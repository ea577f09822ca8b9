#!/bin/bash
# bench line only, on the final kernel sources: traffic.json + bench json
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-r03zb}
timeout 900 python bench.py --keep-pmc gpurun_out/${T}_pmc --write-traffic > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
cp profiles/traffic.json gpurun_out/${T}_traffic.json
python3 tools/bench_summary.py gpurun_out/${T}_bench.json | tee gpurun_out/${T}_bench_summary.txt
timeout 600 python -m pytest tests/test_events.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
