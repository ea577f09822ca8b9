#!/bin/bash
# last check of the round on the final sources: GPU tests, smoke, the bench line with its PMC passes (traffic.json)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-r03zc}
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/${T}_pytest_gpu.log; cat gpurun_out/${T}_pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee gpurun_out/${T}_smoke.log
timeout 600 python bench.py --keep-pmc gpurun_out/${T}_pmc --write-traffic > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
cp profiles/traffic.json gpurun_out/${T}_traffic.json
python3 tools/bench_summary.py gpurun_out/${T}_bench.json | tee gpurun_out/${T}_bench_summary.txt
