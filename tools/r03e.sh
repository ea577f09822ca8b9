#!/bin/bash
# round 3, fifth GPU call: 256-byte-unit slab offsets (100M-particle test), cooperative k_emit_events
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-r03e}
timeout 1800 python -m pytest tests -m gpu -x -q --durations=6 2>&1 | tail -22 > gpurun_out/${T}_pytest.log; cat gpurun_out/${T}_pytest.log
( time timeout 900 python bench.py --no-cpu-baseline > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err ) 2>&1 | grep real
python3 tools/bench_summary.py gpurun_out/${T}_bench.json | tee gpurun_out/${T}_bench_summary.txt
tail -c 300 gpurun_out/${T}_bench.err
