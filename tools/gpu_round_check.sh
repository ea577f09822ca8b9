#!/bin/bash
# Development tool (GPU box): the end-of-round check in one gpurun call: GPU tests, smoke, the default bench line, and the
# profiles of the bench command (kernel stats + the two PMC passes -> profiles/traffic.json). Usage: gpu_round_check.sh <tag>
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/${TAG}_pytest.log; cat gpurun_out/${TAG}_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; tail -c 700 gpurun_out/${TAG}_bench.json
timeout 300 python bench.py --gpus 2 --backend gloo --force-device 0 --steps 10 > gpurun_out/${TAG}_bench_n2_dryrun.json 2> gpurun_out/${TAG}_bench_n2_dryrun.err; tail -c 400 gpurun_out/${TAG}_bench_n2_dryrun.json
timeout 300 python tools/scene_bench.py 1 600 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_scene.log; timeout 300 python tools/scene_bench.py 4 300 2>&1 | grep -v amdgpu.ids >> gpurun_out/${TAG}_scene.log; cat gpurun_out/${TAG}_scene.log
bash tools/prof_bench.sh $TAG 2>&1 | tail -26
