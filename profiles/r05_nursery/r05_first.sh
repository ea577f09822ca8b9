#!/bin/bash
# Round 5, first GPU call: the whole GPU suite on the new build, then same-box A/Bs of the spawn nursery.
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
TAG=${1:-r05a}
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -rf -p no:cacheprovider 2>&1 | tail -60 > gpurun_out/${TAG}_pytest.log; tail -25 gpurun_out/${TAG}_pytest.log
for o in "" "spawn_nursery=0"; do HNB_CTX_OPTIONS=$o timeout 300 python tools/reburst_probe.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/${TAG}_reburst.log; done; cat gpurun_out/${TAG}_reburst.log
OPT=spawn_nursery=0 CONFIGS="c2_mixed c2_interop c2_events" timeout 900 bash tools/ab_option.sh gpurun_out/${TAG}_ab_nursery.log
