#!/bin/bash
# ribbon programs: "the casualties are the last rows" (no k_count_rows): parity of every ribbon test, C5 with the proof on / off
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-r03ab}
timeout 900 python -m pytest tests/test_gpu_scale.py tests/test_gpu_parity.py tests/test_reference_examples.py tests/test_scene_merge.py -m gpu -x -q -k "ribbon or c5 or worms or lightning or example or scene or merged" 2>&1 | tail -6
for i in 1 2 3; do for v in 1 0; do
  HNB_SUFFIX=$v timeout 300 python bench.py --no-cpu-baseline --pmc off --no-extra-configs --config c5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('HNB_SUFFIX=$v c5 ms/step %.4f min %.4f kernel %.4f lists %.4f init %.4f'%(d['ms_per_step'],d['windows']['min_ms_per_step'],d['stages']['update_ms_avg'],d['stages']['lists_ms_avg'],d['stages']['init_ms_avg']))"
done; done 2>&1 | tee gpurun_out/${T}_ab.log
