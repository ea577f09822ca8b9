#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-r03ac}
timeout 900 python -m pytest tests/test_events.py tests/test_gpu_scale.py tests/test_reference_examples.py -m gpu -x -q -k "event or firework or worms or system" 2>&1 | tail -4
for i in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline --pmc off --no-extra-configs --config c2_events 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('c2_events ms/step %.4f min %.4f kernel %.4f lists %.4f init %.4f | %s'%(d['ms_per_step'],d['windows']['min_ms_per_step'],d['stages']['update_ms_avg'],d['stages']['lists_ms_avg'],d['stages']['init_ms_avg'], json.dumps(d.get('per_program', d.get('programs', '')))[:300]))"
done 2>&1 | tee gpurun_out/${T}_ab.log
