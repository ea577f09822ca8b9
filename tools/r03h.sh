#!/bin/bash
# round 3: death horizons (k_count_rows skips row chunks that cannot hold a casualty); A/B against HNB_HORIZON=0
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-r03h}
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/${T}_pytest.log; cat gpurun_out/${T}_pytest.log
for i in 1 2; do for hz in 1 0; do
  for cfg in c2 c2_mixed c2_events c2_dieoff; do
  HNB_HORIZON=$hz timeout 300 python bench.py --no-cpu-baseline --pmc off --no-extra-configs --config $cfg 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('horizon=$hz %-9s ms/step %.4f min %.4f kernel %.4f lists %.4f init %.4f burst-init %.4f | %s'%('$cfg',d['ms_per_step'],d['windows']['min_ms_per_step'],d['stages']['update_ms_avg'],d['stages']['lists_ms_avg'],d['stages']['init_ms_avg'],(d.get('init') or {}).get('kernel_ms',0), d['kernels'][-60:]))"
  done
done; done 2>&1 | tee gpurun_out/${T}_horizon_ab.log
( time timeout 900 python bench.py --no-cpu-baseline > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err ) 2>&1 | grep real
python3 tools/bench_summary.py gpurun_out/${T}_bench.json | tee gpurun_out/${T}_bench_summary.txt
