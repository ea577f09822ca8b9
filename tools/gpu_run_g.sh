#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r02g_pytest.log; cat gpurun_out/r02g_pytest.log
for sk in 1 0 1 0; do
  HNB_SKIP_LISTS=$sk timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('skip=$sk c2 ms/step %.4f kernel %.4f frac %.3f lists %.4f | '%(d['ms_per_step'],d['roofline']['kernel_ms_avg'],d['roofline']['frac'],d['roofline']['lists_ms_avg']) + ' | '.join('%s step %.4f k %.4f frac %.3f lists %.4f'%(k,v['ms_per_step'],v['roofline']['kernel_ms_avg'],v['roofline']['frac'],v['roofline']['lists_ms_avg']) for k,v in d['configs'].items()))
" >> gpurun_out/r02g_ab.log
done
cat gpurun_out/r02g_ab.log
