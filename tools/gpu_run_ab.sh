#!/bin/bash
# Development tool (GPU box): A/B of two builds of the runtime library on one box. Usage: gpu_run_ab.sh <tag> <variant.so> [rounds]
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; TAG=$1; VAR=$2; N=${3:-3}
for i in $(seq 1 $N); do
 for lib in "" "$R/$VAR"; do
  HNB_LIB=$lib timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('lib=%-28s c2 ms/step %.4f kernel %.4f frac %.3f init %.4f | '%('$lib'[-28:] or 'current',d['ms_per_step'],d['roofline']['kernel_ms_avg'],d['roofline']['frac'],d['init']['kernel_ms']) + ' | '.join('%s step %.4f k %.4f frac %.3f init %.4f'%(k,v['ms_per_step'],v['roofline']['kernel_ms_avg'],v['roofline']['frac'],(v.get('init') or {}).get('kernel_ms',0)) for k,v in d['configs'].items()))
" >> gpurun_out/${TAG}_ab.log
 done
done
cat gpurun_out/${TAG}_ab.log
