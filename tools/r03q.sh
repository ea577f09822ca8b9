#!/bin/bash
# rounds per init workgroup pass: 4 / 8 / 16
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-r03q}
for i in 1 2; do for v in 4 8 16; do
  case $v in 4) unset HNB_LIB HNB_JIT_EXTRA;; *) export HNB_LIB=$R/bevy_hanabi_amd/libhanabi_amd_r$v.so HNB_JIT_EXTRA=-DHNB_INIT_ROUNDS=$v;; esac
  for cfg in c2 c3 c4; do
  timeout 300 python bench.py --no-cpu-baseline --pmc off --no-extra-configs --config $cfg 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('rounds=$v %-9s ms/step %.4f kernel %.4f burst-init %.4f'%('$cfg',d['ms_per_step'],d['stages']['update_ms_avg'],(d.get('init') or {}).get('kernel_ms',0)))"
  done
done; done 2>&1 | tee gpurun_out/${T}_rounds.log
