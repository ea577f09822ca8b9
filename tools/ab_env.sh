#!/bin/bash
# A/B of an ENVIRONMENT setting of the HIP runtime over bench configurations on ONE box: default against `env $SET`.
# Usage: SET=HIP_FORCE_DEV_KERNARG=1 CONFIGS="c2_mixed c5" tools/ab_env.sh [out]
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
out=${1:-gpurun_out/ab_env.log}; : > $out
SET=${SET:-HIP_FORCE_DEV_KERNARG=1}
one() { # label env...
  label=$1; shift
  env "$@" timeout 300 python bench.py --config $CFG --steps 20 --no-parity --pmc off --no-cpu-baseline --no-scene --no-comm 2>gpurun_out/ab_env_err.log | tail -1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); f = json.load(open('gpurun_out/bench_full.json')); st = f['stages']
print('$CFG %-28s ms_per_step %.4f  min/med/max %s  init %.4f update %.4f lists %.4f' % ('$label', d['ms_per_step'], d['windows']['ms_per_step_min_median_max'], st['init_ms_avg'], st['update_ms_avg'], st['lists_ms_avg']))" >> $out 2>&1
}
for CFG in ${CONFIGS:-c2_mixed c5 c2_events c2}; do
  for rep in 1 2; do
    one default X=1
    one "$SET" $SET
  done
done
timeout 200 python tools/scene_bench.py 1 600 2>&1 | grep -v amdgpu.ids | tail -2 >> $out
env $SET timeout 200 python tools/scene_bench.py 1 600 2>&1 | grep -v amdgpu.ids | tail -2 >> $out
cat $out
