#!/bin/bash
# round 6: where / how the flat path stores the ages it keeps current (HNB_AGE_STORE_MODE: bit 0 behind the loop, bit 1 nontemporal), c2 and c4
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
out=gpurun_out/r06d_ab_age_store.log
: > $out
export HNB_JIT_CACHE=$GRAFT_REPO_ROOT/bevy_hanabi_amd/jit_cache
for round in 1 2; do
  for m in 0 1 2 3; do
    for cfg in c2 c4; do
      r=$(HNB_LIB=$GRAFT_REPO_ROOT/tools/variants/libhanabi_age$m.so timeout 600 python bench.py --config $cfg --no-parity --pmc off --no-cpu-baseline --no-extra-configs --no-scene --no-comm --windows 15 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms_avg'], d['windows']['ms_per_step_min_median_max'])")
      echo "round $round mode=$m $cfg: $r" | tee -a $out
    done
  done
  for cfg in c2_lean; do
    r=$(HNB_LIB=$GRAFT_REPO_ROOT/tools/variants/libhanabi_age0.so timeout 600 python bench.py --config $cfg --no-parity --pmc off --no-cpu-baseline --no-extra-configs --no-scene --no-comm --windows 15 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms_avg'])")
    echo "round $round LEAN $cfg: $r" | tee -a $out
  done
  r=$(HNB_CTX_OPTIONS=age_cohort=1 HNB_LIB=$GRAFT_REPO_ROOT/tools/variants/libhanabi_age0.so timeout 600 python bench.py --config c4 --no-parity --pmc off --no-cpu-baseline --no-extra-configs --no-scene --no-comm --windows 15 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms_avg'])")
  echo "round $round LEAN c4: $r" | tee -a $out
done
