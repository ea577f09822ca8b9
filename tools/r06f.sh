#!/bin/bash
# round 6: what the HIP-event instrumentation of every n-th timed frame costs the timed step (c5, c2, c2_mixed)
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r06f_timing_period.log
: > $out
for round in 1 2; do
  for tp in 5 15 1000000; do
    for cfg in c5 c2 c2_mixed; do
      r=$(timeout 600 python bench.py --config $cfg --timing-period $tp --no-parity --pmc off --no-cpu-baseline --no-extra-configs --no-scene --no-comm --windows 15 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms_avg'], d['windows']['ms_per_step_min_median_max'])")
      echo "round $round timing-period=$tp $cfg: $r" | tee -a $out
    done
  done
done
