#!/bin/bash
# round 6: the lean age-only update path: the ribbon / C5 / scene tests, then C5 (with the parity gate: its plain replay runs the general path) and the scene
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1500 python -m pytest tests -m gpu -x -q -k "ribbon or c5 or ring or scene or reference_examples or skip or horizon or verification or bench_gate or timed" 2>&1 | tail -3
for round in 1 2; do
  r=$(timeout 600 python bench.py --config c5 --pmc off --no-cpu-baseline --no-extra-configs --no-scene --no-comm --windows 15 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms_avg'], d['windows']['ms_per_step_min_median_max'], d['parity']['ok'], d['configs'] if 'configs' in d else '')")
  echo "round $round c5: $r" | tee -a gpurun_out/r06p_lean_age.log
  r=$(timeout 600 python tools/scene_bench.py 1 600 2>/dev/null | grep "frames:" | tail -1)
  echo "round $round scene: $r" | tee -a gpurun_out/r06p_lean_age.log
done
