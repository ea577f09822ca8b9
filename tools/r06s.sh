#!/bin/bash
# round 6: the lean age-only update (nontemporal stores adopted): the whole GPU suite, then c5 and the scene
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -rf -p no:cacheprovider 2>&1 | tail -8 | tee gpurun_out/r06s_pytest.log
for round in 1 2; do
  r=$(timeout 600 python bench.py --config c5 --pmc off --no-cpu-baseline --no-extra-configs --no-scene --no-comm --windows 15 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms_avg'], d['windows']['ms_per_step_min_median_max'], d['parity']['ok'])")
  echo "round $round c5: $r" | tee -a gpurun_out/r06s_lean_age.log
done
