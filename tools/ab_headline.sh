#!/bin/bash
# Development tool (GPU box): alternate the headline bench between the in-tree library and tools/ab/libhanabi_prev.so.
R=$GRAFT_REPO_ROOT; cd $R
for i in 1 2 3 4; do
  for v in NEW PREV; do
    if [ $v = PREV ]; then export HNB_LIB=$R/tools/ab/libhanabi_prev.so; else unset HNB_LIB; fi
    python bench.py --no-cpu-baseline --steps 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v kernel %.4f ms  step %.4f ms  compact %.4f' % (d['roofline']['kernel_ms_avg'], d['ms_per_step'], d['roofline']['compact_ms_avg']))"
  done
done
unset HNB_LIB
python tools/bench_configs.py c2die 2>&1 | sed -n 2,6p
python tools/bench_configs.py c4 2>&1 | grep "C4 inst"
