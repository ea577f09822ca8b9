# prototype A/B on one box: k_init requests the NEXT round's dead-list slot before it computes the current round (option 17 of the experiment build)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; L=gpurun_out/r05v_ab_init_prefetch.log; : > $L
echo "=== parity tests with the prefetch on" >> $L
HNB_CTX_OPTIONS=init_prefetch=1 timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 300 -p no:cacheprovider 2>&1 | tail -1 >> $L
for CFG in c2 c3 c4; do
  for rep in 1 2; do
    for O in "" "init_prefetch=1"; do
      echo "=== $CFG ${O:-default} (rep $rep)" >> $L
      HNB_CTX_OPTIONS=$O timeout 120 python bench.py --config $CFG --steps 5 --windows 3 --no-parity --pmc off --no-cpu-baseline --no-scene --no-comm --no-extra-configs --full-json /tmp/z.json 2>/dev/null | grep '^{"metric"' | tail -1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print('  burst_init', d.get('burst_init'), ' ms_per_step %.4f' % d['ms_per_step'])" >> $L
    done
  done
done
cat $L
