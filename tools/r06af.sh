#!/bin/bash
# round 6, final tree: long differential soaks (random worlds in both list orders, random parent / child systems, random scenes on the merged launches) against the
# oracle under the default options, with every init pass slot-major, and with the frame parameters copied instead of host-written
cd "$GRAFT_REPO_ROOT" || exit 1
export HNB_JIT_CACHE=$GRAFT_REPO_ROOT/bevy_hanabi_amd/jit_cache
L=gpurun_out/r06af_soak.log; : > $L
for opt in "" slot_init=2 direct_upload=0; do
  echo "HNB_CTX_OPTIONS=${opt:-(defaults)}" >> $L
  HNB_CTX_OPTIONS=$opt SOAK_LO=300 SOAK_HI=380 timeout 900 python tools/soak_fuzz.py 2>&1 | tail -1 >> $L
  HNB_CTX_OPTIONS=$opt timeout 600 python tests/fuzz_sweep.py --backend gpu --jit 0 --scene 8 --seeds 20000:20400 2>&1 | tail -1 >> $L
  HNB_CTX_OPTIONS=$opt timeout 600 python tests/fuzz_sweep.py --backend gpu --jit 1 --capacity 9000 --frames 60 --seeds 21000:21060 2>&1 | tail -1 >> $L
done
cat $L
