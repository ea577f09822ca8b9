#!/bin/bash
# Round 5, second GPU call: why the real-RCCL paths failed (logs kept), the new tests on the slot-indexed nursery, and a same-box A/B of three
# builds of the spawn path on c2_mixed / c2_events: V2 (records at nursery[slot], this tree), V0 (option off: plane-granular stores), V1 (buckets + atomics, the first attempt).
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
TAG=${1:-r05b}
NCCL_DEBUG=INFO NCCL_SOCKET_IFNAME=lo timeout 300 python tests/real_rccl/run_real_comm.py > gpurun_out/${TAG}_rccl.log 2>&1; echo "real rccl rc=$?" ; tail -5 gpurun_out/${TAG}_rccl.log
timeout 300 python bench.py --config c2_mixed --steps 10 --windows 3 --no-parity --pmc off --no-cpu-baseline --no-scene > gpurun_out/${TAG}_bench_comm.out 2> gpurun_out/${TAG}_bench_comm.err; echo "bench with comm rc=$?"; tail -c 600 gpurun_out/${TAG}_bench_comm.err | tail -8; tail -c 300 gpurun_out/${TAG}_bench_comm.out
timeout 900 python -m pytest tests/test_spawn_nursery.py tests/test_verification.py tests/test_device_view.py tests/test_set_module.py tests/test_gpu_parity.py tests/test_gpu_scale.py -m gpu -q --timeout 600 -rf -p no:cacheprovider 2>&1 | tail -40 > gpurun_out/${TAG}_pytest.log; tail -12 gpurun_out/${TAG}_pytest.log
one() { # name, env...
  name=$1; shift
  env "$@" timeout 300 python bench.py --config $CFG --steps 20 --no-parity --pmc off --no-cpu-baseline --no-scene --no-comm 2>gpurun_out/${TAG}_ab_err.log | tail -1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); f = json.load(open('gpurun_out/bench_full.json')); st = f['stages']
print('$CFG %-4s ms_per_step %.4f  min/med/max %s  init %.4f update %.4f lists %.4f' % ('$name', d['ms_per_step'], d['windows']['ms_per_step_min_median_max'], st['init_ms_avg'], st['update_ms_avg'], st['lists_ms_avg']))" >> gpurun_out/${TAG}_ab.log 2>&1
}
for CFG in c2_mixed c2_events; do
  for rep in 1 2; do
    one V2 X=1
    one V0 HNB_CTX_OPTIONS=spawn_nursery=0
    one V1 HNB_LIB=$R/bevy_hanabi_amd/libhanabi_amd_v1.so
  done
done
CFG=c2_interop; one V2 X=1; one V0 HNB_CTX_OPTIONS=spawn_nursery=0
cat gpurun_out/${TAG}_ab.log
HNB_CTX_OPTIONS= timeout 300 python tools/reburst_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${TAG}_reburst.log
