#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
for i in 1 2; do HNB_DEBUG_ALLOC=1 timeout 300 python bench.py --no-cpu-baseline --no-extra-configs 2>&1 | grep -E "candidate|metric" | cut -c1-400; done > gpurun_out/r02c_alloc.log 2>&1
HNB_SLAB_CANDIDATES=1 timeout 300 python tools/bimodal_probe.py multi > gpurun_out/r02c_bimodal.log 2>&1
HNB_SLAB_CANDIDATES=1 timeout 300 python tools/bimodal_probe.py ctx >> gpurun_out/r02c_bimodal.log 2>&1
for env in "X=1" "OMP_PROC_BIND=spread OMP_PLACES=cores" "OMP_PROC_BIND=close OMP_PLACES=cores" "OMP_PROC_BIND=spread OMP_PLACES=threads OMP_NUM_THREADS=256" "OMP_NUM_THREADS=64 OMP_PROC_BIND=spread OMP_PLACES=cores"; do
  echo "== $env" >> gpurun_out/r02c_cpu.log
  env $env timeout 300 python -c "
import bench, json
r = bench.cpu_baseline(1<<24, frames=100)
print(json.dumps({k: r[k] for k in ('value','cores','hbm_equiv_gbs','sample')}))" >> gpurun_out/r02c_cpu.log 2>&1
done
nproc >> gpurun_out/r02c_cpu.log; lscpu | grep -E "Model name|Socket|NUMA|Thread|Core" >> gpurun_out/r02c_cpu.log
cat gpurun_out/r02c_alloc.log gpurun_out/r02c_bimodal.log gpurun_out/r02c_cpu.log
