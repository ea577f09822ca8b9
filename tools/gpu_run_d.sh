#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 300 ./tools/layout_probe 16777216 2 20 2>&1 | tail -8 > gpurun_out/r02d_layout_probe.log; cat gpurun_out/r02d_layout_probe.log
for env in "X=1" "OMP_PROC_BIND=spread OMP_PLACES=cores" "OMP_PROC_BIND=close OMP_PLACES=cores" "OMP_PROC_BIND=spread OMP_PLACES=threads OMP_NUM_THREADS=256" "OMP_NUM_THREADS=64 OMP_PROC_BIND=spread OMP_PLACES=cores"; do
  echo "== $env" >> gpurun_out/r02d_cpu.log
  env $env timeout 300 python -c "
import bench, json
r = bench.cpu_baseline(1<<24, frames=40)
print(json.dumps({k: r[k] for k in ('value','cores','hbm_equiv_gbs','sample')}))" >> gpurun_out/r02d_cpu.log 2>&1
done
cat gpurun_out/r02d_cpu.log
