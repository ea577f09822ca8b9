#!/bin/bash
# round 6: the whole GPU suite, then the default bench with the counter CSVs kept, then a two-rank dry run on one GPU
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06b
( time timeout 2400 python -m pytest tests -m gpu -x -q ) > gpurun_out/r06b/pytest.log 2>&1
tail -4 gpurun_out/r06b/pytest.log
( time timeout 1500 python bench.py --keep-pmc gpurun_out/r06b/pmc ) > gpurun_out/r06b/bench.out 2> gpurun_out/r06b/bench.err
tail -1 gpurun_out/r06b/bench.out | head -c 4500; echo
cp profiles/bench_full.json gpurun_out/r06b/bench_full.json 2>/dev/null
( timeout 900 python bench.py --gpus 2 --backend gloo --force-device 0 --comm-lib tests/fake_rccl/libfake_rccl.so --windows 5 --no-parity ) > gpurun_out/r06b/n2_dryrun.out 2> gpurun_out/r06b/n2_dryrun.err
tail -1 gpurun_out/r06b/n2_dryrun.out | head -c 3000; echo
tail -5 gpurun_out/r06b/n2_dryrun.err
