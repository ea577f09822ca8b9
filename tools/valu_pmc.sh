#!/bin/bash
# Development tool (GPU box): SQ counters of the init and update kernels of c2 / c3 / c4 (is a kernel bound by VALU issue?).
# One pass: 8 SQ counters + GRBM_GUI_ACTIVE. Usage: tools/valu_pmc.sh
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; cd /tmp; O=$R/gpurun_out
for CFG in c2 c2_mixed c2_dieoff c2_reburst c3 c4; do
  rm -rf $O/valu_$CFG
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $O/valu_$CFG -- python $R/bench.py --config $CFG --no-cpu-baseline --no-extra-configs --no-parity --pmc off --no-scene --no-comm --steps 10 > $O/valu_$CFG.log 2>&1
done
python3 - <<'PY'
import csv, glob, os, collections
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out"
for cfg in ("c2", "c2_mixed", "c2_dieoff", "c2_reburst", "c3", "c4"):
    fs = glob.glob(f"{O}/valu_{cfg}/**/*counter_collection.csv", recursive=True)
    if not fs:
        print(cfg, "no counters:", open(f"{O}/valu_{cfg}.log").read()[-400:]); continue
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"]
        if "k_init" in k or "k_update_slots_stream" in k:
            per[("init" if "k_init" in k else "update")][r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    for kind, ctrs in per.items():
        # the largest launch of the kind (the burst init / any update)
        pick = {}
        for name, vals in ctrs.items():
            vals.sort()
            pick[name] = max(v for _, v in vals) if kind == "init" else sorted(v for _, v in vals)[len(vals) // 2]
        g = pick.get("GRBM_GUI_ACTIVE", 0.0)
        print(f"{cfg} {kind}: " + ", ".join(f"{n}={v:.4g}" for n, v in sorted(pick.items())))
        if g:
            simds = 1024.0
            print(f"   VALU issue cycles per SIMD / kernel cycles (gfx94x VALUBusy formula: ACTIVE_INST_VALU*4/SIMDs/(GUI_ACTIVE/8 XCDs)) = {pick.get('SQ_ACTIVE_INST_VALU', 0) * 4 / simds / (g / 8.0):.3f};"
                  f"  VALU instructions per wave = {pick.get('SQ_INSTS_VALU', 0) / max(pick.get('SQ_WAVES', 1), 1):.1f};"
                  f"  wave cycles: active {pick.get('SQ_ACTIVE_INST_ANY', 0) / max(pick.get('SQ_WAVE_CYCLES', 1), 1):.3f}, waiting on memory/barrier {pick.get('SQ_WAIT_ANY', 0) / max(pick.get('SQ_WAVE_CYCLES', 1), 1):.3f}, issue-stalled {pick.get('SQ_WAIT_INST_ANY', 0) / max(pick.get('SQ_WAVE_CYCLES', 1), 1):.3f}")
PY
for CFG in c2 c2_mixed c2_dieoff c2_reburst c3 c4; do rm -rf $O/valu_$CFG; done
