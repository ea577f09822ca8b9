#!/bin/bash
# Development tool (GPU box): counters of fast and slow slab placements. One rocprofv3 run per counter set.
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; cd /tmp; O=$R/gpurun_out; N=${1:-8}
i=0
for SET in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_DRAM_sum" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" \
           "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum TCC_EA0_RDREQ_LEVEL_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_LEVEL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" \
           "GRBM_GUI_ACTIVE GRBM_UTCL2_BUSY" "GRBM_EA_BUSY GRBM_TC_BUSY"; do
  i=$((i+1)); rm -rf $O/pmc_place$i
  timeout 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $O/pmc_place$i -- python $R/tools/placement_pmc.py $N < /dev/null > $O/pmc_place$i.log 2>&1
  f=$(find $O/pmc_place$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp "$f" $O/r02i_pmc_set$i.csv
  rm -rf $O/pmc_place$i
done
python3 $R/tools/placement_pmc_report.py $N $O/r02i_pmc_set*.csv | tee $O/r02i_placement_counters.txt
# keep the merged-back files small: only the update kernel's rows
for f in $O/r02i_pmc_set*.csv; do (head -1 $f; grep k_update_slots_stream $f) | cut -c1-600 > $f.tmp && mv $f.tmp $f; done
