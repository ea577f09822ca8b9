#!/bin/bash
# copy what tools/gpu_round_check.sh <tag> left in gpurun_out/ into profiles/ under the round's names.  Usage: tools/collect_round.sh <tag> <round, e.g. r06>
T=$1; R=$2; G=gpurun_out; P=profiles
tail -1 $G/${T}_bench.json > $P/${R}_bench.json
cp $G/${T}_bench_full.json $P/${R}_bench_full.json; cp $G/${T}_bench_full.json $P/bench_full.json
cp $G/${T}_traffic.json $P/traffic.json
cp $G/${T}_pmc/pmc_FETCH_SIZE.csv $P/${R}_pmc_FETCH_SIZE.csv; cp $G/${T}_pmc/pmc_WRITE_SIZE.csv $P/${R}_pmc_WRITE_SIZE.csv; cp $G/${T}_pmc/kernel_durations.json $P/${R}_kernel_durations.json
cp $G/${T}_kernel_stats.csv $P/${R}_kernel_stats.csv; cp $G/${T}_kernel_stats.txt $P/${R}_kernel_stats.txt
cp $G/${T}_pytest.log $P/${R}_pytest.log
tail -1 $G/${T}_bench_n2_dryrun.json > $P/${R}_bench_n2_dryrun.json
for f in fuzz.log valu_counters.txt icache.txt; do [ -f $G/${T}_$f ] && cp $G/${T}_$f $P/${R}_$f; done
ls $P | grep -c ${R}_
