#!/bin/bash
# usage (in the repository, after `gpurun -- tools/measure_round.sh <tag>` has merged gpurun_out/ back): tools/collect_profiles.sh <tag>
# copies what is judged from the scratch directory into profiles/ under the names profiles/README.md lists
t=$1; o=gpurun_out; p=profiles
last() { ls -t $1 2>/dev/null | head -1; }   # the NEWEST match (a tag measured twice leaves both runs in the scratch directory)
cp $o/${t}_bench.json $p/${t}_bench.json
cp $o/${t}_bench_prof.json $p/${t}_bench_under_rocprof.json
cp "$(last "$o/${t}_stats/*/*kernel_stats.csv")" $p/${t}_kernel_stats.csv
for w in ba_100x10k curvefit_10k ba_so3_500x50k; do cp $o/${t}_bench_$w.json $p/${t}_bench_$w.json; cp "$(last "$o/${t}_stats_$w/*/*kernel_stats.csv")" $p/${t}_kernel_stats_$w.csv; done
cp $o/${t}_bench_ba_10kx1M.json $p/${t}_bench_ba_10kx1M.json
cp $o/${t}_bench_shuffled.json $p/${t}_bench_shuffled.json
for x in nofloor forcedist; do [ -f $o/${t}_bench_$x.json ] && cp $o/${t}_bench_$x.json $p/${t}_bench_$x.json; done
cp $o/${t}_bench_dense.json $p/${t}_bench_dense.json; cp "$(last "$o/${t}_stats_dense/*/*kernel_stats.csv")" $p/${t}_kernel_stats_dense.csv
cp "$(last "$o/${t}_fetch/*/*counter_collection.csv")" $p/${t}_pmc_fetch_sweep.csv; cp "$(last "$o/${t}_write/*/*counter_collection.csv")" $p/${t}_pmc_write_sweep.csv
for w in ba_100x10k ba_so3_500x50k; do
  cp "$(last "$o/${t}_fetch_$w/*/*counter_collection.csv")" $p/${t}_pmc_fetch_sweep_$w.csv; cp "$(last "$o/${t}_write_$w/*/*counter_collection.csv")" $p/${t}_pmc_write_sweep_$w.csv
done
for w in ba_100x10k ba_so3_500x50k ba_1kx100k; do cp "$(last "$o/${t}_valu_$w/*/*counter_collection.csv")" $p/${t}_pmc_issue_sweep_$w.csv; done
for k in band dense; do f="$(last "$o/${t}_mfma_$k/*/*counter_collection.csv")"; [ -n "$f" ] && cp "$f" $p/${t}_pmc_mfma_$k.csv; done
cp $o/${t}_pmc_traffic.json $p/pmc_traffic.json; cp $o/${t}_pmc_mfma.json $p/pmc_mfma.json
[ -f $o/${t}_pmc_iter.json ] && cp $o/${t}_pmc_iter.json $p/pmc_iter.json
for w in ba_1kx100k ba_100x10k; do for x in fetch_mf fetch_mat write_mf write_mat; do f="$(last "$o/${t}_iter_${x}_$w/*/*counter_collection.csv")"; [ -n "$f" ] && cp "$f" $p/${t}_pmc_iter_${x}_$w.csv; done; done
[ -f $o/${t}_bench_materialised.json ] && cp $o/${t}_bench_materialised.json $p/${t}_bench_materialised.json
f="$(last "$o/${t}_stats_materialised/*/*kernel_stats.csv")"; [ -n "$f" ] && cp "$f" $p/${t}_kernel_stats_materialised.csv
for w in ba_grid_40x40 ba_grid_100x100 ba_grid_40x40_windowed; do [ -f $o/${t}_bench_$w.json ] && cp $o/${t}_bench_$w.json $p/${t}_bench_$w.json; done
f="$(last "$o/${t}_stats_ba_grid_40x40/*/*kernel_stats.csv")"; [ -n "$f" ] && cp "$f" $p/${t}_kernel_stats_ba_grid_40x40.csv
[ -f $o/${t}_tsp_try.json ] && cp $o/${t}_tsp_try.json $p/${t}_tsp_try.json
ls -la $p | grep "${t}_" | wc -l
