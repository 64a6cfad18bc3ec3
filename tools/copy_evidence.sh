#!/bin/bash
# copy the files of one tools/final_evidence.sh run (gpurun_out/<tag>/) into profiles/ under the round's prefix:  tools/copy_evidence.sh r04_final r04
tag=$1; pre=$2; src=gpurun_out/$tag
cp $src/bench_n1_driver_args.json profiles/${pre}_bench_n1_driver_args.json
cp $src/bench_n1_default.json profiles/${pre}_bench_n1_default.json
cp $src/bench_n1_no_lookahead.json profiles/${pre}_bench_n1_no_lookahead.json
cp $src/prof_la/bench.json profiles/${pre}_bench_n1_profiled.json
cp $src/prof_la/kernel_stats.csv profiles/${pre}_bench_n1_kernel_stats.csv
cp $src/prof_la/summary.md profiles/${pre}_bench_n1_summary.md
cp $src/prof_nola/bench.json profiles/${pre}_bench_n1_profiled_no_lookahead.json
cp $src/prof_nola/kernel_stats.csv profiles/${pre}_bench_n1_kernel_stats_no_lookahead.csv
cp $src/prof_nola/summary.md profiles/${pre}_bench_n1_summary_no_lookahead.md
cp $src/pmc_traffic.json profiles/${pre}_pmc_traffic.json
python3 tools/pmc_traffic.py --annotate profiles/${pre}_pmc_traffic.json
cp $src/pytest_gpu.txt profiles/${pre}_pytest_gpu.txt
cp $src/render_800x800.txt profiles/${pre}_render_800x800.txt
cp $src/occupancy_refresh.txt profiles/${pre}_occupancy_refresh.txt
cp $src/grid_backward_probe.txt profiles/${pre}_grid_backward_probe.txt 2>/dev/null
cat $src/box_state_before.txt $src/box_state_after.txt > profiles/${pre}_box_state.txt
# round 5 additions (absent in older runs: ignored)
for f in render_summary_300.md render_summary_1.md render_frames_300.txt render_frames_1.txt grid_forward_levels.txt graph_lifetime_probe.txt unroll_probe.txt; do
  [ -f $src/$f ] && cp $src/$f profiles/${pre}_$f
done
true
