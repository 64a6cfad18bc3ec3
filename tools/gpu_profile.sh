#!/bin/bash
# one rocprofv3 kernel-trace run of bench.py on the GPU box; usage: tools/gpu_profile.sh <out name under gpurun_out/> <bench args...>
# writes gpurun_out/<name>/{bench.json, kernel_stats.csv, summary.md}
name=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$name
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o p -- python $root/bench.py "$@" > $out/bench.json 2> $out/bench.err
echo "rocprofv3 rc $?"
cd $root
f=$(find $out/prof -name "*kernel_stats.csv" | head -1)
cp "$f" $out/kernel_stats.csv
python tools/profile_summary.py $out/prof "python bench.py $*" > $out/summary.md
rm -rf $out/prof
head -24 $out/summary.md
