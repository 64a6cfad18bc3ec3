#!/bin/bash
# one gpurun call that refreshes the round's evidence on ONE box: gpurun_out/<tag>/...  (copied into profiles/ afterwards)
tag=${1:-r06_final}
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
rocm-smi --showclocks --showpower --showuse > $out/box_state_before.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -q > $out/pytest_gpu.txt 2>&1; echo "pytest rc $?" >> $out/pytest_gpu.txt; tail -3 $out/pytest_gpu.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_n1_driver_args.json 2> $out/bench_n1_driver_args.err; echo "driver-args bench rc $?"
timeout 900 python bench.py --pmc --no-cpu-baseline --no-extra --no-dropin > $out/bench_n1_default.json 2> $out/bench_n1_default.err; echo "default bench rc $?"
timeout 600 python bench.py --steps 256 --warmup 16 --no-lookahead --no-cpu-baseline --no-extra --no-dropin --no-render > $out/bench_n1_no_lookahead.json 2> $out/bench_n1_no_lookahead.err; echo "no-lookahead bench rc $?"
bash tools/gpu_profile.sh $tag/prof_la --steps 64 --warmup 16 --no-render --no-dropin --no-cpu-baseline --no-extra --no-ddp-probe > /dev/null 2>&1
bash tools/gpu_profile.sh $tag/prof_nola --steps 64 --warmup 16 --no-render --no-dropin --no-cpu-baseline --no-extra --no-lookahead --no-ddp-probe > /dev/null 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 300 rocprofv3 --pmc $c --output-format csv -d $root/$out/pmc_$c -o run -- python $root/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-render --no-dropin --no-extra --no-ddp-probe > $root/$out/pmc_$c.log 2>&1) || echo "pmc $c failed"
done
python tools/pmc_traffic.py $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE > $out/pmc_traffic.json 2> $out/pmc_traffic.err || echo "pmc_traffic failed: $(tail -2 $out/pmc_traffic.err)"
rm -rf $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE
timeout 300 python tools/time_update.py > $out/occupancy_refresh.txt 2>&1
timeout 300 python tools/bench_render.py > $out/render_800x800.txt 2>&1
timeout 120 python tools/grid_bwd_probe.py > $out/grid_backward_probe.txt 2>&1
# the inference frame, per kernel: rocprofv3 kernel trace of 4 frames per bracket (network trained 160 steps, as in bench.py's frame)
timeout 300 python tools/render_frames.py --train-steps 160 --save /tmp/ngp_render_model.pt > $out/render_train.txt 2>&1
for sc_ in 300 1; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/prof_render_$sc_ -o p -- python $root/tools/render_frames.py --scale $sc_ --frames 4 --load /tmp/ngp_render_model.pt > $root/$out/render_frames_$sc_.txt 2>&1) || echo "render profile $sc_ failed"
  python tools/profile_summary.py $out/prof_render_$sc_ "python tools/render_frames.py --scale $sc_ --frames 4 --load <network trained 160 steps> (the trace holds the 4 frames only: divide totals by 4)" > $out/render_summary_$sc_.md 2>/dev/null
  rm -rf $out/prof_render_$sc_
done
timeout 120 python tools/table_adam_probe.py > $out/table_adam_probe.txt 2>&1
timeout 300 python tools/graph_lifetime_probe.py > $out/graph_lifetime_probe.txt 2>&1
timeout 300 python tools/unroll_probe.py > $out/unroll_probe.txt 2>&1
rocm-smi --showclocks --showpower > $out/box_state_after.txt 2>&1
python - <<PY
import json
for f in ('bench_n1_driver_args','bench_n1_default','bench_n1_no_lookahead'):
    try:
        l=json.loads(open('$out/'+f+'.json').read().strip().splitlines()[-1])
        print(f, l['value'], l['ms_per_step'], l.get('render_800x800_ms'), [(r['kernel'][:24], r['avg_kernel_ms'], r['frac']) for r in l['rooflines']][:4])
    except Exception as e: print(f, 'failed', e)
PY
head -20 $out/prof_nola/summary.md; cat $out/pmc_traffic.json | head -12; tail -4 $out/render_800x800.txt
