#!/bin/bash
# same-box A/B of compile-time variants:  tools/ab_variants.sh <out dir under gpurun_out> "<bench args>" base <variant> <variant> ...   (two alternations)
out=gpurun_out/$1; args=$2; shift 2
mkdir -p $out
for rep in 1 2; do
for v in "$@"; do
  if [ $v = base ]; then lib=""; else lib=$(pwd)/torch-ngp_amd/variants/$v/libngp_hip.so; fi
  NGP_HIP_LIBRARY=$lib timeout 600 python bench.py $args --no-render --no-dropin --no-cpu-baseline > $out/${v}_$rep.json 2> $out/${v}_$rep.err || echo "$v failed"
  python - <<PY
import json
try:
    l=json.loads(open('$out/${v}_$rep.json').read().strip().splitlines()[-1])
    print('$v', $rep, l['ms_per_step'], ' '.join(f"{r['kernel']}={r['avg_kernel_ms']*1e3:.1f}" for r in l['rooflines']), flush=True)
except Exception as e:
    print('$v', 'no line', e, open('$out/${v}_$rep.err').read()[-600:])
PY
done
done
