"""GPU: the N > 1 branch of bench.py (one process per rank, HIP graphs around the collectives) exercised on a ONE-GPU box: two ranks share
cuda:0 and exchange through gloo (RCCL refuses two ranks on one device).  Both update modes: sharded (reduce-scatter -> Adam on 1/N ->
all-gather of the fp16 shadows, three graphs) and replicated (averaged all-reduce between two graphs).  Functional check -- the driver
measures scaling on a real 8-GPU node."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('extra', [[], ['--replicated-optim']])
def test_bench_two_ranks_share_one_gpu(extra):
    env = dict(os.environ, NGP_BENCH_SHARE_GPU='1', NGP_BENCH_BACKEND='gloo', MASTER_ADDR='127.0.0.1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1', '--master-port', '29517',
           os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '4', '--warmup', '1', '--no-render', '--no-cpu-baseline', '--no-dropin'] + extra
    res = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    line = json.loads([l for l in res.stdout.splitlines() if l.startswith('{')][-1])
    assert line['n_gpus'] == 2 and line['config']['captures_in_timed_region'] == 0
    assert line['config']['autograd_free_iteration'] and 'hip-graph replay' in line['config']['execution']
    assert ('sharded' in line['config']['parallelism']) == (not extra)
    # gloo moves the 24.5 MB exchange through the host (seconds per step): a functional check, not a speed
    assert line['value'] > 1e4 and line['config']['final_loss'] == line['config']['final_loss']  # finite
    # 2 ranks x ~262 k samples per step
    assert 4e5 < line['config']['samples_per_step_per_gpu'] * 2 < 7e5
