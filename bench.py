#!/usr/bin/env python3
"""Headline benchmark of the instant-ngp hot path on MI355X (contract: see the task statement).

Workload (BASELINE.json configs[1], lego-shaped, synthetic): one STEP is one full training iteration of the
`--fp16 --cuda_ray --ff` path on 4096 rays per GPU --
    near/far -> march_rays_train -> hash-grid encode -> sigma FFMLP -> trunc_exp -> SH encode -> colour FFMLP -> sigmoid
    -> composite_rays_train -> MSE loss -> backward through all of it -> (N>1: RCCL all-reduce of the gradients)
    -> GradScaler + Adam over the 12.26 M parameters,
plus, every 16 steps, the occupancy-grid refresh (`update_extra_state`: 2 x 128^3/4 density queries + EMA + packbits),
exactly the cadence of the reference's Trainer (nerf/utils.py:851-873).  The synthetic scene keeps its analytic
occupancy (the refreshed grid is computed and discarded) so that the number of samples per ray stays at the lego-like
~65.  Inputs (rays, ground-truth colours, occupancy) are resident in HBM before the timed region starts.

metric  : training samples/s = sum over steps and ranks of the samples actually marched (counter[0]) / wall time
execution: the iteration is captured once into a HIP graph (torch-ngp_amd/graph.py) and replayed per step; --no-graph issues
          every launch eagerly, --no-fused uses the reference-style module-by-module network path.
lookahead (N = 1, default; --no-lookahead switches it off): the batch pool plays the data loader, which knows its next batch one step early;
          near/far + march_rays_train of batch k+1 (they need no weights) run on a side stream while the iteration of batch k runs.  Work per
          timed step is unchanged: K timed steps contain K marches (the first step's march ran just before t0, the last step marches the
          batch after the region), every sample that is counted was marched, encoded, evaluated, composited and trained on.
roofline: `roofline` = the dominant kernel (grid_encode_backward, 1100 B per point), `rooflines` = all timed kernels
          (grid fwd/bwd, ffmlp fwd/bwd with their MFMA fraction).  HIP graphs cannot carry timing events, so after the
          timed region the same iteration is run eagerly for a few steps with HIP-event pairs around the named kernels on
          the launch stream; algorithmic bytes per SURVEY.md 8(d); `traffic` from the committed rocprofv3 --pmc passes.
          `traffic` is NOT measured in this run (PMC needs rocprofv3 around the process): it is the per-launch HBM byte count of the
          committed counter pass named in `traffic_source`.
optimizer (N = 1, default; --no-fused-adam switches it off): the hash table's Adam sweep rides in the grid backward's slice accumulate
          (speculative double buffer, device-side parity word: GradScaler's skip rule exactly, torch-ngp_amd/optim.py enable_table_fusion) and the
          step closes with one small launch (MLP weights, dense table levels, loss-scale commit, parity flip); `config.table_adam_in_grid_backward`
          says which form ran.  The `roofline` row of that call counts the Adam bytes it carries (`carries_table_adam` gives both definitions).
          N > 1 exchanges the gradient first and keeps the separate sweep (on 1 / N of the table per rank).
steady state: every HIP graph the timed region replays (the training iteration and the occupancy refresh) is captured AND replayed
          at least once during the untimed setup (33 iterations: the reference's 16 worst-case-sized steps, the first estimate, one
          full refresh period from graphs); `captures_in_timed_region` reports graph captures that happened between t0 and t1 (0).
dropin_path: the same workload through the pure drop-in surface only -- module-by-module network (`model.fused = False`), eager
          launches, torch.optim.Adam + GradScaler (what the reference's unchanged network_ff.py / renderer.py / Trainer execute) --
          timed the same way on a shorter run (N = 1 only).
cpu_baseline: the reference's pure-PyTorch path (NeRFRenderer.run, nn.Linear MLPs, fp32, device='cpu', restated in
          oracle/torch_cpu.py because the reference's own encoders are CUDA-only) on all host cores; `scalar_port` = the scalar-C
          oracle's cuda_ray-shaped step on one thread.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (os.path.join(ROOT, 'torch-ngp_amd'), ROOT):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
MFMA_F16_PEAK_TFLOPS = 2500.0  # dense fp16 MFMA peak, same guide

# Kernels timed for the roofline section: C-ABI symbol -> (label, index of the batch-size argument, algorithmic bytes per
# unit, algorithmic flops per unit, unit).  Byte/flop counts are SURVEY.md 8(d); the ffmlp rows depend on the layer count
# and are resolved at call time (in 64 B + out 32 B + 128 B per stored hidden layer; backward: grad 32 + saved activations
# 128/layer + inputs 64 + dL/dx 64).
def _ff_fwd_bytes(nl): return 64.0 + 32.0 + 128.0 * nl
def _ff_bwd_bytes(nl, stored=True): return 32.0 + (128.0 * nl if stored else 0.0) + 64.0 + 64.0   # stored=False: NGP_FF_RECOMPUTE, no forward buffer read
def _ff_flops(nl): return 2.0 * 64 * (32 + 64 * (nl - 1) + 16)

TIMED = {
    'ngp_grid_encode_forward': ('grid_encode_forward', 4, lambda a: 588.0, lambda a: 0.0, 'point'),
    'ngp_grid_encode_backward': ('grid_encode_backward', 5, lambda a: 1100.0, lambda a: 0.0, 'point'),
    'ngp_ffmlp_forward': ('ffmlp_forward', 2, lambda a: _ff_fwd_bytes(a[6]), lambda a: _ff_flops(a[6]), 'sample'),
    'ngp_ffmlp_backward': ('ffmlp_backward', 4, lambda a: _ff_bwd_bytes(a[8]), lambda a: 2.0 * _ff_flops(a[8]), 'sample'),
    'ngp_grid_encode_forward_ex': ('grid_encode_forward', 4, lambda a: 588.0, lambda a: 0.0, 'point'),
    'ngp_grid_encode_forward_sched': ('grid_encode_forward', 4, lambda a: 588.0, lambda a: 0.0, 'point'),
    'ngp_grid_encode_forward_sel': ('grid_encode_forward', 7, lambda a: 588.0, lambda a: 0.0, 'point'),   # (double-buffered table: optim.NGPAdam.enable_table_fusion)
    'ngp_grid_encode_backward_ex': ('grid_encode_backward', 5, lambda a: 1100.0, lambda a: 0.0, 'point'),
    'ngp_grid_encode_backward_ws': ('grid_encode_backward', 5, lambda a: 1100.0, lambda a: 0.0, 'point'),
    'ngp_grid_encode_backward_checked': ('grid_encode_backward', 5, lambda a: 1100.0, lambda a: 0.0, 'point'),
    # (+ the MLPs' slab reduction in its last launch; + 26 B per table parameter when the table's Adam sweep rides in the accumulate's flush:
    # master weight and two moments read and written, fp16 shadow written -- the gradient's store and re-read are gone)
    'ngp_grid_encode_backward_checked_slabs': ('grid_encode_backward', 5, lambda a: 1100.0 + 26.0 * _table_adam_params(a) / max(int(a[5]), 1),
                                               lambda a: 0.0, 'point'),
    # what is left of the optimizer step when the table is updated in the grid backward: Adam on the MLP weights + commit, one workgroup
    'ngp_optim_adam_small_commit': ('k_adam_small_commit (dense table levels + MLP weights + scaler commit + parity flip)', lambda a: _small_params(a), lambda a: 28.0,
                                    lambda a: 0.0, 'parameter'),
    'ngp_ffmlp_forward_ex': ('ffmlp_forward', 2, lambda a: _ff_fwd_bytes(a[6]), lambda a: _ff_flops(a[6]), 'sample'),
    # the whole network behind the encoder in one launch: enc 64 B + dir 12 B in, both forward buffers (when they are stored: not with the
    # recomputing backward) + h16 32 B + colour input 64 B + sigma 4 B + rgb 12 B out per sample; flops of both MLPs
    'ngp_network_forward': ('network_forward', 2,
                            lambda a: (64.0 + 12.0 + (128.0 * (a[6] + a[7]) if a[10] else 0.0) + 32.0 + 64.0 + 4.0 + 12.0) if a[9] else (64.0 + 12.0 + 4.0 + 12.0),
                            lambda a: _ff_flops(a[6]) + _ff_flops(a[7]), 'sample'),
    'ngp_ffmlp_backward_ex': ('ffmlp_backward', 4, lambda a: _ff_bwd_bytes(a[8], bool(a[3])), lambda a: 2.0 * _ff_flops(a[8]), 'sample'),
    # the colour network's backward (it also writes the sigma network's output gradient: + 4 B sigma gradient in, 32 B out per sample)
    'ngp_network_backward_color': ('ffmlp_backward (colour net)', 4, lambda a: _ff_bwd_bytes(a[5], bool(a[3])) + 36.0,
                                   lambda a: 2.0 * _ff_flops(a[5]), 'sample'),
    # compositing + loss + their backward in one launch: sigma 4 + rgb 12 + deltas 8 in, 4 + 32 gradient out per sample (+ rays, target,
    # image, depth, weights: 80 B per ray) -- SURVEY.md 8(d): 64 B / sample + 80 B / ray
    'ngp_composite_train_loss_backward': ('composite + loss + backward', 4, lambda a: 64.0 + 80.0 * a[5] / max(a[4], 1), lambda a: 0.0, 'sample'),
    # near/far + both passes of march_rays_train: 32 B written per sample (xyz, dir, deltas) + 48 B per ray (origins, directions, rays, near/far)
    'ngp_march_rays_train_aabb': ('march_rays_train (+ near/far)', 9, lambda a: 32.0 + 48.0 * a[6] / max(a[9], 1), lambda a: 0.0, 'sample'),
    # Adam + loss-scale logic + fp16 shadows + gradient zeroing over every parameter: fp32 master / two moments read and written (24 B),
    # fp16 gradient read and zeroed (4 B; 2 B when the producer overwrites the buffer and the kernel keeps it), fp16 shadow written (2 B) = 30 (28) B
    # per parameter.  Only the calls that UPDATE are rows.
    'ngp_optim_adam_step_ex': ('k_adam (Adam + scaler + shadows + gradient zeroing)', lambda a: _adam_params(a), lambda a: _adam_bytes(a), lambda a: 0.0, 'parameter'),
}


def _table_adam_params(a):
    """table parameters whose Adam sweep one ngp_grid_encode_backward_checked_slabs call carries (0: none)"""
    import ctypes
    import _ngp_capi as capi
    ss, oh = a[22], a[18]
    if not ss or not oh:
        return 0
    sets = ctypes.cast(ss, ctypes.POINTER(capi.SlabSets)).contents
    if not sets.table_adam:
        return 0
    return int(ctypes.cast(oh, ctypes.POINTER(ctypes.c_int32))[int(a[8])]) * int(a[7])


def _small_params(a):
    import ctypes
    k = int(a[0])
    if k == 0 or not a[1]:
        return int(a[20] or 0)
    n = ctypes.cast(a[1], ctypes.POINTER(ctypes.c_uint64))
    return int(sum(n[i] for i in range(k))) + int(a[20] or 0)   # (+ the dense-level prefix of the double-buffered table)


def _adam_bytes(a):
    """bytes per parameter of one update call: 30, or 28 for a tensor whose gradient buffer is kept (grad_is_half & 2: not zeroed)"""
    import ctypes
    k = int(a[0])
    if k == 0 or not a[1] or not a[7]:
        return 30.0
    n = ctypes.cast(a[1], ctypes.POINTER(ctypes.c_uint64))
    h = ctypes.cast(a[7], ctypes.POINTER(ctypes.c_int))
    tot = sum(n[i] for i in range(k))
    return sum(n[i] * (28.0 if (h[i] & 2) else 30.0) for i in range(k)) / max(tot, 1)


def _fused_switches():
    """the optional launch fusions / modes of fused.py as this run had them (--ab-off and the NGP_FUSED_* environment switches change them)"""
    import fused
    return {k: bool(getattr(fused, k)) for k in sorted(dir(fused)) if k.startswith('USE_')}


def _adam_params(a):
    """parameters streamed by one ngp_optim_adam_step_ex call: the sum of its n[] array when the UPDATE phase is asked for, else 0"""
    import ctypes
    k, phases = int(a[0]), int(a[19])
    if k == 0 or not (phases & 2) or not a[1]:
        return 0
    n = ctypes.cast(a[1], ctypes.POINTER(ctypes.c_uint64))
    return int(sum(n[i] for i in range(k)))


class KernelTimers:
    """HIP-event pairs around selected C-ABI calls, recorded on the stream the kernels are launched on (the current PyTorch
    stream, which is what `_ngp_capi.stream()` hands to the library)."""

    def __init__(self, capi):
        self.capi, self.records, self.enabled, self.orig, self.step, self.suffix = capi, {}, False, {}, 0, ''
        for sym in TIMED:
            self.orig[sym] = getattr(capi.lib, sym)
            setattr(capi.lib, sym, self._wrap(sym))

    def _wrap(self, sym):
        inner = self.orig[sym]
        label, b_idx, fbytes, fflops, unit = TIMED[sym]

        def call(*args):
            if not self.enabled:
                return inner(*args)
            units = int(b_idx(args)) if callable(b_idx) else int(args[b_idx])
            if units <= 0:   # (a call that moves nothing of this row's kind: e.g. the optimizer's CHECK / COMMIT phases)
                return inner(*args)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            rc = inner(*args)
            b.record()
            self.records.setdefault(label + self.suffix, []).append((a, b, units, fbytes(args), fflops(args), unit, self.step))
            return rc
        return call

    def summary(self, traffic=None, marched=None):
        """marched[k]: samples iteration k really marched.  A launch is sized for the (padded) sample buffer; the rows behind the marched
        ones are zero rows that move no algorithmic bytes, so a launch's units are min(rows launched, samples marched)."""
        out = []
        for label, recs in self.records.items():
            ms = np.array([a.elapsed_time(b) for a, b, *_ in recs])
            unit_of = [min(r[2], marched[r[6]]) if marched and r[5] != 'parameter' and r[6] < len(marched) and marched[r[6]] > 0 else r[2] for r in recs]
            units = np.array(unit_of, dtype=np.float64)
            byts = np.array([u * r[3] for u, r in zip(unit_of, recs)])
            flops = np.array([u * r[4] for u, r in zip(unit_of, recs)])
            t = float(ms.mean()) * 1e-3
            gbs = float(byts.mean()) / t / 1e9
            row = {'kernel': label, 'bound': 'hbm', 'achieved': round(gbs, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                   'frac': round(gbs / HBM_PEAK_GBS, 4), 'traffic': None, 'avg_kernel_ms': round(float(ms.mean()), 4),
                   'units_per_launch': round(float(units.mean()), 1), 'bytes_per_unit': round(float(byts.mean() / units.mean()), 1),
                   'unit_name': recs[0][5], 'launches': len(recs), 'total_ms': round(float(ms.sum()), 3),
                   'rows_per_launch': round(float(np.mean([r[2] for r in recs])), 1)}
            if flops.mean() > 0:
                tf = float(flops.mean()) / t / 1e12
                row['mfma'] = {'achieved': round(tf, 2), 'peak': MFMA_F16_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': round(tf / MFMA_F16_PEAK_TFLOPS, 5)}
            if traffic and label in traffic:
                row['traffic'] = traffic[label]
            if label.startswith('grid_encode_backward') and float(byts.mean() / units.mean()) > 1100.5:
                # the launch also carries the table's Adam sweep (26 B per table parameter): both definitions, side by side
                g1 = 1100.0 * float(units.mean()) / t / 1e9
                row['carries_table_adam'] = {'bytes_per_unit_grid_backward': 1100.0, 'bytes_per_unit_table_adam': round(float(byts.mean() / units.mean()) - 1100.0, 1),
                                             'frac_without_adam_bytes': round(g1 / HBM_PEAK_GBS, 4),
                                             'note': 'frac counts the 1100 B / point of SURVEY 8(d) PLUS the 26 B per table parameter of the Adam sweep this launch performs '
                                                     '(master weight and two moments read and written, fp16 shadow written); frac_without_adam_bytes is the old definition on the '
                                                     'new, longer launch; the separate-kernel figures (--no-fused-adam): grid backward 0.26, k_adam 0.70'}
            if label.startswith('grid_encode_forward'):
                # what actually bounds the encoder (profiles/r03_grid_forward_pmc.json: L1 hit 69 %, L2 hit 91 %, 37 MB from HBM): the L1's
                # line rate.  Line requests per point of k_grid_forward_fast on the L16 table: 4 per level (hashed: the x-pair shares an
                # aligned 64-byte block; dense: four 8-byte loads) + ~4 for the positions re-read per level and the output rows = 68;
                # ceiling = one line per clock and CU: 256 CUs x 2.4 GHz.
                lines = 68.0 * float(units.mean())
                floor_ms = lines / (256 * 2.4e9) * 1e3
                row['l1'] = {'bound': 'l1', 'lines_per_point': 68.0, 'peak_lines_per_s': 256 * 2.4e9, 'floor_ms': round(floor_ms, 4),
                             'frac': round(floor_ms / max(float(ms.mean()), 1e-9), 4)}
            out.append(row)
        out.sort(key=lambda r: -r['total_ms'])
        return out


# whole-frame accounting of the 800x800 inference frame (VERDICT r4 "What's missing" 3): algorithmic bytes of the loop's stages, per sample ROW
# that may carry a sample (alive rays x n_step of the iteration) and per alive ray
#   march_rays      writes xyz 12 + dir 12 + deltas 8 = 32 B per row; reads origin 12 + direction 12 + near/far 8 + t 4 + alive 4 = 40 B per ray
#   network         the fused inference network: positions / directions 24 B in, sigma 4 + rgb (fp16x4 rows: 8) out per row -- the encoder's table
#                   gathers (588 B per point) are what the stage really moves: both figures are reported
#   composite_rays  reads sigma 4 + rgb 12 + deltas 8 = 24 B per row; weights_sum / depth / image / t read + written = 48 B + alive 4 per ray
#   compact_rays    alive index read + written = 8 B per ray
FRAME_STAGE_BYTES = {'march_rays': (32.0, 40.0), 'network (encoder + MLPs + glue)': (36.0 + 588.0, 0.0), 'casts (density_scale, fp32 copies)': (4 + 4 + 8 + 12.0, 0.0),
                     'composite_rays': (24.0, 52.0), 'compact_rays': (0.0, 8.0)}


def frame_stages(model, ro, rd, rkw):
    """one more frame with the loop's stage probe on (nerf/renderer.py `_loop_probe`): per stage the summed HIP-event time, launches, the
    rows / rays that carried work, algorithmic GB/s -- and how much of the frame's wall time the stages account for"""
    model._loop_probe = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    try:
        with torch.no_grad(), torch.autocast('cuda', dtype=torch.float16):
            model.render(ro, rd, **rkw)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) * 1e3
        probe = model._loop_probe
    finally:
        model._loop_probe = None
    rows_by_stage = {}
    first, last = None, None
    states = {}
    for name, e0, e1, lanes, rows, n_total, snap in probe:
        if id(snap) not in states:
            alive = int(snap[0].item())
            states[id(snap)] = (alive, max(1, min(8, n_total // max(alive, 1))))
        alive, n_step = states[id(snap)]
        r = rows_by_stage.setdefault(name, {'ms': 0.0, 'launches': 0, 'rows': 0, 'rays': 0, 'rows_launched': 0})
        r['ms'] += e0.elapsed_time(e1)
        r['launches'] += 1
        r['rows'] += min(rows, alive * n_step)
        r['rays'] += min(lanes, alive)
        r['rows_launched'] += rows
        first = e0 if first is None else first
        last = e1
    out = []
    for name, r in rows_by_stage.items():
        per_row, per_ray = FRAME_STAGE_BYTES.get(name, (0.0, 0.0))
        byts = per_row * r['rows'] + per_ray * r['rays']
        gbs = byts / (r['ms'] * 1e-3) / 1e9 if r['ms'] > 0 else 0.0
        out.append({'stage': name, 'ms': round(r['ms'], 4), 'launch_groups': r['launches'], 'sample_rows': r['rows'], 'rows_launched': r['rows_launched'],
                    'alive_rays_summed': r['rays'], 'algorithmic_GBps': round(gbs, 1), 'frac_of_hbm_peak': round(gbs / HBM_PEAK_GBS, 4)})
    trajectory = []
    for name, e0, e1, lanes, rows, n_total, snap in probe:
        if name == 'march_rays':
            alive, n_step = states[id(snap)]
            trajectory.append([alive, n_step, rows])
    staged = sum(r['ms'] for r in out)
    loop_ms = first.elapsed_time(last) if first is not None else 0.0
    return {'frame_wall_ms_with_probe': round(wall, 3), 'loop_first_to_last_event_ms': round(loop_ms, 3), 'stages_sum_ms': round(staged, 3),
            'iterations': len(probe) // max(len(rows_by_stage), 1), 'stages': sorted(out, key=lambda r: -r['ms']),
            'alive_nstep_rows_per_iteration': trajectory,
            'note': 'HIP-event pairs around the stages of every loop iteration (events add ~2 us each: the probed frame is slower than the timed one); '
                    'rows = alive rays x n_step of the iteration (an upper bound of the samples emitted)'}


def measure_pmc_traffic(argv_tail):
    """`--pmc`: HBM bytes per launch measured FOR THIS RUN'S WORKLOAD on this box: two `rocprofv3 --pmc` passes (FETCH_SIZE, WRITE_SIZE --
    they do not fit one pass, and no tracing domain is combined with --pmc, /opt/skills/guides/MI355X_MICROARCH.md) around a short run of
    this script in the same mode, reduced by tools/pmc_traffic.py (the guide's unit and gfx950 read corrections).  -> (per-launch dict,
    source string) or (None, reason).  ~1 minute; needs rocprofv3 on the box."""
    import shutil
    import subprocess
    import tempfile
    if shutil.which('rocprofv3') is None:
        return None, 'rocprofv3 not found on this box'
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import pmc_traffic
    work = tempfile.mkdtemp(prefix='ngp_pmc_', dir='/tmp')
    env = dict(os.environ, TMPDIR='/tmp')
    dirs = {}
    try:
        for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
            d = os.path.join(work, counter)
            cmd = ['rocprofv3', '--pmc', counter, '--output-format', 'csv', '-d', d, '-o', 'run', '--', sys.executable, os.path.abspath(__file__),
                   '--steps', '6', '--warmup', '2', '--no-cpu-baseline', '--no-roofline', '--no-render', '--no-dropin', '--no-extra', '--no-ddp-probe'] + argv_tail
            r = subprocess.run(cmd, cwd='/tmp', env=env, capture_output=True, text=True, timeout=420)
            if r.returncode != 0:
                return None, f'rocprofv3 --pmc {counter} failed (rc {r.returncode}): {r.stderr[-200:]}'
            dirs[counter] = d
        fetch = pmc_traffic.per_kernel(dirs['FETCH_SIZE'], 'FETCH_SIZE')
        write = pmc_traffic.per_kernel(dirs['WRITE_SIZE'], 'WRITE_SIZE')
        per_launch = {}
        for key in set(fetch) | set(write):
            label, factor = pmc_traffic.KERNELS[key]
            per_launch[label] = per_launch.get(label, 0) + round(fetch.get(key, (0.0, 0))[0] * 1024.0 * factor + write.get(key, (0.0, 0))[0] * 1024.0)
        return per_launch, 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of THIS run (bench.py --pmc; read corrections of tools/pmc_traffic.py)'
    except Exception as e:  # noqa: BLE001 -- a measurement aid: the line must not depend on it
        return None, f'pmc passes failed: {e!r}'[:240]
    finally:
        shutil.rmtree(work, ignore_errors=True)


def load_pmc_traffic():
    """(HBM bytes per launch, file name) from the newest committed rocprofv3 --pmc pass (profiles/*_pmc_traffic.json, produced by
    tools/pmc_traffic.py on the same workload); (None, None) when the file is absent.  NOT a measurement of this run."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_pmc_traffic.json')))
    if not files:
        return None, None
    try:
        return json.load(open(files[-1]))['per_launch'], os.path.relpath(files[-1], ROOT)
    except Exception:
        return None, None


def load_march_pmc():
    """SQ_INSTS_VALU of k_march_rays per opaque 800x800 frame from the newest committed counter pass (profiles/*_march_rays_pmc.json)"""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_march_rays_pmc.json')))
    if not files:
        return None
    try:
        d = json.load(open(files[-1]))
        return {'valu_insts_per_frame': float(d['valu_insts_per_frame']), 'source': os.path.relpath(files[-1], ROOT)}
    except Exception:
        return None


RING_BLOCK = 8         # steps between snapshots of the model's 16-slot sample-counter ring (see TrainingRun.train_step)
SETUP_ITERATIONS = 33  # 16 worst-case-sized eager steps + the first estimate-sized one (graph capture) + one refresh period from graphs


class TrainingRun:
    """one configuration of the training workload (model + optimizer + stepper + resident batches) and its timing protocol"""

    def __init__(self, args, dev, world, rank, fused, graph, torch_optim, autograd, rays=None, config5=False, force_ddp=False):
        import raymarching
        import synthetic_scene as sc
        import ddp
        from nerf.network_ff import NeRFNetwork
        from ddp import GradientAverager
        from graph import GraphedTrainStep, mse_loss
        self.args, self.dev, self.world, self.rank, self.use_graph, self.torch_optim = args, dev, world, rank, graph, torch_optim
        # force_ddp: the data-parallel step (sharded update, collectives, occupancy exchange) on ONE rank over a 1-rank process group -- every
        # RCCL call executes, the exchange is the identity (`ddp_overhead_1rank`: the only way to run RCCL on a one-GPU box)
        ddp_on = self.ddp_on = world > 1 or bool(force_ddp)
        self.rays = int(rays if rays is not None else args.rays)   # rays per GPU and step
        torch.manual_seed(0)  # identical parameters on every rank (FFMLP reseeds to 42 itself)
        self.config5 = config5
        if config5:
            # BASELINE config 5's shape (Tanks&Temples-style, main_nerf.py:44-47,80): bound = 8 -> 4 cascades, dt_gamma = 1/128, background
            # model on a radius-32 sphere (2-D hash grid + nn.Linear head), nn.Linear sigma / colour networks (nerf/network.py: there is
            # no --ff background model in the reference), rays marched through all four cascades
            from nerf.network import NeRFNetwork as NeRFNetworkLinear
            model = NeRFNetworkLinear(bound=8, cuda_ray=True, bg_radius=32.0, density_scale=1, min_near=0.2, density_thresh=10).to(dev)
            grid = sc.occupancy_density(bound=8.0, cascade=4)
            grid = np.maximum(grid, np.where(np.random.default_rng(5).uniform(size=grid.shape) < 0.02, 30.0, 0.0).astype(np.float32))
            occ = torch.from_numpy(grid).to(dev)
        else:
            model = NeRFNetwork(bound=1, cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=10).to(dev)
            occ = torch.from_numpy(sc.occupancy_density()).to(dev)
        model.train()
        model.fused = fused
        model.density_grid.copy_(occ)
        model.density_bitfield = raymarching.packbits(model.density_grid, 10.0, model.density_bitfield)
        fixed_bits = model.density_bitfield.clone()
        model.iter_density = 16          # steady state: partial occupancy refreshes (renderer.py:488-514)
        model.mean_density = float(occ.clamp(min=0).mean())
        if torch_optim:
            # the reference's pair (main_nerf.py:132, nerf/utils.py:393): torch Adam (fused, capturable) + GradScaler
            optimizer = torch.optim.Adam(model.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15, fused=True, capturable=True)
            scaler = torch.amp.GradScaler('cuda')
            averager = GradientAverager(model, world) if ddp_on else None
        else:
            # same update rule and scale dynamics in one fused device-side step (torch-ngp_amd/optim.py)
            from optim import NGPAdam
            # N > 1: ZeRO-1-style sharded update (reduce-scatter -> Adam on 1/N -> all-gather of the fp16 shadows under the next march);
            # --replicated-optim selects the all-reduce + full update on every rank instead
            optimizer = NGPAdam(model.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15, world_size=world,
                                shard=('force' if force_ddp else True) if (ddp_on and not args.replicated_optim) else False,
                                verdict=getattr(args, 'shard_verdict', 'poison'))
            self.shard_fallback = None
            if optimizer.shard:
                # the three collectives of the sharded update, exercised once on zero gradients before anything is captured: if this
                # RCCL build refuses one of them (reduce_scatter_tensor with AVG on fp16, in-place all_gather_into_tensor), every rank sees
                # the same exception and the run continues with the replicated update instead of dying
                try:
                    optimizer.poison_shards()
                    optimizer.reduce_gradients()
                    optimizer.gather_shadows()
                    optimizer.wait_shadows()
                    torch.cuda.synchronize()
                except Exception as e:  # noqa: BLE001
                    self.shard_fallback = repr(e)[:200]
                    optimizer = NGPAdam(model.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15, world_size=world, shard=False)
            scaler = None
            averager = optimizer if ddp_on else None
        # a pool of pre-generated batches resident in HBM (one camera each, 4096 random pixels)
        self.n_pool = 16
        self.pool = []
        for k in range(self.n_pool):
            o, d, gt = sc.training_batch(self.rays, seed=1000 * rank + k)
            if config5:
                o = o * np.float32(2.0)   # cameras at twice the lego distance: rays cross cascades 0..3 of the bound-8 box
            self.pool.append((torch.from_numpy(o)[None].to(dev), torch.from_numpy(d)[None].to(dev), torch.from_numpy(gt).to(dev)))
        self.opt_kwargs = dict(staged=False, bg_color=1, perturb=True, force_all_rays=False, dt_gamma=0, max_steps=1024, T_thresh=1e-4)
        if config5:
            self.opt_kwargs.update(bg_color=None, dt_gamma=1.0 / 128)

        def keep_scene(m):
            # N > 1: the occupancy exchange of the data-parallel path (element-wise MAX of the grid + re-pack + common sample estimate)
            if ddp_on:
                ddp.sync_occupancy(m)
            # the synthetic scene keeps its analytic occupancy: the refresh work is done, its result is discarded
            # (`evolving`: it is NOT discarded -- the extra measurement `evolving_occupancy`, where the grid follows the network)
            if not getattr(self, 'evolving', False):
                m.density_grid.copy_(occ)
                m.density_bitfield.copy_(fixed_bits)
        self.keep_scene = keep_scene
        self.model, self.optimizer = model, optimizer
        # lookahead: the next batch's march on a side stream under this iteration -- single GPU and the sharded data-parallel step alike
        # (the march needs no weights).  Measured over a 1-rank RCCL group in a process of its own (ddp_overhead_1rank): three replays without
        # lookahead 0.566 ms, lookahead with eager collectives between [rest] and [apply] 0.529 ms, lookahead with BOTH collectives captured
        # inside the rest graph 0.483 ms (single GPU: 0.428).  The captured form is used when `args.graph_collectives` says so: set by
        # NGP_GRAPH_COLLECTIVES=1, or by main() after its canary (a short N-rank trial run of that form in child processes) came back clean
        # on every rank -- a captured RCCL call cannot be verified ahead of time on the one-GPU build box.
        graphed = bool(getattr(args, 'graph_collectives', False)) or os.environ.get('NGP_GRAPH_COLLECTIVES', '0') == '1'
        self.lookahead = (graph and fused and not torch_optim and not autograd and not getattr(args, 'no_lookahead', False)
                          and (not ddp_on or bool(getattr(optimizer, 'shard', False))))
        # single GPU: the hash table's Adam sweep rides in the grid backward's slice accumulate (speculative double buffer + parity word)
        self.fused_adam = (graph and fused and not torch_optim and not autograd and not ddp_on and not config5
                           and not getattr(args, 'no_fused_adam', False))
        self.stepper = GraphedTrainStep(model, optimizer, scaler, self.rays, self.opt_kwargs, loss_fn=mse_loss, averager=averager,
                                        after_update=keep_scene, direct=not autograd, lookahead=self.lookahead, fused_table_adam=self.fused_adam)
        if ddp_on:
            self.stepper.graph_collectives = graphed
        self.step_no = 0
        self.caps, self.slots = [], []
        self.count_log = None

    def train_step(self, count=True):
        stepper, model, args = self.stepper, self.model, self.args
        rays_o, rays_d, gt = self.pool[self.step_no % self.n_pool]
        self.step_no += 1
        if not self.use_graph:
            # eager path: same cadence, no capture
            if stepper.global_step % 16 == 0:
                with torch.autocast('cuda', dtype=torch.float16):
                    model.update_extra_state()
                self.keep_scene(model)
            mc = model.mean_count
            cap = mc + (128 - mc % 128) if mc > 0 else self.rays * 1024
            loss = stepper._eager(rays_o, rays_d, gt)
            stepper.global_step += 1
        else:
            if self.lookahead or self.ddp_on:
                # the data loader's next batch is known one step early: its march runs under this iteration (lookahead), or -- sharded
                # data-parallel step -- its copy into the static buffers rides behind this step's reduce-scatter
                nxt = self.pool[self.step_no % self.n_pool]
                loss = stepper.step(rays_o, rays_d, gt, next_rays=nxt)
            else:
                loss = stepper.step(rays_o, rays_d, gt)
            cap = stepper.capacity if stepper.capacity is not None else self.rays * 1024
        if count:
            # log the sample counts: the model keeps the last 16 in its counter ring (renderer.py:352), so one 128-byte device copy every
            # RING_BLOCK = 8 steps is enough (8, not 16: in lookahead mode the side stream hands the NEXT step's count to its ring slot
            # while this step runs, so a snapshot may only rely on slots younger than 15 steps); the clamp to the buffer capacity and
            # the sum happen after the timed region.  Samples that were marched AND evaluated: rays that do not fit the estimated buffer
            # are dropped whole by march_rays_train (raymarching.cu:416), so at most `cap` samples are processed in a step
            self.caps.append(cap)
            self.slots.append((model.local_step - 1) % 16)
            if len(self.caps) % RING_BLOCK == 0:
                self.count_log[len(self.caps) // RING_BLOCK - 1].copy_(model.step_counter, non_blocking=True)
        return loss

    def setup(self, warmup):
        """untimed, not part of --warmup: bring the run to the steady state of a running training.  The first 16 iterations size the
        sample buffer for the worst case and read the count back (raymarching.py:223-231); after the first update_extra_state the
        running estimate `mean_count` exists, the iteration is sync-free and (graph mode) is captured; every graph a later step replays
        is captured here and replayed at least once before the timed region starts."""
        for _ in range(17):
            self.train_step(count=False)
        if self.use_graph:
            self.stepper.precapture()
        while self.step_no < SETUP_ITERATIONS:
            self.train_step(count=False)
        for _ in range(warmup):
            self.train_step(count=False)

    def timed(self, steps):
        dev, world, model = self.dev, self.world, self.model
        self.count_log = torch.zeros(steps // RING_BLOCK + 2, 16, 2, dtype=torch.int32, device=dev)  # snapshots of the model's 16-slot counter ring
        self.caps, self.slots = [], []
        captures0 = self.stepper.captures
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        stamps = []
        for _ in range(steps):
            loss = self.train_step()
            if len(stamps) < 16:
                stamps.append(time.perf_counter() - t0)
        # host time to ISSUE the steps, nothing waited for: over the whole run it approaches `elapsed` once the launch queue is full (back
        # pressure); the first steps after the synchronisation show what the host itself needs per step
        issued = time.perf_counter() - t0
        deltas = sorted(b - a for a, b in zip(stamps[:-1], stamps[1:]))
        self.host_first_steps_ms = round(deltas[len(deltas) // 2] * 1e3, 4) if len(deltas) >= 3 else None   # median: refresh steps excluded
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        captures = self.stepper.captures - captures0
        final_loss = float(loss.item())
        caps, slots = self.caps, self.slots
        if len(caps) % RING_BLOCK:  # the ring still holds the steps since the last snapshot
            self.count_log[len(caps) // RING_BLOCK].copy_(model.step_counter)
        blocks = torch.arange(len(caps), device=dev) // RING_BLOCK
        marched = self.count_log[blocks, torch.tensor(slots, dtype=torch.int64, device=dev), 0].to(torch.int64)
        total = torch.minimum(marched, torch.tensor(caps, dtype=torch.int64, device=dev)).sum()
        el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
            dist.all_reduce(total, op=dist.ReduceOp.SUM)
        return {'elapsed': float(el.item()), 'samples': int(total.item()), 'final_loss': final_loss, 'captures': int(captures), 'issued': issued, 'host_first': self.host_first_steps_ms}

    def execution(self):
        st = self.stepper
        if not self.use_graph:
            return 'eager'
        if st.capture_error is not None:
            return f'eager (graph capture failed: {st.capture_error[:120]})'
        la = f'; next batch marched on a side stream under the iteration ({st.la_hits} steps so far)' if getattr(st, 'la', None) is not None else ''
        return f'hip-graph replay ({st.n_captures} iteration + {st.n_update_captures} refresh capture(s), all before the timed region){la}'


def sdf_encoder_mlp(dev, sizes=(1 << 18, 1 << 21), reps=12):
    """BASELINE config 4 (main_sdf.py --fp16 --ff, sdf/netowrk_ff.py:9-48): hash grid (L16 F2 T2^19, 16 -> 2048) + FFMLP(32 -> 64 x 3 -> 1), no ray
    marching -- the configuration that isolates the encoder / MLP rooflines.  Per batch size (2^18 = main_sdf.py:43's points per step,
    2^21 = testing/test_ffmlp.py:100) the four kernels of one training step are timed on uniform points with HIP events on the launch
    stream (median of `reps` launches, every launch on fresh uniform-random inputs already resident in HBM): grid_encode_forward,
    ffmlp_forward (training: stores activations), ffmlp_backward (+dL/dx), grid_encode_backward (record sort + exact slice accumulation).
    Algorithmic bytes / flops per point as SURVEY.md 8(d).  Untimed relative to the headline metric."""
    import ctypes
    import _ngp_capi as capi
    from encoding import get_encoder
    from ffmlp import FFMLP
    torch.manual_seed(0)
    enc, in_dim = get_encoder('hashgrid')
    enc = enc.to(dev)
    net = FFMLP(input_dim=in_dim, output_dim=1, hidden_dim=64, num_layers=3).to(dev)
    emb16 = enc.embeddings.detach().half().uniform_(-0.1, 0.1)
    w16 = net.weights.detach().half()
    offs = enc.offsets
    L, S, H = int(enc.num_levels), float(np.log2(enc.per_level_scale)), int(enc.base_resolution)
    nl = 3
    out = {'config': 'main_sdf.py --fp16 --ff: hashgrid L16 F2 T2^19 (16 -> 2048) + FFMLP 32 -> 64x3 -> 1 (output padded to 16), uniform points in [0,1]^3',
           'unit': 'us per launch (median), HIP events on the launch stream', 'batches': {}}

    def timed(fn, fresh=None):
        ts = []
        for i in range(reps + 2):
            if fresh is not None:
                fresh()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            fn()
            b.record()
            torch.cuda.synchronize()
            if i >= 2:
                ts.append(a.elapsed_time(b))
        return float(np.median(ts)) * 1e-3   # seconds

    st = capi.stream
    for B in sizes:
        x = torch.rand(B, 3, device=dev)
        enc_out = torch.empty(L, B, 2, device=dev, dtype=torch.half)
        fb = torch.empty(nl, B, 64, device=dev, dtype=torch.half)
        y = torch.empty(B, 16, device=dev, dtype=torch.half)
        gy = torch.zeros(B, 16, device=dev, dtype=torch.half)
        gy[:, 0] = (torch.randn(B, device=dev) * 0.1).half()
        bb = torch.empty(nl, B, 64, device=dev, dtype=torch.half)
        g_enc = torch.empty(L, B, 2, device=dev, dtype=torch.half)
        gw = torch.zeros(w16.numel(), device=dev, dtype=torch.half)
        g_emb = torch.zeros_like(emb16)
        arr, ws, nbytes = capi.grid_backward_workspace(offs, B, 3, 2, L, S, H, 0, False, capi.NGP_F16)

        def k_gf():
            capi.check(capi.lib.ngp_grid_encode_forward_sched(x.data_ptr(), emb16.data_ptr(), offs.data_ptr(), enc_out.data_ptr(), B, 3, 2, L, S, H, None, 0, 0,
                                                              0, capi.NGP_F16, 0.0, None, st()))

        def k_ff():
            capi.check(capi.lib.ngp_ffmlp_forward_ex(enc_out.data_ptr(), w16.data_ptr(), B, 32, 16, 64, nl, 0, 6, fb.data_ptr(), y.data_ptr(),
                                                     capi.NGP_FF_INPUT_PLANAR, st()))

        def k_fb():
            capi.check(capi.lib.ngp_ffmlp_backward_ex(gy.data_ptr(), enc_out.data_ptr(), w16.data_ptr(), fb.data_ptr(), B, 32, 16, 64, nl, 0, 6, 1,
                                                      bb.data_ptr(), g_enc.data_ptr(), gw.data_ptr(), capi.NGP_FF_INPUT_PLANAR | capi.NGP_FF_DX_PLANAR, st()))

        def k_gb():
            capi.check(capi.lib.ngp_grid_encode_backward_checked(g_enc.data_ptr(), x.data_ptr(), None, offs.data_ptr(), g_emb.data_ptr(), B, 3, 2, L, S, H,
                                                                 None, None, 0, 0, 0, capi.NGP_F16, 0.0,
                                                                 None if arr is None else ctypes.cast(arr, ctypes.c_void_p), capi.ptr(ws), nbytes, None, st()))
        rows = {}
        for name, fn, byts, flops, fresh in (
                ('grid_encode_forward', k_gf, 588.0, 0.0, lambda: x.uniform_()),
                ('ffmlp_forward', k_ff, _ff_fwd_bytes(nl), _ff_flops(nl), None),
                ('ffmlp_backward', k_fb, _ff_bwd_bytes(nl), 2.0 * _ff_flops(nl), None),
                ('grid_encode_backward', k_gb, 1100.0, 0.0, lambda: g_emb.zero_())):
            t = timed(fn, fresh)
            row = {'us': round(t * 1e6, 1), 'bytes_per_point': byts, 'achieved_GBps': round(B * byts / t / 1e9, 1),
                   'hbm_frac': round(B * byts / t / 1e9 / HBM_PEAK_GBS, 4)}
            if flops:
                row['mfma_TFLOPs'] = round(B * flops / t / 1e12, 2)
                row['mfma_frac'] = round(B * flops / t / 1e12 / MFMA_F16_PEAK_TFLOPS, 5)
            rows[name] = row
        total = sum(r['us'] for r in rows.values())
        out['batches'][str(B)] = {'kernels': rows, 'sum_us': round(total, 1), 'points_per_s_kernels_only': round(B / (total * 1e-6), 1)}
        del x, enc_out, fb, y, gy, bb, g_enc, ws
    return out


def cpu_baselines(args):
    """rank 0, N = 1: (a) the reference's pure-PyTorch device='cpu' path on the host cores this container may use, (b) the scalar-C
    oracle on one thread.  (a) runs in a subprocess with a hard timeout: an intra-op pool larger than the container's CPU quota can
    stall for minutes, and nothing here may hold up the GPU numbers."""
    import subprocess
    import oracle
    import synthetic_scene as sc
    from oracle.pipeline import time_cpu_baseline
    from oracle.torch_cpu import usable_cores
    half = max(2.0, args.cpu_seconds / 2)
    cores = usable_cores()
    cpu, tried = None, []
    for threads in sorted({cores, min(cores, 64), min(cores, 16)}, reverse=True):
        # glibc malloc tuned to keep the step's large temporaries mapped (otherwise page faults dominate: 3-9x slower, measured)
        env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads), HIP_VISIBLE_DEVICES='',
                   MALLOC_TRIM_THRESHOLD_='34359738368', MALLOC_MMAP_MAX_='0', MALLOC_TOP_PAD_='1073741824')
        try:
            out = subprocess.run([sys.executable, '-m', 'oracle.torch_cpu', str(args.rays), str(half), str(threads)], cwd=ROOT, env=env, capture_output=True,
                                 text=True, timeout=10 * half + 60)
            r = json.loads(out.stdout.strip().splitlines()[-1])
        except Exception as e:  # noqa: BLE001 -- timeout / crash: try fewer threads
            tried.append(f'{threads} threads: {type(e).__name__}')
            continue
        cpu = {'value': round(r['samples_per_s'], 1), 'unit': 'samples/s', 'cores': r['threads'], 'kind': 'port',
               'host_cores_available': os.cpu_count(), 'usable_cores': cores, 'ms_per_step': round(r['median_step_s'] * 1e3, 1),
               'sample': f"{r['steps']} timed training step(s) (median; 1 warm-up) of the reference's pure-PyTorch path (NeRFRenderer.run "
                         f"num_steps={r['num_steps']} upsample_steps=0, hashgrid + nn.Linear MLPs, fp32, Adam) restated in oracle/torch_cpu.py, "
                         f"{r['n_rays']} rays x {r['num_steps']} = {r['samples_per_step']} samples per step, {r['threads']} intra-op threads "
                         f"({cores} cores usable by this container of {os.cpu_count()} on the host; glibc malloc trim/mmap thresholds raised), {r['wall_s']:.0f} s wall"
                         + (f"; gave up on: {tried}" if tried else '')}
        break
    if cpu is None:
        cpu = {'value': None, 'unit': 'samples/s', 'cores': 0, 'kind': 'port', 'sample': f'pure-PyTorch path did not finish in time: {tried}'}
    # the other half of BASELINE's metric on the host cores: the reference's pure-PyTorch 800x800 inference frame (staged, 4096-ray batches)
    if cpu.get('cores'):
        env = dict(os.environ, OMP_NUM_THREADS=str(cpu['cores']), MKL_NUM_THREADS=str(cpu['cores']), HIP_VISIBLE_DEVICES='',
                   MALLOC_TRIM_THRESHOLD_='34359738368', MALLOC_MMAP_MAX_='0', MALLOC_TOP_PAD_='1073741824')
        try:
            out = subprocess.run([sys.executable, '-m', 'oracle.torch_cpu', '4096', str(half), str(cpu['cores']), 'render'], cwd=ROOT, env=env,
                                 capture_output=True, text=True, timeout=10 * half + 90)
            r = json.loads(out.stdout.strip().splitlines()[-1])
            cpu['render_800x800_ms'] = round(r['frame_ms'], 1)
            cpu['render_sample'] = (f"{r['batches_timed']} timed batches (median; 1 warm-up) of {r['rays_per_batch']} rays x {r['num_steps']} samples through the "
                                    f"reference's NeRFRenderer.render(staged=True, max_ray_batch=4096) -> run (pure PyTorch, fp32, eval, no_grad; "
                                    f"oracle/torch_cpu.py), batches spread over the GPU frame's rays; frame = median batch x {r['rays_per_frame']} / "
                                    f"{r['rays_per_batch']} = {r['batches_per_frame']} batches; {r['threads']} threads, {r['wall_s']:.0f} s wall")
        except Exception as e:  # noqa: BLE001
            cpu['render_800x800_ms'] = None
            cpu['render_sample'] = f'pure-PyTorch render did not finish: {type(e).__name__}'
    bits = oracle.packbits(sc.occupancy_density(), 10.0)
    q = time_cpu_baseline(bits, n_rays=args.rays, min_seconds=half, max_steps_timed=4)
    cpu['scalar_port'] = {'value': round(q['samples_per_s'], 1), 'unit': 'samples/s', 'cores': 1, 'kind': 'port',
                          'sample': f"{q['steps']} full oracle training step(s) of the cuda_ray-shaped workload (forward+backward, no optimiser) "
                                    f"of {args.rays} rays = {q['samples']} samples in {q['seconds']:.1f} s, one host thread"}
    return cpu


def ddp_overhead_1rank(args, dev, steps):
    """The data-parallel training step on the ONE GPU of this box, over a 1-rank RCCL process group: the sharded update of optim.NGPAdam
    (`shard='force'`: reduce_scatter_tensor(AVG, fp16) -> verdict -> Adam on the shard -> in-place all_gather_into_tensor on the
    communication stream), the occupancy exchange (two all-reduces per refresh) and the graph replays between the collectives all EXECUTE
    (the exchange is the identity at one rank), so their fixed cost per step is measured instead of projected: HIP-graph boundaries, RCCL
    launches, stream hand-overs.  Timed like the headline (same batches, same protocol), next to the single-GPU step WITHOUT the lookahead
    side stream (the apples-to-apples baseline: the data-parallel step cannot hide the march on a side stream).  What it cannot show is wire
    time: DESIGN.md section 7 adds that on top.  Runs after everything else (a process group changes the capture mode of later graphs)."""
    import socket
    out = {'backend': None, 'steps': steps}
    try:
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0))
            port = sk.getsockname()[1]
        os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)
        torch.cuda.synchronize()
        out['backend'] = f'{dist.get_backend()} (RCCL), {dist.get_world_size()} rank, all_reduce proof = {int(ones.item())}'

        def one(**kw):
            run = TrainingRun(args, dev, 1, 0, fused=True, graph=True, torch_optim=False, autograd=False, **kw)
            return run
        # single-GPU step without the lookahead side stream, same process, same moment
        saved = getattr(args, 'no_lookahead', False)
        args.no_lookahead = True
        base = one()
        base.setup(min(args.warmup, 16))
        r = base.timed(steps)
        out['single_gpu_no_lookahead_ms_per_step'] = round(r['elapsed'] / steps * 1e3, 4)
        del base
        args.no_lookahead = saved
        # ... and the single-GPU lookahead step with the SEPARATE Adam sweep: the data-parallel step cannot carry the table's sweep in its grid
        # backward (the gradient is exchanged first), so this -- not the round-6 headline, which saves ~9 us there -- is the step it is built from
        fa = getattr(args, 'no_fused_adam', False)
        args.no_fused_adam = True
        base = one()
        base.setup(min(args.warmup, 16))
        r = base.timed(steps)
        out['single_gpu_lookahead_separate_adam_ms_per_step'] = round(r['elapsed'] / steps * 1e3, 4)
        del base
        args.no_fused_adam = fa
        # (the variant that captures RCCL calls inside a HIP graph runs LAST: a refused capture must not disturb the others)
        for name, look, graphed, verdict in (('sharded_lookahead', True, False, 'poison'),
                                             ('sharded_3_replays_no_lookahead', False, False, 'poison'),
                                             ('sharded_3_replays_no_lookahead_verdict_allreduce', False, False, 'allreduce'),
                                             ('sharded_lookahead_collectives_in_graph', True, True, 'poison')):
            args.shard_verdict = verdict
            args.no_lookahead = not look
            args.graph_collectives = graphed
            run = one(force_ddp=True)
            run.setup(min(args.warmup, 16))
            r = run.timed(steps)
            st, opt = run.stepper, run.optimizer
            entry = {'ms_per_step': round(r['elapsed'] / steps * 1e3, 4), 'samples_per_s': round(r['samples'] / r['elapsed'], 1),
                     'sharded_optimizer': bool(getattr(opt, 'shard', False)), 'lookahead_hits': int(getattr(st, 'la_hits', 0)),
                     'main_stream_replays_per_step': (1 if graphed else 2) if getattr(st, 'la', None) is not None else 3,
                     'host_ms_per_step_unblocked': r.get('host_first'),
                     'captures_in_timed_region': r['captures'], 'capture_error': st.capture_error, 'final_loss': r['final_loss'],
                     'host_issue_ms_per_step': round(r['issued'] / steps * 1e3, 4)}
            # the two collectives alone, HIP events on the issuing stream, eager, after the timed region
            ev = {}
            for cname, fn in (('reduce_scatter_fp16_24MB', opt.reduce_gradients), ('all_gather_fp16_24MB', lambda: opt.gather_shadows(async_op=False))):
                ts = []
                for _ in range(6):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(); fn(); e1.record()
                    torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1))
                ev[cname] = round(float(np.median(ts[1:])), 4)
            entry['collective_ms_1rank'] = ev
            opt.wait_shadows()
            torch.cuda.synchronize()
            out[name] = entry
            del run, st, opt
        args.shard_verdict, args.no_lookahead, args.graph_collectives = 'poison', saved, False
        out['ddp_overhead_ms_per_step'] = round(out['sharded_lookahead']['ms_per_step'] - getattr(args, '_headline_ms', float('nan')), 4)
        out['ddp_overhead_ms_per_step_collectives_in_graph'] = round(out['sharded_lookahead_collectives_in_graph']['ms_per_step'] - getattr(args, '_headline_ms', float('nan')), 4)
        sep = out['single_gpu_lookahead_separate_adam_ms_per_step']
        out['ddp_overhead_vs_separate_adam_step_ms'] = {'eager_collectives': round(out['sharded_lookahead']['ms_per_step'] - sep, 4),
                                                        'collectives_in_graph': round(out['sharded_lookahead_collectives_in_graph']['ms_per_step'] - sep, 4),
                                                        'note': 'against the single-GPU step the N > 1 step is built from (separate k_adam over the WHOLE table: at N ranks '
                                                                'that sweep shrinks to 1 / N, 60 -> 8 us at N = 8); ddp_overhead_ms_per_step is against the headline, whose '
                                                                "table sweep rides in the grid backward (single GPU only)"}
        out['note'] = ('1-rank RCCL group in a process of its own: every collective and every graph boundary of the N > 1 step executes, wire time is zero; '
                       'ddp_overhead = sharded_lookahead (what bench.py --gpus N runs when its canary does not clear the captured collectives) - the headline '
                       'single-GPU step of this run; ..._collectives_in_graph = the form it runs when the canary clears them')
    except Exception as e:  # noqa: BLE001 -- a probe: the headline line must not depend on it
        import traceback
        out['error'] = repr(e)[:300]
        out['traceback_tail'] = traceback.format_exc()[-600:]
    finally:
        try:
            if dist.is_initialized():
                torch.cuda.synchronize()
                dist.destroy_process_group()
        except Exception:  # noqa: BLE001
            pass
    return out


def collectives_canary(args, world, rank, dev):
    """N > 1 over RCCL: may the sharded step run with its two collectives CAPTURED inside the HIP graph (one replay per step: 0.483 against
    0.529 ms at one rank)?  A captured RCCL call cannot be verified ahead of time on the one-GPU build box, and a hang inside a graph
    replay cannot be caught in-process -- so every rank starts a CHILD process (same RANK / WORLD_SIZE, its own rendezvous port) that runs
    a short trial of exactly that form: 17 eager steps, the capture, 24 replayed steps, then a cross-rank check that the shadow weights
    agree bit for bit.  Clean exit on EVERY rank within the timeout (all-reduce MIN of the verdicts) -> the real run uses the captured
    form; anything else (refused capture, hang -> the child is killed, disagreement) -> lookahead with eager collectives.  ~30-40 s."""
    import subprocess
    base = int(os.environ.get('MASTER_PORT', '29500'))
    port = 30000 + (base + 7919) % 20000
    env = {k: v for k, v in os.environ.items() if not k.startswith('TORCHELASTIC')}   # (the child ranks rendezvous on a store of their own)
    env.update(MASTER_PORT=str(port), MASTER_ADDR='127.0.0.1', NGP_GRAPH_COLLECTIVES='1')
    cmd = [sys.executable, os.path.abspath(__file__), '--gpus', str(world), '--canary', '--rays', str(args.rays), '--watchdog', '170']
    if args.force_ddp:
        cmd.append('--force-ddp')   # (the one-rank form: tests/test_gpu_rccl_one_rank.py drives the canary under a 1-rank launcher)
    detail = ''
    try:
        r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=200)
        ok = r.returncode == 0 and 'CANARY_OK' in r.stdout
        detail = (r.stdout.strip().splitlines() or [''])[-1][:200] if ok else f'rc {r.returncode}: {r.stderr[-240:]}'
    except subprocess.TimeoutExpired:
        ok, detail = False, 'trial did not finish within 200 s (killed)'
    except Exception as e:  # noqa: BLE001
        ok, detail = False, repr(e)[:200]
    flag = torch.tensor([1.0 if ok else 0.0], device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return bool(flag.item() > 0.5), {'this_rank_ok': ok, 'all_ranks_ok': bool(flag.item() > 0.5), 'detail': detail}


def canary_child(args, dev, world, rank):
    """the trial itself (child process of collectives_canary): the sharded lookahead step with captured collectives for 24 replayed steps"""
    args.graph_collectives, args.no_lookahead = True, False
    run = TrainingRun(args, dev, world, rank, fused=True, graph=True, torch_optim=False, autograd=False, force_ddp=args.force_ddp)
    n_steps = int(os.environ.get('NGP_BENCH_CANARY_STEPS', '24'))
    run.setup(4)
    res = run.timed(n_steps)
    st, opt = run.stepper, run.optimizer
    ok = st.capture_error is None and st.la is not None and st.la_apply is None and bool(getattr(opt, 'shard', False)) and res['final_loss'] == res['final_loss']
    opt.wait_shadows()
    torch.cuda.synchronize()
    digest = opt.flat_p16.view(torch.int16).to(torch.int64).sum().reshape(1)     # every rank holds the complete, all-gathered fp16 shadows
    both = [torch.zeros_like(digest) for _ in range(world)]
    dist.all_gather(both, digest)
    same = all(int(b.item()) == int(both[0].item()) for b in both)
    return ok and same, f"{res['elapsed'] / n_steps * 1e3:.4f} ms/step, capture_error={str(st.capture_error)[:80]}, shadows agree={same}"


def self_launch(args):
    """`python bench.py --gpus N` (N > 1) started WITHOUT a launcher: replace this process by
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free> bench.py <same arguments>`,
    one rank per GPU.  Fails loudly when the node has fewer than N devices (NGP_BENCH_SHARE_GPU=1, the one-GPU functional-test hook,
    lifts that check) -- it never falls back to measuring one GPU under an `n_gpus: N` request."""
    import socket
    have = torch.cuda.device_count()
    if have < args.gpus and os.environ.get('NGP_BENCH_SHARE_GPU') != '1':
        raise SystemExit(f'bench.py: --gpus {args.gpus} but only {have} GPU(s) are visible; refusing to benchmark fewer ranks than asked for')
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', NGP_BENCH_SELF_LAUNCHED='1')
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')   # dmabuf IPC: what RCCL needs on this host driver
    env.setdefault('OMP_NUM_THREADS', '4')
    sys.stdout.flush()
    os.execvpe(sys.executable, cmd, env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=1024)
    ap.add_argument('--warmup', type=int, default=64)
    ap.add_argument('--rays', type=int, default=4096, help='rays per GPU and step (reference default, main_nerf.py:26)')
    ap.add_argument('--roofline-kernel', default='grid_encode_backward', help='kernel reported as `roofline` (default: the dominant one)')
    ap.add_argument('--no-graph', action='store_true', help='issue every launch eagerly instead of replaying HIP graphs')
    ap.add_argument('--autograd', action='store_true', help='capture the iteration through torch.autograd instead of the autograd-free fused iteration')
    ap.add_argument('--no-fused', action='store_true', help='module-by-module network path (reference-style glue) instead of fused.py')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--torch-optim', action='store_true', help='torch.optim.Adam(fused) + GradScaler instead of optim.NGPAdam')
    ap.add_argument('--replicated-optim', action='store_true', help='N > 1: all-reduce + full Adam on every rank instead of the sharded update')
    ap.add_argument('--no-render', action='store_true', help='skip the 800x800 inference-frame timing')
    ap.add_argument('--no-lookahead', action='store_true', help='do not march the next batch on a side stream under the current iteration '
                    '(graph.GraphedTrainStep(lookahead=True): single rank, fused + graph + NGPAdam only)')
    ap.add_argument('--no-fused-adam', action='store_true', help="N = 1: keep the table's Adam sweep a launch of its own (k_adam) instead of the "
                    "speculative sweep inside the grid backward's slice accumulate (optim.NGPAdam.enable_table_fusion)")
    ap.add_argument('--no-dropin', action='store_true', help='skip the second, drop-in-surface-only measurement')
    ap.add_argument('--ab-off', default='', help='A/B measurement: comma-separated optional launch fusions of fused.py to switch OFF '
                    '(USE_FUSED_NETWORK, USE_FUSED_COMPOSITE, USE_FUSED_MID, USE_FUSED_SCAN, USE_FUSED_CHECK, USE_SLABS_IN_ACCUMULATE, USE_OVERWRITE_TABLE, ...); '
                    'recorded in config.fusions_off, every switch in config.fused_switches')
    ap.add_argument('--dropin-steps', type=int, default=64)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extra', action='store_true', help='skip the untimed extra workloads: sdf_encoder_mlp (BASELINE config 4) and tnt_bound8 (config 5)')
    ap.add_argument('--extra-steps', type=int, default=48)
    ap.add_argument('--cpu-seconds', type=float, default=20.0)
    ap.add_argument('--scaling', choices=('weak', 'strong'), default='weak',
                    help='N > 1: weak = --rays per GPU (global batch grows with N, the headline); strong = --rays in total, --rays / N per GPU')
    ap.add_argument('--no-strong', action='store_true', help='N > 1, --scaling weak: skip the secondary strong-scaling measurement (`strong_scaling`)')
    ap.add_argument('--strong-steps', type=int, default=128)
    ap.add_argument('--pmc', action='store_true', help='measure `traffic` (HBM bytes per launch) for every roofline row in THIS run: two short rocprofv3 --pmc '
                    'passes of this script (about a minute); default: the committed pass named in traffic_source')
    ap.add_argument('--force-ddp', action='store_true', help='--gpus 1 only: the HEADLINE run itself goes through the data-parallel step over a 1-rank RCCL group '
                    '(sharded update, collectives, occupancy exchange); the default line carries the same measurement as `ddp_overhead_1rank`')
    ap.add_argument('--no-ddp-probe', action='store_true', help='skip the `ddp_overhead_1rank` measurement (N = 1 only)')
    ap.add_argument('--canary', action='store_true', help='(internal) the N-rank trial run of the captured-collectives step, started by collectives_canary()')
    ap.add_argument('--no-canary', action='store_true', help='N > 1: do not try the captured-collectives form (lookahead with eager collectives)')
    ap.add_argument('--ddp-probe-only', action='store_true', help='(internal) run ONLY the 1-rank RCCL probe and print its JSON: the default run starts this in a '
                    'subprocess with a timeout, so that a collective library that hangs or aborts cannot take the headline line with it')
    ap.add_argument('--headline-ms', type=float, default=float('nan'), help='(internal, with --ddp-probe-only) ms / step of the single-GPU headline run')
    ap.add_argument('--ddp-steps', type=int, default=128)
    ap.add_argument('--shard-verdict', choices=('poison', 'allreduce'), default='poison',
                    help="sharded update: how the global skip verdict travels (optim.NGPAdam(verdict=...))")
    ap.add_argument('--watchdog', type=float, default=900.0, help='seconds after which a run that has not finished dumps every thread\'s stack to stderr '
                    'and exits non-zero (a hung collective must not hang the box); 0 disables')
    args = ap.parse_args()
    if args.watchdog > 0:
        import faulthandler
        faulthandler.dump_traceback_later(args.watchdog, exit=True)

    assert torch.cuda.is_available(), 'bench.py needs a GPU (the HIP extension has no CPU fallback)'
    if args.gpus > 1 and int(os.environ.get('WORLD_SIZE', '1') or 1) <= 1 and 'LOCAL_RANK' not in os.environ:
        self_launch(args)   # no launcher around this process: it becomes the launcher of N ranks (does not return)
    # ONE JSON line on stdout, and nothing else: libraries write banners through C stdio (RCCL prints its version block to stdout when the
    # first communicator is created, and libc flushes it at EXIT -- i.e. after our line).  File descriptor 1 is pointed at stderr for the
    # whole run; the result line goes to a private duplicate of the real stdout (`emit`).
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), 'w')
    os.dup2(2, 1)

    def emit(text):
        real_stdout.write(text + '\n')
        real_stdout.flush()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit(f'bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} rank(s)')
    fusions_off = [n for n in args.ab_off.split(',') if n]
    if fusions_off:
        import fused as _fused
        for n in fusions_off:
            assert n.startswith('USE_') and hasattr(_fused, n), f'--ab-off: unknown fusion switch {n}'
            setattr(_fused, n, False)
    # functional-test hooks (not used by the driver): NGP_BENCH_SHARE_GPU=1 lets several ranks share cuda:0 and
    # NGP_BENCH_BACKEND=gloo replaces RCCL, so the N>1 code path can be exercised on a one-GPU box
    if os.environ.get('NGP_BENCH_SHARE_GPU') == '1':
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    comm = {'backend': None, 'rccl_ranks': None}
    if args.force_ddp and world != 1:
        raise SystemExit('bench.py: --force-ddp is the one-rank form of the data-parallel step (--gpus 1)')
    if args.force_ddp:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
    if world > 1 or args.force_ddp:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        backend = os.environ.get('NGP_BENCH_BACKEND', 'nccl')  # 'nccl' is RCCL on ROCm
        if backend == 'nccl':
            if os.environ.get('NGP_BENCH_SHARE_GPU') != '1' and torch.cuda.device_count() < world:
                raise SystemExit(f'bench.py: {world} ranks but {torch.cuda.device_count()} visible GPU(s)')
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(backend)
        # proof that the collective library joined every rank: a SUM of ones over the data-parallel group, on the device
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)
        torch.cuda.synchronize()
        joined = int(ones.item())
        if joined != world or dist.get_world_size() != args.gpus:
            raise SystemExit(f'bench.py: all_reduce over {dist.get_backend()} saw {joined} rank(s), expected {world}')
        comm = {'backend': dist.get_backend() + (' (RCCL over xGMI)' if dist.get_backend() == 'nccl' else ' (functional-test backend, host memory)'),
                'rccl_ranks': dist.get_world_size() if dist.get_backend() == 'nccl' else 0, 'ranks': dist.get_world_size(),
                'self_launched': os.environ.get('NGP_BENCH_SELF_LAUNCHED') == '1'}
        dev_ids = [None] * world
        dist.all_gather_object(dev_ids, torch.cuda.current_device())
        comm['devices'] = dev_ids

    import _ngp_capi as capi
    import synthetic_scene as sc
    if args.canary:      # child of collectives_canary(): one short trial, one line, exit code = verdict
        ok, text = canary_child(args, dev, world, rank)
        emit(('CANARY_OK ' if ok else 'CANARY_FAILED ') + text)
        dist.barrier()
        dist.destroy_process_group()
        sys.exit(0 if ok else 3)
    canary = None
    # (NGP_BENCH_CANARY_ANY_BACKEND=1: functional-test hook -- the hand-over itself (child ranks on a port of their own, scrubbed environment,
    # MIN all-reduce of the verdicts, the kill after 200 s) at 8 ranks over gloo; gloo cannot be captured, so the verdict there is "eager")
    if ((world > 1 or os.environ.get('NGP_BENCH_FORCE_CANARY') == '1') and (comm.get('rccl_ranks') or os.environ.get('NGP_BENCH_CANARY_ANY_BACKEND') == '1')
            and not args.no_canary and not args.torch_optim and not args.replicated_optim and not args.no_fused
            and not args.no_graph and not args.autograd and not args.no_lookahead and os.environ.get('NGP_GRAPH_COLLECTIVES') is None):
        args.graph_collectives, canary = collectives_canary(args, world, rank, dev)
    if args.ddp_probe_only:
        args._headline_ms = args.headline_ms
        emit(json.dumps(ddp_overhead_1rank(args, dev, max(16, args.ddp_steps))))
        return
    timers = KernelTimers(capi)

    rays_per_gpu = args.rays if args.scaling == 'weak' else max(128, args.rays // world)
    run = TrainingRun(args, dev, world, rank, fused=not args.no_fused, graph=not args.no_graph, torch_optim=args.torch_optim,
                      autograd=args.autograd, rays=rays_per_gpu, force_ddp=args.force_ddp)
    run.setup(args.warmup)
    res = run.timed(args.steps)
    elapsed, samples = res['elapsed'], res['samples']
    model, stepper = run.model, run.stepper

    # roofline pass (untimed): HIP graphs cannot carry timing events, so the same iteration is run eagerly for a few steps
    # with HIP-event pairs around the named kernels, on the stream they are launched on, same batches, same state.
    roofs, traffic_source = [], None
    comm_events = {}
    if not args.no_roofline and world > 1 and not args.torch_optim:
        # N > 1: HIP-event pairs around the gradient exchange of the eager iterations below, on the stream each collective is issued on
        # (rank 0's view; with RCCL the pair brackets the collective's kernels, which are ordered with that stream)
        opt = run.optimizer

        def timed_call(name, fn):
            def call(*a, **k):
                if rank != 0:
                    return fn(*a, **k)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                out = fn(*a, **k)
                e1.record()
                comm_events.setdefault(name, []).append((e0, e1))
                return out
            return call
        if getattr(opt, 'shard', False):
            opt.reduce_gradients = timed_call('reduce_scatter_fp16_gradients', opt.reduce_gradients)
            opt._all_gather = timed_call('all_gather_fp16_shadows', opt._all_gather)
        else:
            opt.all_reduce = timed_call('all_reduce_fp16_gradients', opt.all_reduce)
    if not args.no_roofline:
        # every rank runs these iterations (the data-parallel exchange inside them is collective); only rank 0 times its kernels
        timers.enabled = rank == 0
        marched = []
        for k in range(min(16, max(4, args.steps))):
            rays_o, rays_d, gt = run.pool[(run.step_no + k) % run.n_pool]
            saved = (model.mean_count, model.local_step)
            if stepper.captured_capacity:
                model.mean_count = stepper.captured_capacity - 128
            timers.step = k
            slot = model.local_step % 16
            if stepper.used_direct and stepper.captured_capacity and not getattr(stepper, 'sharded', False) and run.stepper.averager is None:
                # the launches of the captured (autograd-free) iteration, issued eagerly: the rows are those of the timed step
                stepper.rays_o.copy_(rays_o.view_as(stepper.rays_o)); stepper.rays_d.copy_(rays_d.view_as(stepper.rays_d)); stepper.target.copy_(gt)
                stepper._iteration_front()
                stepper._iteration_back()
                marched.append(stepper.counter[0, 0].clone())
            else:
                stepper._eager(rays_o, rays_d, gt)
                marched.append(model.step_counter[slot, 0].clone())   # samples this iteration really marched (the launches are sized for the buffer)
            model.mean_count, model.local_step = saved
        torch.cuda.synchronize()
        timers.enabled = False
        if rank == 0:
            traffic, traffic_source, measured = None, None, False
            if args.pmc and world == 1:
                tail = [f for f in ('--no-lookahead', '--no-graph', '--no-fused', '--torch-optim', '--autograd', '--no-fused-adam') if f in sys.argv[1:]]
                traffic, traffic_source = measure_pmc_traffic(tail)
                measured = traffic is not None
            if traffic is None:
                pmc_note = traffic_source
                traffic, traffic_source = load_pmc_traffic()
                if args.pmc and pmc_note:
                    traffic_source = f'{traffic_source} [--pmc asked for but: {pmc_note}]'
            roofs = timers.summary(traffic, marched=[int(m.item()) for m in marched])
            for r in roofs:
                r['traffic_source'] = ((traffic_source if measured else
                                        f'{traffic_source} (committed rocprofv3 --pmc pass of the same workload; not measured in this run)')
                                       if r['traffic'] is not None else None)
    comm_ms = None
    if comm_events:
        torch.cuda.synchronize()
        nbytes = int(run.optimizer.flat_grad16.numel() * 2)
        comm_ms = {'message_bytes': nbytes, 'note': 'eager iterations after the timed region, rank 0, HIP events on the issuing stream'}
        for name, evs in comm_events.items():
            ms = [a.elapsed_time(b) for a, b in evs[1:]] or [a.elapsed_time(b) for a, b in evs]
            comm_ms[name] = {'ms': round(float(np.median(ms)), 4), 'calls': len(evs),
                             'algorithmic_GBps': round(nbytes * (world - 1) / world / (float(np.median(ms)) * 1e-3) / 1e9, 1)}
        for name in ('reduce_gradients', '_all_gather', 'all_reduce'):   # un-wrap: the instance attributes shadow the methods
            run.optimizer.__dict__.pop(name, None)
    if getattr(run.optimizer, 'shard', False):
        run.optimizer.wait_shadows()
        run.optimizer.gather_master()  # collective: every rank's fp32 master weights complete again (every rank renders with them below)
        torch.cuda.synchronize()

    render = None
    if not args.no_render:
        # second half of BASELINE.json's metric: wall time of one 800x800 inference frame through NeRFRenderer.run_cuda's eval branch
        # (renderer.py:322-367), same scene, same (randomly initialised) network.  Two bracketing cases: the random-init density
        # (~1 everywhere: no ray terminates early, every one of the ~43 M samples is evaluated) and the same network with
        # density_scale = 300 (opaque surfaces: rays saturate after a few samples, as in a trained scene).
        # N > 1 (strong scaling by construction: ONE frame): every rank renders its block of pixel rows, the [N/R, 4] blocks are
        # all-gathered (ddp.render_sharded); timed on all ranks between barriers, the MAX over ranks is reported.
        import ddp
        model.eval()
        o, d = sc.full_image_rays(seed=0)
        ro, rd = torch.from_numpy(o)[None].to(dev), torch.from_numpy(d)[None].to(dev)
        rkw = dict(staged=True, bg_color=1, perturb=False, dt_gamma=0, max_steps=1024, T_thresh=1e-4)
        render = {'unit': f'ms per 800x800 frame (640000 rays), {world} GPU(s)' + ('' if world == 1 else ', pixel rows sharded over the ranks, image all-gathered'),
                  'n_gpus': world, 'scaling': 'strong' if world > 1 else None}
        for name, scale in (('transparent_random_init', 1.0), ('opaque_density_scale_300', 300.0)):
            model.density_scale = scale
            ts = []
            for f in range(6):
                torch.cuda.synchronize()
                if world > 1:
                    dist.barrier()
                t1 = time.perf_counter()
                with torch.no_grad(), torch.autocast('cuda', dtype=torch.float16):
                    if world > 1:
                        ddp.render_sharded(model, ro, rd, **rkw)
                    else:
                        model.render(ro, rd, **rkw)
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t1) * 1e3)
            best = torch.tensor([min(ts[1:])], dtype=torch.float64, device=dev)
            if world > 1:
                dist.all_reduce(best, op=dist.ReduceOp.MAX)
            render[name] = round(float(best.item()), 2)
        if rank == 0 and not args.no_roofline:
            # whole-frame accounting: every stage of the loop (march, network, casts, composite, compaction) of one opaque and one transparent frame
            for name, scale in (('transparent_random_init', 1.0), ('opaque_density_scale_300', 300.0)):
                model.density_scale = scale
                try:
                    render['stages_' + name] = frame_stages(model, ro, rd, rkw)
                except Exception as e:  # noqa: BLE001 -- accounting only
                    render['stages_' + name] = {'error': repr(e)[:200]}
            # the inference kernels of one more opaque frame with HIP-event pairs (encoder and fused network launches of the eval loop; the
            # march / composite / compaction kernels of the loop are in the committed rocprofv3 summary, profiles/)
            model.density_scale = 300.0
            timers.suffix, timers.enabled = ' (800x800 render, opaque frame)', True
            # units per launch = the sample rows the iteration really EMITTED (rows with dt > 0, counted on the device right behind the march),
            # not the rows launched and not alive x n_step (VERDICT r5 "What's weak" 7: the launched rows inflated these rows by 1.4x)
            emitted = []

            def count_rows(deltas):
                timers.step = len(emitted)
                emitted.append((deltas[:, 0] > 0).sum())
            model._loop_iter_hook = count_rows
            try:
                with torch.no_grad(), torch.autocast('cuda', dtype=torch.float16):
                    model.render(ro, rd, **rkw)   # rank 0 alone, the whole frame: no collective in here
            finally:
                model._loop_iter_hook = None
            torch.cuda.synchronize()
            timers.suffix, timers.enabled = '', False
            emitted = [max(int(e.item()), 1) for e in emitted]
            for r in timers.summary(marched=emitted):
                if r['kernel'].endswith('opaque frame)'):
                    r['traffic_source'] = None
                    r['units'] = 'sample rows emitted by the march of the same iteration (device count of rows with dt > 0)'
                    r['rows_emitted_in_frame'] = int(sum(emitted))
                    roofs.append(r)
            # k_march_rays (the largest kernel of the opaque frame) says nothing against HBM: it is bound by its instruction stream.  Row:
            # the stage's HIP-event time of the probed frame above against the issue floor of its vector instructions -- SQ_INSTS_VALU per
            # frame from the committed rocprofv3 --pmc pass (profiles/*_march_rays_pmc.json, tools/bench_render.py under the counter) x 4
            # cycles per wave64 instruction / (1024 SIMDs x 2.4 GHz).
            st = render.get('stages_opaque_density_scale_300', {})
            march = [x for x in st.get('stages', []) if x['stage'] == 'march_rays'] if isinstance(st, dict) else []
            pmc = load_march_pmc()
            if march and pmc:
                floor_ms = pmc['valu_insts_per_frame'] * 4.0 / (1024 * 2.4e9) * 1e3
                roofs.append({'kernel': 'k_march_rays (800x800 render, opaque frame: all iterations)', 'bound': 'issue', 'achieved': round(floor_ms, 4),
                              'peak': march[0]['ms'], 'unit': 'ms of pure VALU issue vs ms measured', 'frac': round(floor_ms / max(march[0]['ms'], 1e-9), 4),
                              'traffic': None, 'avg_kernel_ms': round(march[0]['ms'] / max(march[0]['launch_groups'], 1), 4),
                              'launches': march[0]['launch_groups'], 'total_ms': march[0]['ms'], 'valu_insts_per_frame': pmc['valu_insts_per_frame'],
                              'hbm_frac': march[0]['frac_of_hbm_peak'], 'source': pmc['source'],
                              'note': 'frac = issue floor / measured stage time (1.0 = nothing but its own vector instructions); HBM says nothing here'})
        model.density_scale = 1
        model.train()

    strong = None
    if world > 1 and args.scaling == 'weak' and not args.no_strong:
        # secondary line: the SAME global batch as one GPU (--rays in total, --rays / N per rank), same protocol, shorter run
        s_rays = max(128, args.rays // world)
        s_run = TrainingRun(args, dev, world, rank, fused=not args.no_fused, graph=not args.no_graph, torch_optim=args.torch_optim,
                            autograd=args.autograd, rays=s_rays)
        s_run.setup(min(args.warmup, 16))
        s_res = s_run.timed(args.strong_steps)
        strong = {'scaling': 'strong', 'value': round(s_res['samples'] / s_res['elapsed'], 1), 'unit': 'samples/s', 'steps': args.strong_steps,
                  'ms_per_step': round(s_res['elapsed'] / args.strong_steps * 1e3, 4), 'rays_per_gpu_per_step': s_rays,
                  'global_rays_per_step': s_rays * world, 'samples_per_step_global': round(s_res['samples'] / args.strong_steps, 1),
                  'captures_in_timed_region': s_res['captures'], 'execution': s_run.execution()}
        if getattr(s_run.optimizer, 'shard', False):
            s_run.optimizer.wait_shadows()
            torch.cuda.synchronize()
        del s_run

    dropin = None
    if rank == 0 and world == 1 and not args.no_dropin and not (args.no_fused and args.no_graph and args.torch_optim):
        # the pure drop-in surface: what the reference's unchanged network_ff.py / renderer.py / Trainer would execute on this library
        d_run = TrainingRun(args, dev, 1, 0, fused=False, graph=False, torch_optim=True, autograd=True)
        d_run.setup(min(args.warmup, 8))
        d_res = d_run.timed(args.dropin_steps)
        dropin = {'value': round(d_res['samples'] / d_res['elapsed'], 1), 'unit': 'samples/s', 'steps': args.dropin_steps,
                  'ms_per_step': round(d_res['elapsed'] / args.dropin_steps * 1e3, 4),
                  'samples_per_step': round(d_res['samples'] / args.dropin_steps, 1),
                  'execution': 'eager launches, module-by-module network (model.fused = False), torch.optim.Adam(fused) + GradScaler',
                  'final_loss': d_res['final_loss']}
        del d_run

    sdf = tnt = evolving = None
    if rank == 0 and world == 1 and not args.no_extra:
        sdf = sdf_encoder_mlp(dev)
        # BASELINE config 5 as a measured workload: the bound-8 / 4-cascade / dt_gamma = 1/128 / background-model training step through the
        # drop-in modules (nerf/network.py nn.Linear networks: the reference has no --ff background model), torch Adam + GradScaler, the
        # iteration replayed from a HIP graph (autograd inside the capture)
        # (round 6: optim.NGPAdam instead of torch.optim.Adam(fused) + GradScaler -- five multi-tensor launches of ~38 us + the scaler's own per
        # step became three k_adam launches: 1.09 -> 0.94 ms / step on one box; the torch pair is timed beside it)
        tt_run = TrainingRun(args, dev, 1, 0, fused=False, graph=not args.no_graph, torch_optim=True, autograd=True, config5=True)
        tt_run.setup(4)
        tt_res = tt_run.timed(args.extra_steps)
        del tt_run
        t_run = TrainingRun(args, dev, 1, 0, fused=False, graph=not args.no_graph, torch_optim=False, autograd=True, config5=True)
        t_run.setup(4)
        t_res = t_run.timed(args.extra_steps)
        # the headline keeps the scene's analytic occupancy (every refresh is computed, its result discarded), so its graphs never see the
        # sample count move.  Here the refreshes STAND: the occupancy follows the (randomly initialised, training) density network from the
        # analytic start, the running sample estimate moves, and the captured capacity has to follow it -- re-captures inside the timed region
        # are expected and reported.  Not a steady state and not comparable with the headline: evidence that the capacity logic of
        # graph.GraphedTrainStep works under a changing grid (VERDICT r4 "Measurement" 9).
        e_run = TrainingRun(args, dev, 1, 0, fused=True, graph=not args.no_graph, torch_optim=False, autograd=False)
        e_run.setup(4)
        e_run.evolving = True
        e_caps0, e_sw0 = e_run.stepper.captures, e_run.stepper.n_switches
        e_res = e_run.timed(96)
        evolving = {'steps': 96, 'ms_per_step': round(e_res['elapsed'] / 96 * 1e3, 4), 'samples_per_step': round(e_res['samples'] / 96, 1),
                    'value': round(e_res['samples'] / e_res['elapsed'], 1), 'unit': 'samples/s', 'graph_captures_in_timed_region': e_res['captures'],
                    'captured_capacity_at_end': e_run.stepper.captured_capacity, 'mean_count_at_end': int(e_run.model.mean_count),
                    'capacity_switches_in_timed_region': e_run.stepper.n_switches - e_sw0, 'captured_capacities': sorted(e_run.stepper._captured),
                    'capture_error': e_run.stepper.capture_error, 'final_loss': e_res['final_loss'],
                    'note': 'the occupancy refreshes are kept (6 of them in 96 steps): the grid follows the density network; not a steady state.  '
                            'Round 6: precapture() records a ladder of capacities, the moving estimate SWITCHES between captured graphs instead of re-capturing'}
        del e_run
        tnt = {'config': 'Tanks&Temples-shaped: bound=8, 4 cascades x 128^3, dt_gamma=1/128, background model (radius-32 sphere, 2-D hashgrid + nn.Linear), '
                         'nn.Linear sigma/colour/background networks (nerf/network.py) evaluated on the fused-MLP kernels under autocast (fused_linear: one-hidden-layer stacks through an exact identity layer; '
                         'round 4: library GEMMs, 2.1 ms/step), --fp16 --cuda_ray, 4096 rays, optim.NGPAdam (round 6; torch.optim.Adam(fused) + GradScaler: torch_adam_ms_per_step), modules through autograd',
               'value': round(t_res['samples'] / t_res['elapsed'], 1), 'unit': 'samples/s', 'steps': args.extra_steps,
               'ms_per_step': round(t_res['elapsed'] / args.extra_steps * 1e3, 4),
               'torch_adam_ms_per_step': round(tt_res['elapsed'] / args.extra_steps * 1e3, 4),
               'samples_per_step': round(t_res['samples'] / args.extra_steps, 1), 'final_loss': t_res['final_loss'],
               'execution': t_run.execution(), 'captures_in_timed_region': t_res['captures']}
        del t_run

    ddp1 = None
    if rank == 0 and world == 1 and not args.force_ddp and not args.no_ddp_probe and not args.torch_optim and not args.no_fused and not args.no_graph:
        # in a SUBPROCESS with a timeout: a collective library that hangs or aborts must not take this run's line with it
        import subprocess
        torch.cuda.synchronize()
        n_probe = max(16, min(args.ddp_steps, max(args.steps, 64)))
        cmd = [sys.executable, os.path.abspath(__file__), '--ddp-probe-only', '--ddp-steps', str(n_probe), '--rays', str(args.rays), '--warmup', str(args.warmup),
               '--headline-ms', repr(elapsed / args.steps * 1e3), '--watchdog', '280']
        env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT')}
        try:
            pr = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
            out_lines = [ln for ln in pr.stdout.splitlines() if ln.startswith('{')]
            ddp1 = json.loads(out_lines[-1]) if out_lines else {'error': f'probe process ended with rc {pr.returncode}', 'stderr_tail': pr.stderr[-400:]}
        except subprocess.TimeoutExpired:
            ddp1 = {'error': 'the 1-rank RCCL probe did not finish within 300 s (hung collective?)'}
        except Exception as e:  # noqa: BLE001
            ddp1 = {'error': repr(e)[:300]}
    if rank == 0:
        roof = None
        for r in roofs:
            if r['kernel'] == args.roofline_kernel:
                roof = r
        if roof is None and roofs:
            roof = roofs[0]  # the dominant one by total time
        cpu = None
        if not args.no_cpu_baseline and world == 1:
            cpu = cpu_baselines(args)
        sharded = bool(getattr(run.optimizer, 'shard', False))
        fallback = getattr(run, 'shard_fallback', None)
        if world == 1 and not args.force_ddp:
            par = 'dp1'
        elif sharded:
            par = f'dp{world} (reduce-scatter, sharded Adam, all-gather of fp16 shadows)'
        else:
            par = f'dp{world} (all-reduce, replicated Adam' + ('; FALLBACK: the sharded collectives were refused by this backend' if fallback else '') + ')'
        line = {
            'metric': 'training samples/s (rays x steps), lego-shaped synthetic, fp16 autocast, full step incl. Adam',
            'value': round(samples / elapsed, 1), 'unit': 'samples/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(elapsed / args.steps * 1e3, 4), 'higher_is_better': True, 'scaling': args.scaling, 'vs_baseline': None,
            'dtype': 'f16', 'data': 'synthetic',
            'rccl_ranks': comm['rccl_ranks'], 'comm': comm, 'sharded_update_fallback': fallback,
            'collectives_in_graph': {'used': bool(getattr(run.stepper, 'graph_collectives', False) and getattr(run.stepper, 'la', None) is not None
                                                  and getattr(run.stepper, 'la_apply', None) is None) if world > 1 or args.force_ddp else None,
                                     'canary': canary},
            'config': {'workload': 'nerf_synthetic/lego-shaped --fp16 --cuda_ray --ff training step (hashgrid L=16 F=2 T=2^19, SH deg 4, '
                                   'FFMLP 64x2 / 64x3), bound=1, 128^3 occupancy grid, dt_gamma=0, max_steps=1024',
                       'rays_per_gpu_per_step': run.rays, 'global_rays_per_step': run.rays * world,
                       'samples_per_step_per_gpu': round(samples / args.steps / world, 1),
                       'rays_per_s': round(run.rays * world * args.steps / elapsed, 1), 'parallelism': par,
                       'sharded_update_fallback': fallback,
                       'execution': run.execution(), 'setup_iterations_untimed': SETUP_ITERATIONS, 'fusions_off': fusions_off,
                       'fused_switches': _fused_switches(),
                       'captures_in_timed_region': res['captures'],
                       'host_issue_ms_per_step': round(res['issued'] / args.steps * 1e3, 4), 'host_ms_per_step_unblocked': res.get('host_first'),
                       'autograd_free_iteration': bool(stepper.used_direct), 'fused_pipeline': bool(model.fused),
                       'table_adam_in_grid_backward': bool(getattr(stepper, 'table_fused', False)),
                       'optimizer': 'torch.optim.Adam(fused)+GradScaler' if args.torch_optim else 'optim.NGPAdam (fused Adam + loss scaling)',
                       'final_loss': res['final_loss']},
            'roofline': roof, 'rooflines': roofs, 'cpu_baseline': cpu, 'dropin_path': dropin, 'render_800x800_ms': render,
            'strong_scaling': strong, 'collectives': comm_ms, 'sdf_encoder_mlp': sdf, 'tnt_bound8': tnt, 'evolving_occupancy': evolving, 'ddp_overhead_1rank': ddp1,
        }
        emit(json.dumps(line))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
