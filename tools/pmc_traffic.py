#!/usr/bin/env python3
"""HBM traffic per launch of the hot-path kernels from rocprofv3 PMC passes (run on the GPU box, one pass per counter group, as
/opt/skills/guides/MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE do not fit one pass, and no tracing domains are
combined with --pmc):

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -o run -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline ...
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -o run -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline ...
    python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write ["<how bench.py ran>"] > profiles/rNN_pmc_traffic.json
(tools/final_evidence.sh runs the two passes in bench.py's DEFAULT mode: HIP-graph replay with the lookahead march on the side stream)

Units and corrections (guide, section HBM): the counters are in KiB-like units (bytes = value * 1024); on gfx950 FETCH_SIZE reports
exactly half of the bytes of a wide (16 B per lane) coalesced streaming read, so the read side of the streaming kernels (ffmlp,
composite) is doubled; the 4-byte gather/scatter kernels of the grid encoder are reported raw (uncalibrated width) and flagged.
bench.py reads the newest profiles/*_pmc_traffic.json to fill `roofline.traffic`."""
import collections, csv, glob, json, os, sys

KERNELS = {  # substring of the kernel name (first match wins) -> (label used by bench.py, read-side correction factor)
    # grid_encode_backward is three launches (atomic levels, record sort, slice accumulate): their traffic is summed under one label
    'k_grid_backward_bin': ('grid_encode_backward', 1.0),
    'k_grid_backward_accumulate': ('grid_encode_backward', 2.0),
    'k_grid_backward': ('grid_encode_backward', 1.0),
    'k_grid_forward_pair': ('grid_encode_forward', 1.0),
    'k_grid_forward_fast': ('grid_encode_forward', 1.0),   # round 5: the instant-ngp shape (4 / 8-byte gathers: raw)
    'k_network_forward': ('network_forward', 2.0),  # sigma MLP + exp/SH + colour MLP + sigmoid in one launch
    'k_ffmlp_forward': ('ffmlp_forward', 2.0),
    # the paired backward of the training step: <WIDTH 64, IN_JB 1, NHM 2> = the colour network (3 layers), NHM 1 = the sigma network
    'k_ffmlp_backward_pairedILi64ELi1ELi2E': ('ffmlp_backward (colour net)', 2.0),
    'k_ffmlp_backward_paired<64, 1, 2': ('ffmlp_backward (colour net)', 2.0),
    'k_ffmlp_backward': ('ffmlp_backward', 2.0),
    'k_march_train_wave': ('march_rays_train (+ near/far)', 1.0),
    # (round 6: what is left of the optimizer step when the table's Adam sweep rides in k_grid_backward_accumulate; listed BEFORE 'k_adam': first match wins)
    'k_adam_small_commit': ('k_adam_small_commit (dense table levels + MLP weights + scaler commit + parity flip)', 2.0),
    'k_adam': ('k_adam (Adam + scaler + shadows + gradient zeroing)', 2.0),   # 16-byte streaming reads: doubled
    'k_composite_train_loss_bwd': ('composite + loss + backward', 2.0),
    'k_composite_train_fwd': ('composite_rays_train_forward', 2.0),
    'k_composite_train_bwd': ('composite_rays_train_backward', 2.0),
}


def per_kernel(directory, counter):
    f = glob.glob(os.path.join(directory, '**', '*counter_collection.csv'), recursive=True)[0]
    vals = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] != counter:
            continue
        for key in KERNELS:
            if key in r['Kernel_Name']:
                vals[key].append((int(r['Grid_Size']), float(r['Counter_Value'])))
                break
    out = {}
    for key, lst in vals.items():
        sizes = collections.Counter(g for g, _ in lst)
        mode = sizes.most_common(1)[0][0]  # the training-step launches (update_extra_state uses other sizes)
        sel = [v for g, v in lst if g == mode]
        out[key] = (sum(sel) / len(sel), len(sel))
    return out


def annotate(d):
    """notes that need the numbers of BOTH grid-backward kernels (also applied to an existing file:  pmc_traffic.py --annotate file.json)"""
    det = d.get('detail', {})
    acc, srt = det.get('k_grid_backward_accumulate'), det.get('k_grid_backward_bin')
    if acc and srt and acc.get('read_correction') == 2.0:
        raw, stream = acc['fetch_bytes_raw'], srt['write_bytes']
        known = stream + 4.4e6 + 24.5e6   # records the sort wrote + descriptors + the gradient entries it may touch
        total = d['per_launch'].get('grid_encode_backward', 0)
        acc['note'] = ('FETCH_SIZE doubled as for 16 B/lane streaming reads (MI355X_MICROARCH.md HBM section) -- an UPPER bound here: the kernel reads '
                       '12 B per lane (pair units), a width the guide does not calibrate.  Known input of the kernel: the record stream the sort wrote '
                       f'(WRITE_SIZE of k_grid_backward_bin, {stream / 1e6:.0f} MB) + 4.4 MB of descriptors + the touched gradient entries (<= 24.5 MB) '
                       f'~= {known / 1e6:.0f} MB, i.e. a factor of ~{known / raw:.2f} on the raw {raw / 1e6:.0f} MB; with that factor grid_encode_backward '
                       f'moves ~{(total - 2.0 * raw + known) / 1e6:.0f} MB per launch instead of the {total / 1e6:.0f} MB reported')
    return d


def main():
    if len(sys.argv) == 3 and sys.argv[1] == '--annotate':
        d = annotate(json.load(open(sys.argv[2])))
        json.dump(d, open(sys.argv[2], 'w'), indent=1)
        return
    fetch = per_kernel(sys.argv[1], 'FETCH_SIZE')
    write = per_kernel(sys.argv[2], 'WRITE_SIZE')
    per_launch, detail = collections.defaultdict(int), {}
    for key in sorted(set(fetch) | set(write)):
        label, factor = KERNELS[key]
        rd = fetch.get(key, (0.0, 0))[0] * 1024.0
        wr = write.get(key, (0.0, 0))[0] * 1024.0
        per_launch[label] += round(rd * factor + wr)
        detail[key] = {'label': label, 'fetch_bytes_raw': round(rd), 'read_correction': factor, 'write_bytes': round(wr),
                       'launches_averaged': fetch.get(key, (0, 0))[1],
                       'note': 'raw FETCH_SIZE, 4-byte gather/scatter width uncalibrated' if factor == 1.0 else
                               'FETCH_SIZE doubled (16 B/lane streaming reads are tallied at half, MI355X_MICROARCH.md HBM section)'}
    json.dump(annotate({'per_launch': per_launch, 'detail': detail, 'unit': 'bytes of HBM traffic per kernel launch',
                        'source': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on ' +
                                  (sys.argv[3] if len(sys.argv) > 3 else 'python bench.py (default mode: HIP-graph replay, lookahead march on the side stream)')}),
              sys.stdout, indent=1)


if __name__ == '__main__':
    main()
