#!/usr/bin/env python3
"""Per-kernel SQ counter summary from a `rocprofv3 --pmc SQ_... --output-format csv` run: where do the wave cycles go?
    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_LDS \\
        --output-format csv -d gpurun_out/sq -o run -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-graph --no-roofline --no-render
    python tools/pmc_sq.py gpurun_out/sq
WAIT_ANY = wave parked (s_waitcnt / barrier), WAIT_INST_ANY = issue stall, ACTIVE_INST_ANY = issuing (MI355X_MICROARCH.md)."""
import collections, csv, glob, os, sys


def main():
    f = glob.glob(os.path.join(sys.argv[1], '**', '*counter_collection.csv'), recursive=True)[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.Counter()
    for r in csv.DictReader(open(f)):
        name = r['Kernel_Name'].split('(')[0][:48]
        acc[name][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == 'SQ_WAVE_CYCLES':
            calls[name] += 1
    rows = sorted(acc.items(), key=lambda kv: -kv[1].get('SQ_WAVE_CYCLES', 0))[:14]
    print(f"{'kernel':48s} {'calls':>5s} {'wait%':>6s} {'stall%':>6s} {'issue%':>6s} {'lds-stall%':>10s} {'VALU/call':>10s} {'MFMA/call':>10s} {'LDS/call':>9s}")
    for name, c in rows:
        wc = c.get('SQ_WAVE_CYCLES', 1.0) or 1.0
        n = max(calls[name], 1)
        print(f"{name:48s} {n:5d} {100 * c.get('SQ_WAIT_ANY', 0) / wc:6.1f} {100 * c.get('SQ_WAIT_INST_ANY', 0) / wc:6.1f} "
              f"{100 * c.get('SQ_ACTIVE_INST_ANY', 0) / wc:6.1f} {100 * c.get('SQ_WAIT_INST_LDS', 0) / wc:10.1f} "
              f"{c.get('SQ_INSTS_VALU', 0) / n:10.0f} {c.get('SQ_INSTS_MFMA', 0) / n:10.0f} {c.get('SQ_INSTS_LDS', 0) / n:9.0f}")


if __name__ == '__main__':
    main()
