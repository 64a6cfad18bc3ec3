#!/usr/bin/env python3
"""Markdown summary of a `rocprofv3 --kernel-trace --stats --output-format csv` run (the *kernel_stats.csv it writes):

    python tools/profile_summary.py gpurun_out/prof 'python bench.py --steps 64 --warmup 16 ...' > profiles/rNN_..._summary.md
"""
import csv, glob, os, sys


def main():
    f = glob.glob(os.path.join(sys.argv[1], '**', '*kernel_stats.csv'), recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    total = sum(float(r['TotalDurationNs']) for r in rows)
    print(f'# rocprofv3 --kernel-trace --stats : {sys.argv[2]} (1x MI355X)\n')
    print(f'total kernel time {total / 1e6:.1f} ms\n')
    print('| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|')
    for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:40]:
        print(f"| `{r['Name'][:100]}` | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.2f} | {float(r['AverageNs']) / 1e3:.1f} | "
              f"{100 * float(r['TotalDurationNs']) / total:.1f} |")


if __name__ == '__main__':
    main()
