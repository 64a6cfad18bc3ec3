#!/usr/bin/env python3
"""mean per launch of every counter in one or more `rocprofv3 --pmc ... --output-format csv` output directories, for the kernels whose
name contains <substring>:   python tools/pmc_kernel.py <substring> <dir> [<dir> ...]  -> JSON on stdout"""
import collections, csv, glob, json, os, sys

sub = sys.argv[1]
acc, n = collections.defaultdict(float), collections.Counter()
for d in sys.argv[2:]:
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            if sub in r['Kernel_Name']:
                acc[r['Counter_Name']] += float(r['Counter_Value'])
                n[r['Counter_Name']] += 1
print(json.dumps({k: {'per_launch': acc[k] / n[k], 'launches': n[k]} for k in sorted(acc)}, indent=1))
