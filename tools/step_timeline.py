#!/usr/bin/env python3
"""kernel timeline of the last steady training steps from a `rocprofv3 --kernel-trace --output-format csv` run of bench.py:

    python tools/step_timeline.py <dir with *kernel_trace.csv> [steps=2] [anchor kernel substring = k_adam]

Prints, for the window that starts at the (steps+1)-th last launch of the anchor kernel and ends at the last one: start / end (us from the
first row), duration, queue, kernel name.  (What profiles/rNN_step_timeline.txt is made from.)"""
import csv, glob, os, sys

d = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
anchor = sys.argv[3] if len(sys.argv) > 3 else 'k_adam'
f = glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if anchor in r['Kernel_Name']]
skip = int(sys.argv[4]) if len(sys.argv) > 4 else 0   # anchors to drop at the end (e.g. the eager roofline pass)
idx = idx[:len(idx) - skip] if skip else idx
lo, hi = idx[-(steps + 1)], idx[-1]
t0 = int(rows[lo]['Start_Timestamp'])
queues = {}
for r in rows[lo:hi + 1]:
    q = queues.setdefault(r['Queue_Id'], f'q{len(queues) + 1}')
    s, e = (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3
    print(f'{s:9.1f} {e:9.1f} {e - s:7.1f}  {q}  {r["Kernel_Name"][:96]}')
a = [int(rows[i]['Start_Timestamp']) for i in idx[-(steps + 1):]]
print('# anchor to anchor (us):', [round((y - x) / 1e3, 1) for x, y in zip(a[:-1], a[1:])])
