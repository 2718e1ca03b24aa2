#!/usr/bin/env python3
"""Timeline of the last TIMED bench step from a rocprofv3 --kernel-trace CSV: kernels in launch order with their
durations (us) and the gap to the previous kernel, one line per suffix-sort round."""
import csv
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
def short(n):
    n = n.split('(')[0].replace('cjs::', '').replace('void ', '')
    return n.split('<')[0] + ('<' + n.split('<')[1][:12] if '<' in n else '')
names = [short(r['Kernel_Name']) for r in rows]
# bench.py's last step is the untimed one with per-stage synchronisation: take the step in front of it (a timed one)
marks = [i for i, n in enumerate(names) if n.startswith('rle_tile_summary')]
start, stop = (marks[-2], marks[-1]) if len(marks) > 1 else (marks[-1], len(rows))
line, t_prev, total, gaps = [], None, 0.0, 0.0
for r, n in list(zip(rows, names))[start:stop]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    d = (e - s) / 1e3
    gap = (s - t_prev) / 1e3 if t_prev is not None else 0.0
    t_prev = e
    total += d; gaps += max(gap, 0.0)
    if (n.startswith('bwt_gather_keys') or (n.startswith('bwt_tile_sort') and not any(x.startswith('gather_keys') for x in line))
            or n.startswith('mtf_head_tiles') and line and not line[-1].startswith('mtf')):
        print('  '.join(line)); line = []
    line.append('%s %.0f%s' % (n.replace('bwt_', '').replace('rs_', 'r:'), d, ('(+%.0f)' % gap) if gap > 8 else ''))
print('  '.join(line))
print('last timed step: kernel time %.2f ms, gaps %.2f ms' % (total / 1e3, gaps / 1e3))
