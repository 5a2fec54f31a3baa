"""Where the pipelined step goes, from a rocprofv3 kernel trace of tools/pipeline_run.py: in the steady-state window (the middle half of
the last run of launches) per kernel its launches per step, mean duration (next to other slots' kernels), workgroups, and CU time =
duration x min(workgroups, 256) / 256; per hardware queue its busy share and the gaps between consecutive kernels; the number of
kernels running at once; and the CU time of a step against the measured step.
usage: python tools/pipeline_occupancy.py <kernel_trace.csv> [steps]"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps_hint = int(sys.argv[2]) if len(sys.argv) > 2 else 0
short = lambda n: re.sub(r'\(.*$', '', re.sub(r'\(anonymous namespace\)::', '', re.sub(r'^void ', '', n)))[:44]
ks = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name']), r.get('Queue_Id'),
             max(1, int(r['Grid_Size_X']) * int(r.get('Grid_Size_Y', 1) or 1) * int(r.get('Grid_Size_Z', 1) or 1)
                 // max(1, int(r['Workgroup_Size_X']) * int(r.get('Workgroup_Size_Y', 1) or 1) * int(r.get('Workgroup_Size_Z', 1) or 1))))
            for r in rows)
fit_starts = [k[0] for k in ks if k[2].startswith('fit_kernel')]
n_timed = steps_hint or (len(fit_starts) - 16)
w0, w1 = fit_starts[-n_timed + n_timed // 4], fit_starts[-n_timed // 4]     # the middle half of the timed loop (the run's last launches)
win = [k for k in ks if k[0] >= w0 and k[1] <= w1]
span = (w1 - w0) / 1e3
fits = sum(1 for k in win if k[2].startswith('fit_kernel'))
print(f'window {span:.0f} us, {len(win)} launches, {fits} fit launches (= batches) -> {span / max(fits, 1):.1f} us per batch')
by = collections.defaultdict(list)
for s, e, n, q, wg in win:
    by[n].append(((e - s) / 1e3, wg))
tot = 0.0
print(f'{"kernel":46s} {"/batch":>6s} {"avg us":>8s} {"wgs":>6s} {"CU-us/batch":>11s}')
for n, v in sorted(by.items(), key=lambda kv: -sum(d * min(w, 256) / 256 for d, w in kv[1])):
    cu = sum(d * min(w, 256) / 256 for d, w in v) / max(fits, 1)
    tot += cu
    print(f'{n:46s} {len(v) / max(fits, 1):6.2f} {sum(d for d, _ in v) / len(v):8.2f} {sum(w for _, w in v) / len(v):6.0f} {cu:11.2f}')
print(f'sum of CU time per batch {tot:.1f} us (workgroups / 256 x duration: an UPPER bound for kernels whose workgroups share CUs) vs {span / max(fits, 1):.1f} us per batch')
byq = collections.defaultdict(list)
for k in win:
    byq[k[3]].append(k)
for q, v in sorted(byq.items()):
    busy = sum(e - s for s, e, *_ in v) / 1e3
    gaps = sorted((v[i + 1][0] - v[i][1]) / 1e3 for i in range(len(v) - 1))
    print(f'queue {q}: {len(v)} launches, busy {100 * busy / span:.0f} %, gaps: median {gaps[len(gaps) // 2]:.2f} us, mean {sum(gaps) / len(gaps):.2f} us, sum {sum(gaps) / max(fits, 1) * len(byq):.0f} us per batch of this queue')
ev = sorted([(s, 1) for s, *_ in win] + [(e, -1) for _, e, *_ in win])
cur, last, hist = 0, w0, collections.defaultdict(float)
for t, d in ev:
    hist[cur] += t - last
    last, cur = t, cur + d
print('kernels running at once: ' + ', '.join(f'{k}: {100 * v / sum(hist.values()):.0f} %' for k, v in sorted(hist.items())))
