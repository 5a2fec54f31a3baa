"""Gaps between consecutive kernels of a hardware queue in the pipelined (HIP-graph) timed region, from a rocprofv3 kernel trace:
usage: python tools/graph_gaps.py <kernel_trace.csv>     (rocprofv3 --kernel-trace --output-format csv -- python3 bench.py ...)"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
byq = collections.defaultdict(list)
for r in rows:
    byq[(r.get('Queue_Id'), r.get('Stream_Id', ''))].append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
short = lambda n: re.sub(r'\(.*$', '', re.sub(r'\(anonymous namespace\)::', '', re.sub(r'^void ', '', n)))[:40]
allgaps = collections.defaultdict(list)
for q, ks in byq.items():
    ks.sort()
    if len(ks) < 200:
        continue
    tail = ks[len(ks) // 2:]                      # the later half: the timed region and what follows it
    busy = sum(e - s for s, e, _ in tail)
    span = tail[-1][1] - tail[0][0]
    gaps = [(tail[i + 1][0] - tail[i][1], short(tail[i][2]), short(tail[i + 1][2])) for i in range(len(tail) - 1)]
    gaps = [g for g in gaps if g[0] < 200_000]    # (drop the pauses between phases of the script)
    print(f'queue {q}: {len(tail)} kernels, busy {100 * busy / span:.0f} % of {span / 1e3:.0f} us; gap median {sorted(g[0] for g in gaps)[len(gaps) // 2] / 1e3:.2f} us, '
          f'mean {sum(g[0] for g in gaps) / len(gaps) / 1e3:.2f} us')
    for g, a, b in gaps:
        allgaps[(a, b)].append(g)
print('mean gap by (kernel -> next kernel), us:')
for k, v in sorted(allgaps.items(), key=lambda kv: -sum(kv[1]))[:24]:
    print(f'  {k[0]:40s} -> {k[1]:40s} {sum(v) / len(v) / 1e3:7.2f} us  x {len(v)}')
