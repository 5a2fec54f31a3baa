#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, as MI355X_MICROARCH.md prescribes) into
profiles/hbm_traffic.json: HBM bytes per launch for every kernel of the hot path.

    python profiles/collect_traffic.py <fetch counter_collection.csv> <write counter_collection.csv>

Corrections (MI355X_MICROARCH.md, HBM / rocprofv3 section): FETCH_SIZE and WRITE_SIZE are in KiB-like units of
1024 B; on gfx950 FETCH_SIZE reports exactly HALF of the bytes of wide coalesced streaming reads (128-B requests
tallied at 64 B), so the read side is doubled; WRITE_SIZE is exact for 16-B-per-lane stores (ours are 8-16 B/lane)."""
import collections
import csv
import json
import re
import sys


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    name = re.sub(r'\(.*$', '', name)
    # the last template argument of the conv kernels is the element type (false = bf16, true = fp16); the library's
    # profile names (spsp_profile_kernel_name) leave it out
    name = re.sub(r'<(false|true)>$', '', name)
    name = re.sub(r', (false|true)>$', '>', name)
    return name


def per_kernel(path):
    agg = collections.defaultdict(list)
    for row in csv.DictReader(open(path)):
        agg[short(row['Kernel_Name'])].append(float(row['Counter_Value']))
    return {k: sum(v) / len(v) for k, v in agg.items()}


fetch, write = per_kernel(sys.argv[1]), per_kernel(sys.argv[2])
out = {}
for k in sorted(set(fetch) | set(write)):
    if not any(t in k for t in ('conv', 'fit_kernel', 'smpl_', 'fc_kernel', 'maxpool', 'rot6d')):
        continue
    rd, wr = 2 * 1024 * fetch.get(k, 0.0), 1024 * write.get(k, 0.0)
    out[k] = {'hbm_bytes_per_launch': rd + wr, 'read_bytes': rd, 'write_bytes': wr}
json.dump(out, open('profiles/hbm_traffic.json', 'w'), indent=1)
for k, v in out.items():
    print(f'{k:48s} read {v["read_bytes"] / 1e6:8.2f} MB  write {v["write_bytes"] / 1e6:8.2f} MB')
