#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, as MI355X_MICROARCH.md prescribes) into
profiles/hbm_traffic.json: HBM bytes per launch for every kernel of the hot path.

    python profiles/collect_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> [<fetch> <write> of a plan-"throughput" run]

(the second pair supplies the kernels only plan "throughput" launches; a kernel both runs launch keeps the first pair's bytes)

Corrections (MI355X_MICROARCH.md, HBM / rocprofv3 section): FETCH_SIZE and WRITE_SIZE are in KiB-like units of
1024 B; on gfx950 FETCH_SIZE reports exactly HALF of the bytes of wide coalesced streaming reads (128-B requests
tallied at 64 B), so the read side is doubled; WRITE_SIZE is exact for 16-B-per-lane stores (ours are 8-16 B/lane)."""
import collections
import csv
import json
import os
import re
import sys


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    name = re.sub(r'\(.*$', '', name)
    return name.strip()


def split_template(name):
    m = re.match(r'^([^<]+)<(.*)>$', name)
    if not m:
        return name, []
    return m.group(1), [a.strip() for a in m.group(2).split(',')]


def library_names():
    """The kernel names of the library's own table (spsp_profile_kernel_name): the ONE source bench.py keys on."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from soccerplayershapepose_amd import _abi
    lib = _abi.load()
    out = []
    for i in range(lib.spsp_profile_num_kernels()):
        for part in lib.spsp_profile_kernel_name(i).decode().split('+'):
            out.append(part)
    return out


def to_library_name(rocprof_name, names):
    """rocprofv3 prints every template argument; the library's names stop after the arguments that select the tiling
    (element type / buffering flags follow).  A rocprof kernel belongs to the library name with the same base and the
    longest matching argument prefix."""
    base, args = split_template(short(rocprof_name))
    base = re.sub(r'^conv1_pool32r?_kernel$', 'conv1_pool_kernel', base)          # the stem's forms are one table row
    if base == 'ief_poll_kernel':                                                  # ... and the IEF's
        base, args = 'ief_fused_kernel', []
    best = None
    for n in names:
        nb, na = split_template(n)
        if nb == base and args[:len(na)] == na and (best is None or len(na) > len(split_template(best)[1])):
            best = n
    return best


def per_kernel(path, names):
    agg = collections.defaultdict(list)
    for row in csv.DictReader(open(path)):
        k = to_library_name(row['Kernel_Name'], names)
        if k is not None:
            agg[k].append(float(row['Counter_Value']))
    return {k: sum(v) / len(v) for k, v in agg.items()}


if __name__ != '__main__':
    raise SystemExit
NAMES = library_names()
fetch, write = per_kernel(sys.argv[1], NAMES), per_kernel(sys.argv[2], NAMES)
if len(sys.argv) > 4:
    fetch2, write2 = per_kernel(sys.argv[3], NAMES), per_kernel(sys.argv[4], NAMES)
    for k in set(fetch2) | set(write2):
        if k not in fetch and k not in write:
            fetch[k], write[k] = fetch2.get(k, 0.0), write2.get(k, 0.0)
out = {}
for k in sorted(set(fetch) | set(write)):
    rd, wr = 2 * 1024 * fetch.get(k, 0.0), 1024 * write.get(k, 0.0)
    out[k] = {'hbm_bytes_per_launch': rd + wr, 'read_bytes': rd, 'write_bytes': wr}
json.dump(out, open('profiles/hbm_traffic.json', 'w'), indent=1)
for k, v in out.items():
    print(f'{k:48s} read {v["read_bytes"] / 1e6:8.2f} MB  write {v["write_bytes"] / 1e6:8.2f} MB')
