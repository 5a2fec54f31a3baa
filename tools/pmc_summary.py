"""Per-kernel means of a rocprofv3 --pmc counter_collection.csv (kernel names mapped onto the library's table)."""
import collections
import csv
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location('ct', os.path.join(ROOT, 'profiles', 'collect_traffic.py'))
ct = importlib.util.module_from_spec(spec)
try:
    spec.loader.exec_module(ct)
except SystemExit:
    pass
names = ct.library_names()
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sys.argv[1:]:
    for row in csv.DictReader(open(path)):
        k = ct.to_library_name(row['Kernel_Name'], names) or ct.short(row['Kernel_Name'])
        agg[k][row['Counter_Name']].append(float(row['Counter_Value']))
for k, cs in sorted(agg.items()):
    print(k)
    for c, v in sorted(cs.items()):
        print(f'    {c:32s} {sum(v) / len(v):16.1f}   (n={len(v)})')
