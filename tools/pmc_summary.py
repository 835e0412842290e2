#!/usr/bin/env python3
"""Summarise rocprofv3 counter-collection CSVs: per kernel name (regex filter), per counter: mean value per dispatch."""
import csv
import glob
import json
import re
import sys
from collections import defaultdict

outdir, pattern = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else 'raster')
acc = defaultdict(lambda: defaultdict(list))
for path in glob.glob(f'{outdir}/**/*counter_collection.csv', recursive=True):
    with open(path) as f:
        rd = csv.DictReader(f)
        per_dispatch = defaultdict(float)
        meta = {}
        for row in rd:
            name = row.get('Kernel_Name', '')
            if not re.search(pattern, name):
                continue
            key = (row.get('Dispatch_Id'), row['Counter_Name'])
            per_dispatch[key] += float(row['Counter_Value'])
            meta[row.get('Dispatch_Id')] = (name.split('(')[0][:60], row.get('Grid_Size'), row.get('VGPR_Count'), row.get('LDS_Block_Size'))
        for (did, cname), v in per_dispatch.items():
            acc[meta[did]][cname].append(v)
res = {}
for k, counters in acc.items():
    res[' | '.join(str(x) for x in k)] = {c: dict(mean=sum(v) / len(v), n=len(v)) for c, v in sorted(counters.items())}
print(json.dumps(res, indent=1))
