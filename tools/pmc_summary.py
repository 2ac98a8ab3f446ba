#!/usr/bin/env python3
"""Aggregate rocprofv3 counter_collection csv files (one directory per --pmc pass): per kernel, mean counter value per launch."""
import csv, glob, json, os, sys
root = sys.argv[1]
acc = {}
for path in glob.glob(os.path.join(root, '**', '*counter_collection.csv'), recursive=True):
    with open(path) as fh:
        for row in csv.DictReader(fh):
            k = row['Kernel_Name'][:48]; c = row['Counter_Name']; v = float(row['Counter_Value'])
            a = acc.setdefault(k, {}).setdefault(c, {})
            d = row.get('Dispatch_Id', '0')
            a[d] = a.get(d, 0.0) + v             # a dispatch may report one row per dimension instance
out = {k: {c: sum(d.values()) / len(d) for c, d in cs.items()} for k, cs in acc.items()}
print(json.dumps(out, indent=1, sort_keys=True))
