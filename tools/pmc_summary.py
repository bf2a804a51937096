"""Sums rocprofv3 --pmc counter_collection.csv files by kernel and counter.
  python3 tools/pmc_summary.py <dir-or-csv> [kernel-substring]"""
import csv, collections, pathlib, sys
root = pathlib.Path(sys.argv[1])
want = sys.argv[2] if len(sys.argv) > 2 else ''
files = [root] if root.is_file() else sorted(root.rglob('*counter_collection.csv'))
sums = collections.defaultdict(float)
calls = collections.defaultdict(set)
for f in files:
  for row in csv.DictReader(open(f)):
    name = row['Kernel_Name']
    if want not in name:
      continue
    key = (name.split('(')[0][:70], row['Counter_Name'])
    sums[key] += float(row['Counter_Value'])
    calls[key].add((f.name, row['Dispatch_Id']))
for (name, counter), v in sorted(sums.items()):
  n = len(calls[(name, counter)])
  print('%-72s %-28s total %.6e  launches %d  per launch %.6e' % (name, counter, v, n, v / n))
