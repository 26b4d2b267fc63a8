#!/usr/bin/env python3
"""HBM traffic per launch of the two streaming kernels from the rocprofv3 PMC passes of
profiles/run_pmc.sh: make_traffic.py PMC_DIR OUT_JSON

FETCH_SIZE and WRITE_SIZE are reported in KiB-like units of 1024 B; on gfx950 FETCH_SIZE
tallies the 128-byte requests of wide coalesced reads at 64 B, so it is doubled (the select
kernel streams 16 B per lane; the match kernel reads almost nothing).  WRITE_SIZE is taken
as is."""
import collections
import csv
import glob
import json
import os
import re
import sys

root, out = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        m = re.search(r"(\w+)_kernel", row["Kernel_Name"])
        if m and row["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
            acc[m.group(1)][row["Counter_Name"]].append(float(row["Counter_Value"]))
res = {}
for short, name in (("match", "match_geom"), ("select", "select_rows6")):
    c = acc.get(name)
    if not c or "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
        continue
    fetch = sum(c["FETCH_SIZE"]) / len(c["FETCH_SIZE"]) * 1024.0 * 2.0
    write = sum(c["WRITE_SIZE"]) / len(c["WRITE_SIZE"]) * 1024.0
    res[short] = int(fetch + write)
    res[short + "_detail"] = {"fetch_bytes_x2": int(fetch), "write_bytes": int(write)}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res))
