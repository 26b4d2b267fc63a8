#!/usr/bin/env python3
"""HBM traffic per launch of the streaming kernels from the rocprofv3 PMC passes of
profiles/run_pmc.sh: make_traffic.py PMC_DIR OUT_JSON

FETCH_SIZE and WRITE_SIZE are reported in KiB-like units of 1024 B; on gfx950 FETCH_SIZE
tallies the 128-byte requests of wide coalesced reads at 64 B, so it is doubled (the select
kernel streams 16 B per lane; the match kernel reads almost nothing).  WRITE_SIZE is taken
as is.  Kernels are keyed by their full template instance: the (conf, locs) and the y_true
variants of the match kernel write different amounts and must not be averaged together."""
import collections
import csv
import glob
import json
import os
import re
import sys

root, out = sys.argv[1], sys.argv[2]
INSTANCES = {
    "match": r"match_geom_kernel<false, false, false>",
    "match_ytrue": r"match_geom_kernel<true, false, false>",
    "select": r"select_rows6_kernel<false>",
}
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] not in ("FETCH_SIZE", "WRITE_SIZE"):
            continue
        for short, pat in INSTANCES.items():
            if re.search(pat, row["Kernel_Name"]):
                acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
res = {}
for short in INSTANCES:
    c = acc.get(short)
    if not c or "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
        continue
    fetch = sum(c["FETCH_SIZE"]) / len(c["FETCH_SIZE"]) * 1024.0 * 2.0
    write = sum(c["WRITE_SIZE"]) / len(c["WRITE_SIZE"]) * 1024.0
    res[short] = int(fetch + write)
    res[short + "_detail"] = {"fetch_bytes_x2": int(fetch), "write_bytes": int(write),
                              "launches": [len(c["FETCH_SIZE"]), len(c["WRITE_SIZE"])]}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res))
