"""Turns two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; --output-format csv) into profiles/<name>_pmc_traffic.json.
usage: python scripts/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <kernel substring> <out.json>
Corrections as MI355X_MICROARCH.md prescribes for gfx950: counters are KB; FETCH_SIZE reports half of wide coalesced
reads (x2); WRITE_SIZE is exact for 16-byte-per-lane stores."""
import csv, json, sys
from collections import defaultdict


def mean_per_dispatch(path, counter, sub):
    per = defaultdict(float)
    name = None
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and sub in r["Kernel_Name"]:
            per[r["Dispatch_Id"]] += float(r["Counter_Value"])
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    vals = list(per.values())
    return name, sum(vals) / max(1, len(vals)), len(vals)


fetch_csv, write_csv, sub, out = sys.argv[1:5]
commit = sys.argv[5] if len(sys.argv) > 5 else None
name, fkb, n = mean_per_dispatch(fetch_csv, "FETCH_SIZE", sub)
_, wkb, _ = mean_per_dispatch(write_csv, "WRITE_SIZE", sub)
res = {name: {"fetch_kb_raw": fkb, "write_kb": wkb, "fetch_bytes_corrected": fkb * 1024 * 2, "write_bytes": wkb * 1024, "launches": n},
       "_total_bytes_per_call": fkb * 1024 * 2 + wkb * 1024,
       "_kernel": name, "_commit": commit,
       "_note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes of `python3 scripts/lossgrad_once.py`; KB per dispatch averaged; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 "
                "of wide coalesced reads); WRITE_SIZE exact for 16-B/lane stores."}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
