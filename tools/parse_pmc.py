#!/usr/bin/env python3
"""Turn rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter_collection.csv files into a per-launch HBM-traffic
figure for one kernel, applying the gfx950 correction of /opt/skills/guides/MI355X_MICROARCH.md §HBM:
FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request for wide
coalesced reads, i.e. exactly half the bytes -> doubled.  WRITE_SIZE is exact for 16-B/lane stores.

    python tools/parse_pmc.py <fetch_counter_collection.csv> <write_counter_collection.csv> <kernel-substring> <out.json>
"""
import csv
import json
import sys


def mean_counter(path, kernel, counter):
    vals = []
    for r in csv.DictReader(open(path)):
        if kernel in r["Kernel_Name"] and r["Counter_Name"] == counter:
            vals.append(float(r["Counter_Value"]))
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)


def main():
    fetch_csv, write_csv, kernel, out = sys.argv[1:5]
    f, nf = mean_counter(fetch_csv, kernel, "FETCH_SIZE")
    w, nw = mean_counter(write_csv, kernel, "WRITE_SIZE")
    res = dict(kernel=kernel, dispatches_fetch=nf, dispatches_write=nw, FETCH_SIZE_KiB_mean=f, WRITE_SIZE_KiB_mean=w,
               fetch_bytes_corrected=None if f is None else 2 * f * 1024, write_bytes=None if w is None else w * 1024,
               correction="FETCH_SIZE x2 on gfx950 (wide coalesced reads tallied at 64 B per 128-B request); KiB -> bytes")
    if f is not None and w is not None:
        res["traffic_bytes_per_launch"] = res["fetch_bytes_corrected"] + res["write_bytes"]
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
