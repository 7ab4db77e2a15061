#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; one counter per pass as MI355X_MICROARCH.md
prescribes: they do not fit one TCC pass) into profiles/hbm_traffic_<workload>.json.

usage: parse_pmc.py <fetch_dir> <write_dir> <workload> <out.json> [kernel_substring kernel_family n_rows source_tag]
Units/corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950
FETCH_SIZE reports half the bytes of a wide (16 B/lane) coalesced read stream, so the read side is doubled;
WRITE_SIZE is exact for 16-B stores (our stores are 4-B per lane: uncalibrated, reported as is)."""
import csv
import glob
import json
import os
import sys


def per_launch(d, counter, kernel="rollout_kernel"):
    vals = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if kernel in row.get("Kernel_Name", "") and row.get("Counter_Name") == counter:
                    vals.append(float(row["Counter_Value"]))
    return vals


def main():
    import datetime
    fd, wd, wl, out = sys.argv[1:5]
    ksub = sys.argv[5] if len(sys.argv) > 5 else "rollout_kernel"
    family = sys.argv[6] if len(sys.argv) > 6 else "rollout_kernel<shape-specialised>"     # what nocf_last_rollout_kernel() reports
    nrows = int(sys.argv[7]) if len(sys.argv) > 7 else 1024
    source = sys.argv[8] if len(sys.argv) > 8 else "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes)"
    fetch = per_launch(fd, "FETCH_SIZE", ksub)
    write = per_launch(wd, "WRITE_SIZE", ksub)
    assert fetch and write, "no rows of that kernel found"
    f_kib = sum(fetch) / len(fetch)
    w_kib = sum(write) / len(write)
    rec = {"workload": wl, "kernel": ksub, "kernel_family": family, "n": nrows, "source": source,
           "collected": datetime.date.today().isoformat(), "launches": [len(fetch), len(write)],
           "FETCH_SIZE_KiB_per_launch_raw": f_kib, "WRITE_SIZE_KiB_per_launch_raw": w_kib,
           "read_bytes_per_launch_corrected": 2.0 * f_kib * 1024.0, "write_bytes_per_launch": w_kib * 1024.0,
           "hbm_bytes_per_launch": 2.0 * f_kib * 1024.0 + w_kib * 1024.0,
           "note": "FETCH_SIZE doubled (gfx950 reports 64 B per 128-B request on wide coalesced reads); counters "
                   "sit on the L2's memory side, Infinity-Cache hits included; separate --pmc passes"}
    json.dump(rec, open(out, "w"), indent=1)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
