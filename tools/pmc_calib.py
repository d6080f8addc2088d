#!/usr/bin/env python3
"""Calibration of the rocprofv3 HBM counters for the access shapes of the training kernels (round-3 review, weak #4:
FETCH x2 was applied to sc1 / nt row loads and WRITE_SIZE taken raw although MI355X_MICROARCH.md marks both as
uncalibrated outside plain streaming reads).

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d D1 -o cal -- tools/row_probe calib > calib.txt
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d D2 -o cal -- tools/row_probe calib
    tools/pmc_calib.py calib.txt D1/cal_counter_collection.csv D2/cal_counter_collection.csv out.json

Every configuration of `row_probe calib` is ONE dispatch with a known number of row bytes read and written (random
3200-byte rows of a 1.28 GB table, 16 bytes per lane).  factor = known bytes / (counter value in KB x 1024); the
factors of the policy a kernel uses (sc1 for coherent rows) replace the blanket "FETCH x2, WRITE raw" of round 3 in
tools/pmc_summary.py (argument `calib.json`)."""
import csv
import json
import sys


def counter_by_dispatch(path, name):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == name and "k_probe" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    return [float(r["Counter_Value"]) for r in rows]


def main():
    txt, fetch_csv, write_csv, out = sys.argv[1:5]
    known = []
    for line in open(txt):
        if line.startswith("CALIB "):
            name, rb, wb = [x.strip() for x in line[6:].split("|")]
            known.append((name, float(rb.split()[1]), float(wb.split()[1])))
    f = counter_by_dispatch(fetch_csv, "FETCH_SIZE")
    w = counter_by_dispatch(write_csv, "WRITE_SIZE")
    assert len(f) == len(known) == len(w), (len(f), len(w), len(known))
    res = {"unit": "factor = known row bytes / (counter KB x 1024)", "configs": {}}
    for (name, rb, wb), fk, wk in zip(known, f, w):
        res["configs"][name] = {"read_bytes": rb, "write_bytes": wb, "FETCH_SIZE_KB": fk, "WRITE_SIZE_KB": wk,
                                "fetch_factor": (rb / (fk * 1024)) if rb and fk else None,
                                "write_factor": (wb / (wk * 1024)) if wb and wk else None,
                                "FETCH_KB_per_written_byte": (fk * 1024 / wb) if (wb and not rb) else None}
    for pol in ("plain", "sc1", "nt"):
        c = res["configs"]
        if pol + " reads" in c and pol + " writes" in c:
            res[pol] = {"fetch_factor": c[pol + " reads"]["fetch_factor"], "write_factor": c[pol + " writes"]["write_factor"],
                        "fetch_factor_in_read+write": (c[pol + " read+write"]["fetch_factor"] if pol + " read+write" in c else None),
                        "write_factor_in_read+write": (c[pol + " read+write"]["write_factor"] if pol + " read+write" in c else None)}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: res[k] for k in ("plain", "sc1", "nt") if k in res}))


if __name__ == "__main__":
    main()
