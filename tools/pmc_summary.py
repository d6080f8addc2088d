#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE / TCC_HIT_sum,TCC_MISS_sum collected in SEPARATE
runs, as /opt/skills/guides/MI355X_MICROARCH.md prescribes) into profiles/<round>_pmc.json.

HBM bytes per launch of the training kernel:
    read  = FETCH_SIZE [KB] * 1024 * 2     (gfx950: FETCH_SIZE reports exactly 1/2 of a wide coalesced
                                            16-B/lane stream -- the guide's correction; our row loads are
                                            buffer_load_dwordx4 of 1 KiB per wavefront)
    write = WRITE_SIZE [KB] * 1024         (uncalibrated in the guide; compared against the algorithmic
                                            write bytes below as a plausibility check)
usage: pmc_summary.py <fetch.csv> <write.csv> <l2.csv> <out.json> [kernel substring] [bench.json of the profiled command]
The bench line of the profiled command supplies the shape (vocab, dim, negative, bitlevel) and the centre words per
launch, so that bench.py can scale the measured bytes to its own launch size (roofline.traffic).
"""
import csv
import json
import sys


def per_dispatch(path, kernel_sub):
    agg = {}
    for r in csv.DictReader(open(path)):
        if kernel_sub in r.get("Kernel_Name", ""):
            c = r["Counter_Name"]
            a = agg.setdefault(c, [0, 0.0])
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    return {c: (v[1] / v[0], v[0]) for c, v in agg.items()}


def main():
    fetch, write, l2, out = sys.argv[1:5]
    ksub = sys.argv[5] if len(sys.argv) > 5 else "k_train_tuples"
    bench = sys.argv[6] if len(sys.argv) > 6 else None
    f = per_dispatch(fetch, ksub)
    w = per_dispatch(write, ksub)
    l = per_dispatch(l2, ksub) if l2 != "-" else {}
    fetch_kb, nf = f["FETCH_SIZE"]
    write_kb, nw = w["WRITE_SIZE"]
    # calibration (round 4): tools/row_probe calib under the same two counters, known row bytes, for the cache policy the
    # kernel's row accesses carry (sc1) -- profiles/r04_pmc_calibration.json; round 3 applied the guide's x2 / x1 unchecked
    ff, wf, src = 2.0, 1.0, "x2 / x1 (MI355X_MICROARCH.md, uncalibrated for this access shape)"
    import os
    cal = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r04_pmc_calibration.json")
    if os.path.exists(cal):
        c = json.load(open(cal)).get("sc1", {})
        if c.get("fetch_factor_in_read+write") and c.get("write_factor_in_read+write"):
            ff, wf = c["fetch_factor_in_read+write"], c["write_factor_in_read+write"]
            src = ("profiles/r04_pmc_calibration.json: row_probe calib, sc1 16-byte-lane random-row read+write with known bytes: "
                   "FETCH_SIZE x %.4f, WRITE_SIZE x %.4f" % (ff, wf))
    res = {
        "kernel": ksub,
        "dispatches_sampled": {"FETCH_SIZE": nf, "WRITE_SIZE": nw},
        "FETCH_SIZE_KB_per_launch": fetch_kb,
        "WRITE_SIZE_KB_per_launch": write_kb,
        "hbm_read_bytes_per_launch": fetch_kb * 1024 * ff,
        "hbm_write_bytes_per_launch": write_kb * 1024 * wf,
        "hbm_bytes_per_launch": fetch_kb * 1024 * ff + write_kb * 1024 * wf,
        "counter_factors": src,
    }
    if l:
        hit, miss = l["TCC_HIT_sum"][0], l["TCC_MISS_sum"][0]
        res["l2_hit_rate"] = hit / (hit + miss)
    if bench:
        for line in open(bench):
            if line.startswith("{"):
                b = json.loads(line)
                c = b["config"]
                res.update({"words_per_launch": c["words_per_step_per_gpu"], "vocab": c["vocab"], "dim": c["dim"],
                            "negative": c["negative"], "bitlevel": c["bitlevel"], "ids": c["ids"],
                            "algorithmic_bytes_per_launch": b["roofline"]["algorithmic_bytes_per_launch"],
                            "avg_launch_ms_unprofiled": b["roofline"]["avg_launch_ms"]})
                res["hbm_bytes_over_algorithmic"] = res["hbm_bytes_per_launch"] / res["algorithmic_bytes_per_launch"]
                res["counter_GBps_at_unprofiled_launch_time"] = res["hbm_bytes_per_launch"] / (b["roofline"]["avg_launch_ms"] * 1e-3) / 1e9
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
