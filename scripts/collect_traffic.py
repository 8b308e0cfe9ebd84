#!/usr/bin/env python3
"""HBM traffic of the bench's kernels from rocprofv3 PMC counters (GPU box).

Runs `python bench.py --steps S --warmup W --no-cpu-baseline --graph 0` under rocprofv3 twice -- FETCH_SIZE and WRITE_SIZE
cost 3 and 2 of the 4 TCC slots, so they need SEPARATE passes (MI355X_MICROARCH.md, rocprofv3 PMC slots) -- with
--kernel-trace only (no other trace domain), and writes profiles/<tag>_hbm_traffic.json:
    per kernel: launches, FETCH_SIZE and WRITE_SIZE sums (KiB, as reported), bytes per launch with the gfx950 correction
    bytes = 2 * FETCH_SIZE * 1024 (FETCH_SIZE reports half the bytes of wide coalesced reads on gfx950) + WRITE_SIZE * 1024.
bench.py reads that file and fills roofline.traffic for its dominant kernel.
"""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_pass(counter, out_dir, steps, warmup, extra=()):
    cmd = ["rocprofv3", "--kernel-trace", "--pmc", counter, *extra, "--output-format", "csv", "-d", out_dir, "--",
           sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(steps), "--warmup", str(warmup), "--no-cpu-baseline", "--graph", "0", "--train-steps", "0", "--no-parity", "--no-live-traffic"]
    env = dict(os.environ, TMPDIR="/tmp")
    subprocess.run(cmd, check=True, env=env, cwd="/tmp", stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    rows = list(csv.DictReader(open(glob.glob(os.path.join(out_dir, "*", "*_counter_collection.csv"))[0])))
    if extra:                                             # several counters in one pass: {kernel: {counter: [n, sum]}}
        multi = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
        for r in rows:
            a = multi[r["Kernel_Name"]][r["Counter_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
        return multi
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in rows:
        if r["Counter_Name"] != counter:
            continue
        a = agg[r["Kernel_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return agg


def bench_source_digest():
    sys.path.insert(0, ROOT)
    import bench
    return bench.source_digest()


def short(name):
    n = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0]


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    steps, warmup = 5, 2
    scratch = os.path.join(ROOT, "gpurun_out", "traffic")
    fetch = run_pass("FETCH_SIZE", os.path.join(scratch, "fetch"), steps, warmup)
    write = run_pass("WRITE_SIZE", os.path.join(scratch, "write"), steps, warmup)
    out = {}
    for k in sorted(set(fetch) | set(write)):
        n = max(fetch.get(k, [0, 0])[0], write.get(k, [0, 0])[0])
        f, w = fetch.get(k, [0, 0.0])[1], write.get(k, [0, 0.0])[1]
        if n == 0:
            continue
        out[short(k)] = {"launches": n, "FETCH_SIZE_KiB_sum": f, "WRITE_SIZE_KiB_sum": w,
                         "hbm_bytes_per_launch": (2.0 * f * 1024.0 + w * 1024.0) / n,
                         "read_bytes_per_launch_corrected": 2.0 * f * 1024.0 / n, "write_bytes_per_launch": w * 1024.0 / n}
    path = os.path.join(ROOT, "profiles", tag + "_hbm_traffic.json")
    json.dump({"method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE in separate passes over bench.py (eager launches); "
                         "bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (gfx950: FETCH_SIZE reports half of wide coalesced reads)",
               "source_sha256_16": bench_source_digest(), "kernels": out}, open(path, "w"), indent=1, sort_keys=True)
    for k, v in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])[:12]:
        print("%-90s launches %5d  %10.2f MB/launch" % (k[:90], v["launches"], v["hbm_bytes_per_launch"] / 1e6))
    print("wrote", path)

    # MFMA utilisation: SQ_VALU_MFMA_BUSY_CYCLES (cycles, summed over all SIMDs) against the cycles the chip was active during the
    # kernel (GRBM_GUI_ACTIVE is the sum over the 8 XCDs) x 1024 SIMDs
    mm = run_pass("SQ_VALU_MFMA_BUSY_CYCLES", os.path.join(scratch, "mfma"), steps, warmup, extra=("GRBM_GUI_ACTIVE",))
    util = {}
    for k, c in mm.items():
        busy, act = c.get("SQ_VALU_MFMA_BUSY_CYCLES", [0, 0.0]), c.get("GRBM_GUI_ACTIVE", [0, 0.0])
        if act[1] <= 0 or busy[0] == 0:
            continue
        util[short(k)] = {"launches": busy[0], "mfma_busy_cycles_per_launch": busy[1] / busy[0],
                          "active_cycles_per_launch_per_xcd": act[1] / act[0] / 8.0,
                          "mfma_busy_frac": busy[1] / (act[1] / 8.0 * 1024.0)}
    tot_busy = sum(v["mfma_busy_cycles_per_launch"] * v["launches"] for v in util.values())
    tot_act = sum(v["active_cycles_per_launch_per_xcd"] * v["launches"] for v in util.values())
    path = os.path.join(ROOT, "profiles", tag + "_mfma_util.json")
    json.dump({"method": "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE over bench.py (eager launches); "
                         "mfma_busy_frac = MFMA_BUSY / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs)",
               "whole_run_mfma_busy_frac": tot_busy / (tot_act * 1024.0) if tot_act else None, "kernels": util},
              open(path, "w"), indent=1, sort_keys=True)
    for k, v in sorted(util.items(), key=lambda kv: -kv[1]["mfma_busy_cycles_per_launch"] * kv[1]["launches"])[:8]:
        print("%-90s MFMA busy %5.1f %%" % (k[:90], 100 * v["mfma_busy_frac"]))
    print("whole run MFMA busy %.1f %%; wrote %s" % (100 * tot_busy / (tot_act * 1024.0) if tot_act else 0.0, path))


if __name__ == "__main__":
    main()
