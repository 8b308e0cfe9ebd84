#!/usr/bin/env python3
"""Per-launch PMC counters of one kernel (GPU box): runs a driver script under rocprofv3 once per counter group (--kernel-trace --pmc only,
no other trace domain) and prints the per-launch averages of the kernels whose name contains a pattern.
    python scripts/pmc_kernel.py <name-substring> <driver.py> [driver args ...]"""
import collections, csv, glob, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GROUPS = ["SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD",
          "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS",
          "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES",
          "SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM"]
pat, driver = sys.argv[1], [os.path.abspath(sys.argv[2])] + sys.argv[3:]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for grp in GROUPS:
    out_dir = tempfile.mkdtemp(prefix="pmc_", dir="/tmp")
    cmd = ["rocprofv3", "--kernel-trace", "--pmc", *grp.split(), "--output-format", "csv", "-d", out_dir, "--", sys.executable, *driver]
    subprocess.run(cmd, check=True, env=dict(os.environ, TMPDIR="/tmp"), cwd="/tmp", stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    for r in csv.DictReader(open(glob.glob(os.path.join(out_dir, "*", "*_counter_collection.csv"))[0])):
        if pat in r["Kernel_Name"]:
            a = agg[r["Kernel_Name"]][r["Counter_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
for k, cs in agg.items():
    print(k)
    for c, (n, v) in cs.items():
        print("    %-28s per launch %16.0f   (%d launches)" % (c, v / n, n))
