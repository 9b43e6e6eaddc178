#!/usr/bin/env python3
"""profiles/r06_q2_apply_traffic.json from the PMC summaries tools/r06_profile.sh left under gpurun_out/r06prof_<tag>/:
HBM bytes per q2_apply_nb_kernel launch = FETCH_SIZE x 2 (gfx950: 128-byte requests tallied at 64 bytes,
MI355X_MICROARCH.md) + WRITE_SIZE (16-byte stores: exact), both in KiB in the counter files; with the sha256 of the
kernel's source file and the git commit, so that bench.py can tell a stale record.  usage: make_traffic_record.py <tag> [n]"""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def counter(path, kernel, name):
    for line in open(path):
        f = line.split()
        if len(f) >= 5 and f[0].startswith(kernel) and name in f:
            i = f.index(name)
            return float(f[i + 1]), float(f[i + 2])
        if False:
            return float(f[2]), float(f[3])
    raise SystemExit("no %s for %s in %s" % (name, kernel, path))


NOTE = ("four blocks of sweeps per pass: WRITE_SIZE is Z stored once per bundle (n^3/32 bytes = 137 GB; 270 GB with pairs, 180 "
        "with three).  FETCH_SIZE is not the reads of Z (137 GB as well: the 8-byte sc1 loads do not appear in it -- "
        "MI355X_MICROARCH.md: access widths other than 16 B per lane are uncalibrated) but the records' way into L2: 12 x 16 B per "
        "thread and group = 3.2 GB of records x 256 slabs = 0.82 TB requested, mostly Infinity-Cache hits, which the counter "
        "includes.  hbm_bytes_per_launch is the guide's formula on these counters: an upper bound of the HBM traffic, not a "
        "byte count; algorithmic_bytes_per_launch counts Z once each way per bundle and every record once.")


def main():
    tag = sys.argv[1]
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
    d = os.path.join(ROOT, "gpurun_out", "r06prof_" + tag)
    nf, fetch = counter(os.path.join(d, "pmc_fetch_c3.txt"), "q2_apply", "FETCH_SIZE")
    nw, write = counter(os.path.join(d, "pmc_write_c3.txt"), "q2_apply", "WRITE_SIZE")
    sha_box = open(os.path.join(d, "source_sha256.txt")).read().split()[0]
    src = os.path.join(ROOT, "eigenkernel_amd", "csrc", "ek_sb2st.hip")
    sha = hashlib.sha256(open(src, "rb").read()).hexdigest()
    if sha != sha_box:
        raise SystemExit("ek_sb2st.hip has changed since the measurement (%s vs %s)" % (sha[:12], sha_box[:12]))
    git = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
    rec = {"n": n, "ncols": n, "kernel": "q2_apply_nb_kernel<4>", "git": git, "source_sha256": sha,
           "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-include-regex) on "
                     "`python3 bench.py --steps 1 --warmup 0 ...` (tools/r06_profile.sh); counters are KiB; FETCH_SIZE doubled as "
                     "MI355X_MICROARCH.md prescribes for gfx950, WRITE_SIZE taken as is (16-byte stores)",
           "fetch_size_kib": fetch / nf, "write_size_kib": write / nw,
           "hbm_bytes_per_launch": (2.0 * fetch / nf + write / nw) * 1024.0,
           # four blocks of 32 sweeps per pass: Z (on average its lower half) is read and written once per 128 sweeps,
           # and every group record (2 x 96 x 32 doubles, n^2 / 4096 of them) comes from memory once
           "algorithmic_bytes_per_launch": 2.0 * (n / 128.0) * (8.0 * n * n / 2.0) + (n * n / 4096.0) * 6144 * 8.0,
           "note": NOTE}
    out = os.path.join(ROOT, "profiles", "r06_q2_apply_traffic.json")
    json.dump(rec, open(out, "w"), indent=1)
    print(out, rec["hbm_bytes_per_launch"] / 1e12, "TB per launch")


if __name__ == "__main__":
    main()
