#!/usr/bin/env python3
"""Folds tools/profile_gpu.sh summaries into profiles/traffic_per_launch.json (runs on the CPU, after the profile directories were copied
under profiles/).

    python tools/update_traffic.py s-human:125000000=profiles/r03/r03a_s-human reads:s-human:120000000=profiles/r03/r03a_reads_s-human ...

Every record gets the hash of the kernel sources it was measured on (bench.kernel_source_hash): bench.py prints a traffic figure only while
that hash still matches.  HBM bytes per launch = fabric read requests x 128 bytes (every request of these kernels is one 128-byte L2
line: TCC_EA0_RDREQ_128B == TCC_EA0_RDREQ within 1 %; the summary is checked for that) + WRITE_SIZE (KB) x 1024.  The record that a key
held before moves under "previous"."""
import json
import os
import re
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import bench  # noqa: E402

WANT = ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_128B_sum", "WRITE_SIZE", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD",
        "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_VALU", "TCC_EA0_RDREQ_LEVEL_sum")


def parse(path):
    text = open(os.path.join(path, "summary.txt")).read()
    vals, kernel = {}, None
    for line in text.splitlines():
        m = re.match(r"== pmc_\w+ (?:void )?(.+)$", line)
        if m and kernel is None:
            kernel = m.group(1)
        p = line.split()
        if len(p) == 3 and p[0] in WANT and p[0] not in vals:
            vals[p[0]] = float(p[2].split("=")[1])
    m = re.search(r"(k2u\w+)\S* .*calls=(\d+) avg_ns=(\d+)", text)
    return vals, kernel, (int(m.group(3)) if m else None), text.splitlines()[0]


def main():
    tf = os.path.join(ROOT, "profiles", "traffic_per_launch.json")
    table = json.load(open(tf))
    h = bench.kernel_source_hash()
    for arg in sys.argv[1:]:
        key, path = arg.split("=", 1)
        v, kernel, avg_ns, command = parse(os.path.join(ROOT, path))
        rd, rd128 = v["TCC_EA0_RDREQ_sum"], v["TCC_EA0_RDREQ_128B_sum"]
        assert abs(rd - rd128) <= 0.02 * rd, (key, rd, rd128)
        rec = {"hbm_bytes_per_launch": int(rd * 128 + v["WRITE_SIZE"] * 1024), "read_requests": int(rd), "bytes_per_read_request": 128,
               "write_kb": v["WRITE_SIZE"], "valu_wave_insts_per_launch": int(v["SQ_INSTS_VALU"]), "kernel": kernel,
               "kernel_avg_ns_under_kernel_trace": avg_ns, "kernel_source_hash": h, "source": f"{path}/summary.txt ({command.strip()})",
               "counters": {k: v[k] for k in WANT if k in v}}
        old = table.get(key)
        if old:
            prev = old.pop("previous", {})
            prev[old.get("source", "earlier").split("/summary")[0]] = {k: old[k] for k in ("hbm_bytes_per_launch", "read_requests", "source") if k in old}
            rec["previous"] = prev
        table[key] = rec
        print(key, rec["hbm_bytes_per_launch"], rec["valu_wave_insts_per_launch"], kernel, avg_ns)
    json.dump(table, open(tf, "w"), indent=1)


if __name__ == "__main__":
    main()
