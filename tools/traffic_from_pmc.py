#!/usr/bin/env python
"""HBM traffic per kernel from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, collected in
separate runs as MI355X_MICROARCH.md prescribes).  gfx950 correction: FETCH_SIZE counts 64 B per
128-B request for wide coalesced streaming reads -> doubled; WRITE_SIZE is exact; both are in KiB.

    python tools/traffic_from_pmc.py fetch_counter_collection.csv write_counter_collection.csv out.json
"""
import collections
import csv
import json
import sys


# op kind of each kernel (bench.py KIND_KERNELS): traffic is reported per kind, like bench.py's roofline
KIND_OF = {"pwb_kernel": "pw", "pws_kernel": "pw", "pwb_shared_kernel": "pw", "pw_kernel": "pw", "mbxb_kernel": "mbx", "mbxd_kernel": "mbx", "mbxp_kernel": "mbx", "mbx_kernel": "mbx",
           "sep_kernel": "sep", "sepf_kernel": "sep", "dw_kernel": "dw", "se_kernel": "se", "fuse_kernel": "fuse", "stem_kernel": "stem", "stem16_kernel": "stem", "stem_u8_kernel": "stem",
           "aggregate_kernel": "aggregate", "aggregate_reg_kernel": "aggregate", "preprocess_kernel": "preprocess",
           "nms_eval_kernel": "nms", "nms_commit_kernel": "nms", "nms_bound_kernel": "nms", "nms_init_kernel": "nms",
           "nms_coop_kernel": "nms", "nms_reg_kernel": "nms", "nms_solo_kernel": "nms", "prefix_select_kernel": "nms",
           "prefix_check_kernel": "nms"}


def per_kernel(path, counter):
    tot, n = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("uda::", "")
        k = KIND_OF.get(k.split("<")[0], k.split("<")[0])
        tot[k] += float(r["Counter_Value"])
        n[k] += 1
    return tot, n


def main():
    fetch, nf = per_kernel(sys.argv[1], "FETCH_SIZE")
    write, nw = per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in sorted(set(fetch) | set(write)):
        launches = max(nf.get(k, 0), nw.get(k, 0))
        if not launches:
            continue
        rd = 2.0 * fetch.get(k, 0.0) * 1024 / max(nf.get(k, 1), 1)
        wr = write.get(k, 0.0) * 1024 / max(nw.get(k, 1), 1)
        out[k] = {"launches": launches, "read_bytes_per_launch": int(rd), "write_bytes_per_launch": int(wr),
                  "hbm_bytes_per_launch": int(rd + wr),
                  "note": "FETCH_SIZE doubled (gfx950 counts 64 B per 128-B request); KiB -> bytes"}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    for k, v in out.items():
        print("%-24s launches %5d  read %10.1f MB  write %10.1f MB per launch" % (
            k, v["launches"], v["read_bytes_per_launch"] / 1e6, v["write_bytes_per_launch"] / 1e6))


if __name__ == "__main__":
    main()
