"""Turns rocprofv3 outputs into the per-kernel summary that bench.py's `roofline.traffic` reads.

  rocprofv3 --kernel-trace --stats ...              -> <dir>/<p>_kernel_trace.csv
  rocprofv3 --pmc FETCH_SIZE  (its own pass)        -> <dir_f>/<p>_counter_collection.csv
  rocprofv3 --pmc WRITE_SIZE  (its own pass)        -> <dir_w>/<p>_counter_collection.csv

FETCH_SIZE / WRITE_SIZE are in KiB.  Per MI355X_MICROARCH.md (HBM section) FETCH_SIZE reports half the bytes of wide
coalesced streaming reads on gfx950 (TCC_EA0_RDREQ x 64 B for 128-B requests): the corrected read figure doubles
it; gather-heavy kernels are uncalibrated, so both raw and corrected values are kept.  Only ACTIVE launches count
(a launch whose windows have all terminated exits at once): active = duration above 20% of the kernel's maximum."""
import collections
import csv
import json
import sys


def per_kernel(path, value_key):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].split("(")[0]
        if value_key == "dur":
            d[name].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        else:
            d[name].append(float(r["Counter_Value"]))
    return d


def csrc_sha():
    import hashlib, os
    h = hashlib.sha256()
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mc_slam_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".h", ".hip")):
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def main(trace_csv, fetch_csv, write_csv, out_json, batch, workload="c3"):
    dur = per_kernel(trace_csv, "dur")
    fe = per_kernel(fetch_csv, "ctr")
    wr = per_kernel(write_csv, "ctr")
    out = {"batch": int(batch), "workload": workload, "csrc_sha": csrc_sha(), "unit": "bytes per launch (active launches)", "kernels": {}}
    for k, v in dur.items():
        if not k.startswith("k_"):
            continue
        mx = max(v)
        act = [x for x in v if x > 0.2 * mx]
        f = [x for x in fe.get(k, []) if x > 0.2 * max(fe.get(k, [1]))]
        w_ = [x for x in wr.get(k, []) if x > 0.2 * max(wr.get(k, [1]))]
        med = lambda a: sorted(a)[len(a) // 2] if a else 0.0
        out["kernels"][k] = {"active_launches": len(act), "launches": len(v), "avg_active_us": sum(act) / len(act) / 1e3,
                             "fetch_raw": med(f) * 1024, "fetch_corrected": 2 * med(f) * 1024, "write": med(w_) * 1024}
    json.dump(out, open(out_json, "w"), indent=1)
    for k, r in sorted(out["kernels"].items(), key=lambda kv: -kv[1]["avg_active_us"] * kv[1]["active_launches"]):
        print("%-16s act %4d/%4d  avg %8.1f us  fetch %8.1f MB (x2: %8.1f)  write %8.1f MB" % (
            k, r["active_launches"], r["launches"], r["avg_active_us"], r["fetch_raw"] / 1e6, r["fetch_corrected"] / 1e6, r["write"] / 1e6))


if __name__ == "__main__":
    main(*sys.argv[1:7])
