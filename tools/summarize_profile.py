"""Summarise a rocprofv3 run (kernel-trace --stats csv + separate --pmc FETCH_SIZE / WRITE_SIZE passes) into a
small markdown + csv pair under profiles/.   python tools/summarize_profile.py gpurun_out/prof_r01 profiles/r01_vae_render"""
import collections
import csv
import os
import sys


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:70]


def main(src, dst, require_traffic=False):
    stats = list(csv.DictReader(open(os.path.join(src, "trace", "vae_kernel_stats.csv"))))
    traffic = collections.defaultdict(lambda: {"FETCH_SIZE": [0.0, 0], "WRITE_SIZE": [0.0, 0]})
    for sub, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        p = os.path.join(src, sub, "vae_counter_collection.csv")
        if not os.path.exists(p):
            continue
        for r in csv.DictReader(open(p)):
            if r["Counter_Name"] == ctr:
                t = traffic[short(r["Kernel_Name"])][ctr]
                t[0] += float(r["Counter_Value"]); t[1] += 1
    n_rows = {c: sum(1 for t in traffic.values() if t[c][1]) for c in ("FETCH_SIZE", "WRITE_SIZE")}
    if require_traffic and not (n_rows["FETCH_SIZE"] and n_rows["WRITE_SIZE"]):
        # round 4 lost its traffic columns to two empty PMC passes without anybody noticing: an incomplete summary never
        # replaces a complete one
        sys.stderr.write("summarize_profile: %s: PMC passes gave %r kernels with counters - NOT writing %s_kernel_stats.csv "
                         "(the previous file stays)\n" % (src, n_rows, dst))
        return 3
    tmp = dst + "_kernel_stats.csv.tmp"
    with open(tmp, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_ms", "avg_us", "pct", "fetch_KB_per_launch_raw", "write_KB_per_launch_raw",
                    "hbm_MB_per_launch_corrected"])
        for r in stats[:40]:
            k = short(r["Name"])
            tr = traffic.get(k)
            fk = tr["FETCH_SIZE"][0] / tr["FETCH_SIZE"][1] if tr and tr["FETCH_SIZE"][1] else ""
            wk = tr["WRITE_SIZE"][0] / tr["WRITE_SIZE"][1] if tr and tr["WRITE_SIZE"][1] else ""
            # MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads -> x2;
            # both counters are in KB
            corr = (2 * fk + wk) / 1024.0 if fk != "" and wk != "" else ""
            w.writerow([k, r["Calls"], "%.3f" % (float(r["TotalDurationNs"]) / 1e6), "%.2f" % (float(r["AverageNs"]) / 1e3),
                        r["Percentage"], "%.1f" % fk if fk != "" else "", "%.1f" % wk if wk != "" else "",
                        "%.3f" % corr if corr != "" else ""])
    os.replace(tmp, dst + "_kernel_stats.csv")
    print("wrote", dst + "_kernel_stats.csv", "(kernels with FETCH / WRITE counters: %d / %d)" % (n_rows["FETCH_SIZE"], n_rows["WRITE_SIZE"]))
    return 0


if __name__ == "__main__":
    a = [x for x in sys.argv[1:] if not x.startswith("--")]
    sys.exit(main(a[0], a[1], require_traffic="--require-traffic" in sys.argv))
