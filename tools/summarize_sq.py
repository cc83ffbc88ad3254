"""Summarise a rocprofv3 --pmc SQ_* pass (one pass, 8 SQ slots) into profiles/<name>_sq_stalls.csv:
   python tools/summarize_sq.py gpurun_out/prof_sq/sq_counter_collection.csv profiles/r01
Columns are fractions of SQ_WAVE_CYCLES (MI355X_MICROARCH.md: WAIT_ANY = parked on s_waitcnt/barrier, WAIT_INST_ANY =
issue stall (MFMA pipe / RAW), ACTIVE = issuing); lds_conflict = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE."""
import collections
import csv
import sys


def main(src, dst):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(src)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVE_CYCLES":
            cnt[k] += 1
    rows = sorted(agg.items(), key=lambda kv: -kv[1]["SQ_WAVE_CYCLES"])[:30]
    with open(dst + "_sq_stalls.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "launches", "wave_cycles_share", "wait_any", "wait_inst_any", "wait_inst_lds", "active_inst", "mfma_busy_per_wave_cycle",
                    "lds_conflict"])
        tot = sum(v["SQ_WAVE_CYCLES"] for v in agg.values())
        for k, v in rows:
            wc = v["SQ_WAVE_CYCLES"]
            w.writerow([k, cnt[k], "%.4f" % (wc / tot), "%.3f" % (v["SQ_WAIT_ANY"] / wc), "%.3f" % (v["SQ_WAIT_INST_ANY"] / wc),
                        "%.3f" % (v["SQ_WAIT_INST_LDS"] / wc), "%.3f" % (v["SQ_ACTIVE_INST_ANY"] / wc),
                        "%.3f" % (v["SQ_VALU_MFMA_BUSY_CYCLES"] / wc), "%.3f" % (v["SQ_LDS_BANK_CONFLICT"] / max(v["SQ_LDS_IDX_ACTIVE"], 1.0))])
    print("wrote", dst + "_sq_stalls.csv")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
