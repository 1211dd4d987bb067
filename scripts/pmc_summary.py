"""Aggregate rocprofv3 --pmc counter_collection CSVs per kernel.

    python scripts/pmc_summary.py gpurun_out/pmc_<tag>_{1,2,3,4}.csv > profiles/<name>.txt
Per kernel: dispatch count, mean duration, mean of every counter per dispatch, and derived
MFMA utilisation / HBM traffic (FETCH_SIZE doubled per guides/MI355X_MICROARCH.md §HBM:
rocprofv3 on gfx950 reports half the bytes of wide coalesced reads; WRITE_SIZE taken as is).
"""
import collections
import csv
import re
import sys


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.split(r"\(", n)[0][:52]


def main():
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(dict)
    for path in sys.argv[1:]:
        for r in csv.DictReader(open(path)):
            k = short(r["Kernel_Name"])
            per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur[k][(path, r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print("# per-kernel PMC means per dispatch (profiled run: clocks are lower than in the timed run)")
    for k in sorted(per, key=lambda k: -sum(dur[k].values())):
        c = {n: sum(v) / len(v) for n, v in per[k].items()}
        nd = max(len(v) for v in per[k].values())
        us = sum(dur[k].values()) / len(dur[k])
        line = f"{k:<52} n={nd:<5} avg {us:9.1f} us"
        if "GRBM_GUI_ACTIVE" in c and "SQ_VALU_MFMA_BUSY_CYCLES" in c and c["GRBM_GUI_ACTIVE"] > 0:
            cyc = c["GRBM_GUI_ACTIVE"] / 8.0          # summed over 8 XCDs
            # cycles / duration is a clock only when the dispatch is long against the counter's start / stop window: below
            # ~50 us the quotient came out at 3.8 - 14.6 GHz in round 4 (a method artefact) and is not printed
            clk = f"{cyc / us / 1e3:4.2f}" if us >= 50.0 else " n/a"
            line += f"  clk {clk} GHz  mfma_util {c['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024):5.1%}"
        if "SQ_WAVE_CYCLES" in c and c["SQ_WAVE_CYCLES"] > 0:
            w = c["SQ_WAVE_CYCLES"]
            line += (f"  wait_any {c.get('SQ_WAIT_ANY', 0) / w:4.0%} wait_inst {c.get('SQ_WAIT_INST_ANY', 0) / w:4.0%}"
                     f" active {c.get('SQ_ACTIVE_INST_ANY', 0) / w:4.0%}")
        if "SQ_LDS_IDX_ACTIVE" in c and c["SQ_LDS_IDX_ACTIVE"] > 0:
            line += f"  lds_conflict {c.get('SQ_LDS_BANK_CONFLICT', 0) / c['SQ_LDS_IDX_ACTIVE']:4.0%}"
        if "FETCH_SIZE" in c:
            rd = 2.0 * c["FETCH_SIZE"] * 1024 / 1e6
            wr = c.get("WRITE_SIZE", 0) * 1024 / 1e6
            line += f"  hbm_rd {rd:8.1f} MB wr {wr:8.1f} MB ({(rd + wr) / us if us else 0:4.2f} TB/s)"
        if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c:
            line += f"  l2_hit {c['TCC_HIT_sum'] / (c['TCC_HIT_sum'] + c['TCC_MISS_sum']):4.0%}"
        print(line)


if __name__ == "__main__":
    main()
