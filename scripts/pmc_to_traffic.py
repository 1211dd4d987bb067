"""profiles/<tag>_pmc_per_kernel.txt (scripts/pmc_summary.py output) -> profiles/gemm_traffic.json, the `traffic` field
of the bench line: mean HBM-side bytes per GEMM launch (all encoder linears), PMC FETCH_SIZE x 2 + WRITE_SIZE.

    python scripts/pmc_to_traffic.py profiles/r02n_pmc_per_kernel.txt
"""
import datetime
import json
import re
import sys

src = sys.argv[1]
tot_n = tot_b = 0
busy_t = busy_w = clk_w = 0.0
per = {}
for line in open(src):
    m = re.match(r"(gemm_\w+<.*?)\s+n=(\d+)\s+avg\s+([\d.]+) us\s+clk\s+([\d.]+) GHz\s+mfma_util\s+([\d.]+)%.*hbm_rd\s+([\d.]+) MB wr\s+([\d.]+) MB", line)
    if not m:
        continue
    name, n, rd, wr = m.group(1), int(m.group(2)), float(m.group(6)) * 1e6, float(m.group(7)) * 1e6
    avg_us, clk, util = float(m.group(3)), float(m.group(4)), float(m.group(5))
    if name.startswith("gemm_smf16_kernel<64,"):
        continue            # the 24 position-table projections of model load (64-row tiles): not part of a step
    per[name] = {"launches": n, "read_bytes": rd, "write_bytes": wr, "avg_us": avg_us, "mfma_busy_pct": util, "clock_ghz": clk}
    busy_t += n * avg_us * util
    clk_w += n * avg_us * clk
    busy_w += n * avg_us
    tot_n += n
    tot_b += n * (rd + wr)
out = {
    "kernel": "all GEMM launches of one encoder pass at B = 256 (gemm_smf16_kernel, every tile height / epilogue in use)",
    "hbm_bytes_per_launch": round(tot_b / tot_n),
    # SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 4 SIMDs x CUs), weighted by kernel time: the fraction of the MFMA
    # pipes' cycles AT THE CLOCK THE CHIP RAN AT (power-capped) that issued an MFMA
    "mfma_busy_pct": round(busy_t / busy_w, 1),
    "clock_ghz": round(clk_w / busy_w, 2),
    "per_kernel": per,
    "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over `bench.py --steps 1 --warmup 1 --no-pipeline`; "
              "FETCH_SIZE doubled (gfx950 reports half of wide coalesced reads, guides/MI355X_MICROARCH.md §HBM); launch-weighted "
              "mean.  These are L2-miss bytes at the TCC/EA boundary: re-reads served by the 256 MiB infinity cache count too.",
    "source": src,
    "collected": datetime.date.today().isoformat() + " (" + src + ")",
}
json.dump(out, open("profiles/gemm_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
