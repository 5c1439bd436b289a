"""rocprofv3 --pmc counter_collection csv -> profiles/r01_pmc_sa_mlp_fused.json (per-launch averages of the fused kernel).
usage: python profiles/pmc_mfma_summarize.py <counter_collection.csv> <algorithmic_flops> > out.json"""
import collections
import csv
import json
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "sa_mlp_fused_kernel<128>" in r["Kernel_Name"]]
alg = float(sys.argv[2])
per = collections.defaultdict(lambda: collections.defaultdict(float))
dur = {}
for r in rows:
    per[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
    dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
ids = sorted(per, key=int)[2:]                       # skip warm-up launches
n = len(ids)
avg = {k: sum(per[i][k] for i in ids) / n for k in per[ids[0]]}
d = sum(dur[i] for i in ids) / n
gui = avg["GRBM_GUI_ACTIVE"] / 8.0                   # reported summed over the 8 XCDs
flops = avg["SQ_INSTS_VALU_MFMA_MOPS_F32"] * 512.0
clock = gui / (d * 1e-6) / 1e9
out = {
    "command": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace "
               "--output-format csv -- python profiles/mfma_probe.py  (bench.roofline_sa_mlp_fused: 800 clouds x 512 pts, 128 centres x "
               "64 samples, 128-128-128)",
    "formulas": "GRBM_GUI_ACTIVE is summed over the 8 XCDs -> /8; MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 1024 SIMDs); "
                "MFMA flops = SQ_INSTS_VALU_MFMA_MOPS_F32 * 512 and must equal the algorithmic flops (every tile served once); "
                "peak at the effective clock = 256 CUs * 4 SIMDs * 64 flop/clk * clock",
    "launches_averaged": n,
    "kernel": "prcnn::sa_mlp_fused_kernel<128>",
    "dur_us_under_pmc": d,
    "counters": avg,
    "MfmaUtil_percent": 100.0 * avg["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui * 1024.0),
    "mfma_flops_counted": flops,
    "algorithmic_flops": alg,
    "counted_over_algorithmic": flops / alg,
    "effective_clock_GHz": clock,
    "achieved_TFLOPs_under_pmc": alg / (d * 1e-6) / 1e12,
    "f32_mfma_peak_at_effective_clock_TFLOPs": 256 * 4 * 64 * clock * 1e9 / 1e12,
}
out["frac_of_peak_at_effective_clock"] = out["achieved_TFLOPs_under_pmc"] / out["f32_mfma_peak_at_effective_clock_TFLOPs"]
print(json.dumps(out, indent=1))
