"""Waves per SIMD each kernel of the product step can hold, from the launch records of a rocprofv3 kernel trace (csv): workgroup size,
LDS per workgroup, VGPRs (architectural + accumulation) per lane -> resident workgroups per CU by each limit (160 KB LDS, 512 registers
per SIMD lane, 32 waves per CU... the smallest binds), and how many CUs' worth of workgroups a launch brings.
usage: rocprofv3 --kernel-trace --output-format csv -d out -- python profiles/pmc_step_probe.py 2; python profiles/occupancy_table.py out/*/*kernel_trace.csv"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
by = collections.OrderedDict()
for r in rows:
    n = r["Kernel_Name"].replace("void ", "").replace("prcnn::", "").split("(")[0][:44]
    wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
    grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // wg
    key = (n, wg, int(r["LDS_Block_Size"]), int(r.get("VGPR_Count", 0) or 0), int(r.get("Accum_VGPR_Count", 0) or 0))
    d = by.setdefault(key, {"n": 0, "grid": [], "us": 0.0})
    d["n"] += 1; d["grid"].append(grid); d["us"] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print("| kernel | threads | LDS B | VGPR + AGPR | workgroups per CU by LDS / registers / wave slots | waves per SIMD | workgroups per launch (median) | rounds of the chip | us per launch |")
print("|---|---|---|---|---|---|---|---|---|")
for (n, wg, lds, v, a), d in sorted(by.items(), key=lambda kv: -kv[1]["us"]):
    waves = (wg + 63) // 64
    v, a = 2 * v, 2 * a                         # (rocprofv3 reports the counts of this wave64 target halved: 144 for rpn_tail_lin_kernel's 288)
    regs = max(((v + a + 7) // 8) * 8, 8)
    by_lds = 163840 // max(lds, 1) if lds else 99
    per_simd = 512 // regs                       # waves per SIMD by registers
    by_reg = (per_simd * 4) // waves if waves <= per_simd * 4 else 0
    by_slots = 32 // waves
    wpc = max(min(by_lds, by_reg, by_slots), 0)
    g = sorted(d["grid"])[len(d["grid"]) // 2]
    print("| `%s` | %d | %d | %d + %d | %s / %d / %d | %.1f | %d | %.2f | %.1f |" % (n, wg, lds, v, a, by_lds if lds else "-", by_reg, by_slots, wpc * waves / 4.0, g,
                                                                       g / max(wpc * 256, 1), d["us"] / d["n"]))
