"""gpurun_out/pmc_inst_mix.txt (profiles/pmc_inst_mix.sh) -> profiles/r04_pmc_inst_mix.md: non-MFMA instructions per MFMA of the step's MFMA
kernels, priced with profiles/r04_coexec_probe.md (5.5 cycles per VALU instruction, 5 per vector-memory instruction behind a 64-cycle MFMA).
usage: python profiles/pmc_inst_mix_summarize.py gpurun_out/pmc_inst_mix.txt > profiles/r04_pmc_inst_mix.md"""
import re
import sys

txt = open(sys.argv[1]).read()
rows = []
for b in re.split(r'^## ', txt, flags=re.M)[1:]:
    lines = b.strip().split('\n')
    d = {}
    for l in lines[1:]:
        p = l.split()
        if p[0] == 'launches':
            d['launches'] = int(p[1]); d['dur'] = float(p[-1])
        else:
            d[p[0]] = float(p[1])
    if 'SQ_INSTS_MFMA' in d:
        rows.append((lines[0], d))
print("# Instruction mix of the product step's MFMA kernels (rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES | "
      "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES SQ_WAIT_INST_ANY, two passes over profiles/pmc_step_probe.py 6; single stream; "
      "profiles/pmc_inst_mix.sh)\n")
print("Wave-level instruction counts per launch (SQ_INSTS_VALU includes the MFMAs).  Price list of `r04_coexec_probe.md`: behind a 64-cycle f32 "
      "MFMA a VALU instruction of the same wave costs 4.5-8 cycles, a vector-memory instruction 4-6, up to four LDS / scalar instructions nothing.\n")
print("| kernel | launches | avg us | waves | MFMA | other VALU per MFMA | vector-memory per MFMA | LDS per MFMA | scalar per MFMA | "
      "issue cycles per MFMA (64 + 5.5 VALU + 5 VMEM) | ceiling | waiting (SQ_WAIT_INST_ANY / waves / duration) |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|")
for name, d in rows:
    mf = d['SQ_INSTS_MFMA']
    valu = (d['SQ_INSTS_VALU'] - mf) / mf
    vm = (d['SQ_INSTS_VMEM_RD'] + d['SQ_INSTS_VMEM_WR']) / mf
    cyc = 64 + 5.5 * valu + 5 * vm
    print("| `%s` | %d | %.1f | %d | %.3g | %.2f | %.2f | %.2f | %.2f | %.0f | %.2f | %.2f |" % (
        name, d['launches'], d['dur'], d['SQ_WAVES'], mf, valu, vm, d['SQ_INSTS_LDS'] / mf, d['SQ_INSTS_SALU'] / mf, cyc, 64 / cyc,
        d['SQ_WAIT_INST_ANY'] / d['SQ_WAVES'] / (d['dur'] * 2300)))
print("\n(`ceiling` = 64 / issue cycles: what the instruction stream alone allows before any stall; measured MfmaUtil: "
      "`r04_pmc_mfma_product_kernels.md`.)\n\n## raw\n\n```\n" + txt.strip() + "\n```")
