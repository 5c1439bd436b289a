"""Probe used for the rocprofv3 PMC passes over the fused SA-MLP MFMA kernel (profiles/r01_pmc_sa_mlp_fused.json)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, torch
r = bench.roofline_sa_mlp_fused(torch.device("cuda:0"), reps=5)
print(r["launch_ms"], r["achieved"], r["frac"])
