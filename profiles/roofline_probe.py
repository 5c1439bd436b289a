import sys; sys.path.insert(0,"/root/repo")
import bench, torch
r=bench.roofline_query_and_group(torch.device("cuda:0"), reps=10)
print(r["launch_ms"], r["frac"])
