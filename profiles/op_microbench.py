"""fused ball_query + group (prcnn_query_and_group) on one MI355X over
B in {1,8}, N=16384, M=4096, r in {0.1,0.2,0.4}, nsample in {32,64}, C in {0,1,128}; plus the un-subsampled dense-cloud
sizes of config C5 (N=131072 / 180000).  Prints a markdown table (HIP-event time per launch, achieved algorithmic GB/s,
fraction of the 8 TB/s HBM peak).   usage: python profiles/op_microbench.py > profiles/r01_query_and_group_sweep.md"""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
pkg = importlib.import_module(bench.PKG); sys.path.insert(0, pkg.DROPIN_DIR)
import pointnet2_cuda as P
synth = importlib.import_module(bench.PKG + ".synth")
dev = torch.device("cuda", 0)


def run(B, N, M, C, NS, R, reps=10):
    if N == 16384:
        xyz = torch.from_numpy(synth.scenes(B, N, seed0=1000)).to(dev)
    else:
        xyz = torch.from_numpy(np.stack([synth.dense_scene(1000 + i, N) for i in range(B)])).to(dev)
    temp = torch.full((B, N), 1e10, device=dev); sel = torch.empty((B, M), dtype=torch.int32, device=dev)
    P.furthest_point_sampling_wrapper(B, N, M, xyz, temp, sel)
    new_xyz = torch.gather(xyz, 1, sel.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    feats = torch.randn((B, C, N), device=dev) if C else None
    idx = torch.empty((B, M, NS), dtype=torch.int32, device=dev); out = torch.empty((B, 3 + C, M, NS), device=dev)
    for _ in range(3):
        P.query_and_group_wrapper(B, N, M, C, R, NS, new_xyz, xyz, feats, idx, out)
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        a.record(); P.query_and_group_wrapper(B, N, M, C, R, NS, new_xyz, xyz, feats, idx, out); b.record()
    torch.cuda.synchronize()
    ms = float(np.median([a.elapsed_time(b) for a, b in evs]))
    nbytes = B * bench.algorithmic_bytes_qg(N, M, C, NS)
    return ms, nbytes / (ms * 1e-3) / 1e9, nbytes


print("# prcnn_query_and_group (ball query + grouping, channel-major drop-in form) on one MI355X, HIP-event medians\n")
print("| B | N | M | r | nsample | C | algorithmic MB | us / launch | GB/s | frac of 8 TB/s |\n|---|---|---|---|---|---|---|---|---|---|")
for B in (1, 8):
    for R in (0.1, 0.2, 0.4):
        for NS in (32, 64):
            for C in (0, 1, 128):
                ms, gbs, nb = run(B, 16384, 4096, C, NS, R)
                print("| %d | 16384 | 4096 | %g | %d | %d | %.1f | %.1f | %.0f | %.3f |" % (B, R, NS, C, nb / 1e6, ms * 1e3, gbs, gbs / 8000))
for N in (131072, 180000):
    for C in (0, 128):
        ms, gbs, nb = run(2, N, 4096, C, 32, 0.2)
        print("| 2 | %d | 4096 | 0.2 | 32 | %d | %.1f | %.1f | %.0f | %.3f |" % (N, C, nb / 1e6, ms * 1e3, gbs, gbs / 8000))
