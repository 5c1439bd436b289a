"""BASELINE.json configs[1] / SURVEY.md section 8d config C2: the reference operator pair ball_query + grouping
(prcnn_query_and_group: QueryAndGroup of pointnet2_utils.py:241-264 in one call) on one MI355X over
B in {1, 8} x r in {0.1, 0.2, 0.4} x nsample in {32, 64} x C in {0, 1, 128}, N = 16384, M = 4096 (FPS centres of the scene).

  python profiles/qg_sweep.py [uniform|lidar] [reps]            -> HIP-event medians per configuration (markdown on stdout)
  rocprofv3 --kernel-trace --output-format csv -d D -- python profiles/qg_sweep.py uniform 5
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d F -- python profiles/qg_sweep.py uniform 5      (and WRITE_SIZE)
  python profiles/qg_sweep_summarize.py D/*/*kernel_trace.csv F/*/*counter_collection.csv W/*/*counter_collection.csv
                                         -> per configuration: per-KERNEL average duration and HBM traffic (profiles/r04_query_and_group_sweep.md)

Every configuration issues exactly 1 + reps calls of the operator, in the order of CONFIGS below: the summariser cuts the trace
into calls at every `dense_build_kernel` / first kernel of a call and assigns them to configurations by position."""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
pkg = importlib.import_module(bench.PKG); sys.path.insert(0, pkg.DROPIN_DIR)
import pointnet2_cuda as P
synth = importlib.import_module(bench.PKG + ".synth")
dev = torch.device("cuda", 0)
CONFIGS = [(B, R, NS, C) for B in (1, 8) for R in (0.1, 0.2, 0.4) for NS in (32, 64) for C in (0, 1, 128)]


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "uniform"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    make = synth.lidar_scenes if kind == "lidar" else synth.scenes
    N, M = 16384, 4096
    xyz8 = torch.from_numpy(make(8, N, seed0=1000)).to(dev)
    temp = torch.full((8, N), 1e10, device=dev); sel = torch.empty((8, M), dtype=torch.int32, device=dev)
    P.furthest_point_sampling_wrapper(8, N, M, xyz8, temp, sel)
    new8 = torch.gather(xyz8, 1, sel.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    feats8 = torch.randn((8, 128, N), device=dev)
    torch.cuda.synchronize()
    print("# prcnn_query_and_group on one MI355X, %s scenes, HIP-event medians of %d calls (N = 16384, M = 4096 FPS centres)\n" % (kind, reps))
    print("| B | r | nsample | C | algorithmic MB | us / call | GB/s | frac of 8 TB/s |\n|---|---|---|---|---|---|---|---|")
    for B, R, NS, C in CONFIGS:
        xyz, new_xyz = xyz8[:B].contiguous(), new8[:B].contiguous()
        feats = feats8[:B, :C].contiguous() if C else None
        idx = torch.empty((B, M, NS), dtype=torch.int32, device=dev); out = torch.empty((B, 3 + C, M, NS), device=dev)
        call = lambda: P.query_and_group_wrapper(B, N, M, C, R, NS, new_xyz, xyz, feats, idx, out)
        call()
        torch.cuda.synchronize()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for a, b in evs:
            a.record(); call(); b.record()
        torch.cuda.synchronize()
        ms = float(np.median([a.elapsed_time(b) for a, b in evs]))
        nb = B * bench.algorithmic_bytes_qg(N, M, C, NS)
        gbs = nb / (ms * 1e-3) / 1e9
        print("| %d | %g | %d | %d | %.2f | %.1f | %.0f | %.3f |" % (B, R, NS, C, nb / 1e6, ms * 1e3, gbs, gbs / 8000), flush=True)


if __name__ == "__main__":
    main()
