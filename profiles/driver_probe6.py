"""Round 6: the whole driver (eval_rcnn.eval_scenes: loader processes, upload, engine, D2H, KITTI result files by writer processes) on the
three sources of bench.py's driver_leg -- the uniform synthetic generator, a KITTI-format tree of LiDAR-shaped sweeps with the host
sampler, the same tree with --device_input -- each beside the engine's own closed-loop rate on the same kind of scene, then a cProfile of the
parent process on the KITTI-tree run (tottime: where the feeding thread's time goes).
usage: python profiles/driver_probe6.py [scenes] [profile]"""
import cProfile, importlib, io, os, pstats, shutil, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if __name__ == "__main__":
    sys.path.insert(0, ROOT)
    import torch
    import bench
    PKG = bench.PKG
    C, E, K, S = (importlib.import_module(PKG + "." + m) for m in ("config", "eval_rcnn", "kitti_io", "synth"))
    cfg = C.default_eval_cfg(); dev = torch.device("cuda", 0); model = E.build_model(cfg, dev, seed=0)
    scenes = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 4096
    r = bench.driver_leg(cfg, model, dev, scenes=scenes)
    print("| driver | scenes/s (steady state) | loaders | feeding thread, ms per batch by phase |\n|---|---|---|---|")
    print("| uniform synthetic source, host sampler | %.0f | %s | %s |" % (r["value"], r["loaders"], r["host_phases_ms_per_batch"]))
    for k in ("lidar_kitti_tree", "lidar_kitti_tree_device_input"):
        if k in r:
            print("| %s | %.0f | %s | %s |" % (k, r[k]["value"], r[k]["loaders"], r[k]["host_phases_ms_per_batch"]))
    if "lidar_kitti_tree_error" in r:
        print("error:", r["lidar_kitti_tree_error"])
    # the engine alone on the clouds the KITTI tree's loader produces (closed loop, inputs resident)
    tree = tempfile.mkdtemp(prefix="prcnn_tree_")
    try:
        S.write_kitti_tree(tree, 64 + 16, pool=64)
        src = K.KittiSource(tree, cfg)
        batches = [torch.from_numpy(__import__("numpy").stack([src.load(8 * b + i)[0] for i in range(8)], 0)).to(dev) for b in range(8)]
        runner = E.make_runner(model, cfg, dev)
        def loop(k):
            for i in range(k):
                nxt = [batches[(i + d) % 8] for d in range(1, runner.depth + 1) if i + d < k]
                runner.submit(batches[i % 8], nxt)
            runner.drain(); torch.cuda.synchronize()
        loop(24)
        t0 = time.perf_counter(); loop(100); dt = time.perf_counter() - t0
        print("\nengine alone on the KITTI tree's sampled clouds, closed loop of 100 batches: %.0f scenes/s" % (800 / dt))
        if "profile" in sys.argv:
            big = tempfile.mkdtemp(prefix="prcnn_tree_")
            S.write_kitti_tree(big, 2048, pool=64)
            src2 = K.KittiSource(big, cfg)
            for dev_in in (False, True):
                out = tempfile.mkdtemp(prefix="prcnn_prof_"); stats = {}
                pr = cProfile.Profile(); pr.enable()
                E.eval_scenes(model, cfg, dev, src2, src2.ids, 8, out, device_input=dev_in, stats=stats)
                pr.disable(); shutil.rmtree(out, ignore_errors=True)
                s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22)
                print("\n## parent process, KITTI tree, device_input=%s: %.0f scenes/s under cProfile\n```\n%s\n```" % (dev_in, E.steady_state_rate(stats, 8), s.getvalue()[:5000]))
            shutil.rmtree(big, ignore_errors=True)
    finally:
        shutil.rmtree(tree, ignore_errors=True)
