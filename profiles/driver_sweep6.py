"""Round 6: the whole driver on the KITTI-format tree of LiDAR-shaped sweeps against loader / writer process counts and the loaders'
numeric-library threads (PRCNN_LOADER_WORKERS / PRCNN_WRITER_PROCS / PRCNN_LOADER_THREADS), host sampler and --device_input.
One child process per setting.   usage: python profiles/driver_sweep6.py [scenes]"""
import importlib, os, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if __name__ != "__main__":
    pass
elif len(sys.argv) > 2 and sys.argv[1] == "run":
    sys.path.insert(0, ROOT)
    import torch
    import bench
    PKG = bench.PKG
    C, E, K = (importlib.import_module(PKG + "." + m) for m in ("config", "eval_rcnn", "kitti_io"))
    cfg = C.default_eval_cfg(); dev = torch.device("cuda", 0); model = E.build_model(cfg, dev, seed=0)
    tree, kind = sys.argv[2], sys.argv[3]
    src = K.KittiSource(tree, cfg) if kind != "uniform" else K.SyntheticSource(cfg, int(sys.argv[4]))
    rates = []
    for rep in range(2):
        out = tempfile.mkdtemp(prefix="prcnn_sweep_"); stats = {}
        E.eval_scenes(model, cfg, dev, src, src.ids, 8, out, device_input=(kind == "device"), stats=stats)
        shutil.rmtree(out, ignore_errors=True)
        rates.append(E.steady_state_rate(stats, 8))
    print("| %s | %s + %s | %s | %s |" % (kind, os.environ.get("PRCNN_LOADER_WORKERS", "budget"), os.environ.get("PRCNN_WRITER_PROCS", "budget"),
                                     os.environ.get("PRCNN_LOADER_THREADS", "1"), ", ".join("%.0f" % r for r in rates)), flush=True)
else:
    sys.path.insert(0, ROOT)
    S = importlib.import_module("3d_adapt_auto_driving_amd.synth")
    scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    tree = tempfile.mkdtemp(prefix="prcnn_tree_")
    S.write_kitti_tree(tree, scenes, pool=64)
    print("| source | loaders + writers | library threads per loader | scenes/s (two runs) |\n|---|---|---|---|", flush=True)
    settings = [("host", None, None, "1"), ("host", 12, 2, "1"), ("host", 18, 2, "1"), ("device", None, None, "1"), ("device", 12, 2, "1"), ("uniform", None, None, "1")]
    for kind, lw, wp, th in settings:
        env = dict(os.environ, PRCNN_LOADER_THREADS=th)
        if lw is not None:
            env.update(PRCNN_LOADER_WORKERS=str(lw), PRCNN_WRITER_PROCS=str(wp))
        subprocess.run([sys.executable, os.path.abspath(__file__), "run", tree, kind, str(scenes)], env=env)
    shutil.rmtree(tree, ignore_errors=True)
