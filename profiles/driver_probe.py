"""The whole driver (eval_rcnn.eval_scenes: loader processes, pinned upload, engine, D2H, KITTI result files by writer processes) at
several loader / writer process counts (PRCNN_LOADER_WORKERS / PRCNN_WRITER_PROCS override eval_rcnn.host_budget).
usage: python profiles/driver_probe.py"""
import importlib, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if __name__ != "__main__":
    pass                                    # re-imported by the fork server of the loader / writer processes: nothing to run
elif len(sys.argv) > 1:
    sys.path.insert(0, ROOT)
    import torch
    import bench
    C = importlib.import_module("3d_adapt_auto_driving_amd.config"); E = importlib.import_module("3d_adapt_auto_driving_amd.eval_rcnn")
    cfg = C.default_eval_cfg(); dev = torch.device("cuda", 0); model = E.build_model(cfg, dev, seed=0)
    r = bench.driver_leg(cfg, model, dev, scenes=2048)
    print("loaders %s writers %s: %.0f scenes/s  %s" % (os.environ.get("PRCNN_LOADER_WORKERS"), os.environ.get("PRCNN_WRITER_PROCS"), r["value"], r["host_budget"]), flush=True)
else:
    for lw, wp in ((16, 6), (32, 12), (48, 12), (64, 16)):
        env = dict(os.environ, PRCNN_LOADER_WORKERS=str(lw), PRCNN_WRITER_PROCS=str(wp))
        subprocess.run([sys.executable, os.path.abspath(__file__), "run"], env=env)
