"""cProfile of the whole driver (eval_rcnn.eval_scenes with loader and writer processes) on synthetic scenes: where the
parent process spends its time per batch.  usage: python profiles/driver_profile.py [scenes]"""
import cProfile, importlib, os, pstats, shutil, sys, tempfile, io
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
PKG = "3d_adapt_auto_driving_amd"
C = importlib.import_module(PKG + ".config"); E = importlib.import_module(PKG + ".eval_rcnn"); K = importlib.import_module(PKG + ".kitti_io")
if __name__ == "__main__":
    dev = torch.device("cuda", 0); cfg = C.default_eval_cfg(); model = E.build_model(cfg, dev, seed=0)
    scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    src = K.SyntheticSource(cfg, scenes)
    for rep in range(2):
        out = tempfile.mkdtemp(prefix="prcnn_prof_"); stats = {}
        pr = cProfile.Profile()
        try:
            if rep: pr.enable()
            E.eval_scenes(model, cfg, dev, src, src.ids, 8, out, stats=stats)
            if rep: pr.disable()
        finally:
            shutil.rmtree(out, ignore_errors=True)
        print("rep", rep, "steady state %.1f scenes/s" % E.steady_state_rate(stats, 8))
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45); print(s.getvalue()[:9000])
