"""The drop-in path proper under a kernel trace: the nn.Module graph in the reference's operation order over the compiled
dropin_native modules (eval_rcnn.reference_api_only), batches of 8 scenes.
  rocprofv3 --kernel-trace --stats --output-format csv -d D -- python profiles/dropin_path_probe.py [steps]
  python profiles/dropin_path_probe.py summarize D/*/*kernel_trace.csv [steps]     -> per-kernel time per step"""
import collections, csv, importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "summarize":
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    d = collections.defaultdict(list)
    rows = list(csv.DictReader(open(sys.argv[2])))
    for r in rows:
        d[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    warm = 2                                                    # two warm-up steps precede the `steps` timed ones
    tot = sum(sum(v) for v in d.values()) / (steps + warm)
    print("| kernel | launches per step | us per step | avg us |\n|---|---|---|---|")
    for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:40]:
        print("| `%s` | %.1f | %.0f | %.1f |" % (k[:100], len(v) / (steps + warm), sum(v) / (steps + warm), sum(v) / len(v)))
    print("\nsum of kernel durations per step (8 scenes): %.1f ms, %d launches per step" % (tot / 1e3, len(rows) / (steps + warm)))
    sys.exit(0)
import torch
import bench
E = importlib.import_module(bench.PKG + ".eval_rcnn"); C = importlib.import_module(bench.PKG + ".config")
synth = importlib.import_module(bench.PKG + ".synth")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda", 0)
cfg = C.default_eval_cfg()
model = E.build_model(cfg, dev, seed=0)
batches = [torch.from_numpy(synth.scenes(8, 16384, seed0=9000 + 8 * k)).to(dev) for k in range(2)]
with E.reference_api_only(native=True):
    for k in range(2):
        E.infer_batch(model, cfg, batches[k])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        det = E.infer_batch(model, cfg, batches[k % 2])
        host = [det[key].cpu() for key in ("boxes", "scores", "num")]
    torch.cuda.synchronize()
    print("reference-order module path over dropin_native: %.1f scenes/s (%.1f ms per batch of 8)" % (
        steps * 8 / (time.perf_counter() - t0), (time.perf_counter() - t0) / steps * 1e3))
