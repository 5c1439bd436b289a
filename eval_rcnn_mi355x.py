#!/usr/bin/env python3
"""Command-line entry of the MI355X eval harness (the package name starts with a digit, so it cannot be
run with ``python -m``).  Mirrors the flags of the reference's pointrcnn/tools/eval_rcnn.py that apply to
inference:  --cfg_file --eval_mode --ckpt --batch_size --output_dir --set K V ...  plus --data_root/--split
(KITTI tree) or --scenes N (synthetic).  Multi-GPU: launch with torch.distributed.run, one process per GPU."""
import importlib
import os
import sys
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")   # before the HIP runtime starts: see 3d_adapt_auto_driving_amd/__init__.py (graph replay)

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
if __name__ == "__main__":      # (loader / writer processes re-import this file: they must not run main again)
    importlib.import_module("3d_adapt_auto_driving_amd.eval_rcnn").main()
