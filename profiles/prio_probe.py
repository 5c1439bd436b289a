"""A/B of the feature stream's HIP priority: bench.py's loop with the caller's stream -- the stream the RPN / RCNN feature graphs and the
final stage replay on -- replaced by a stream of the given priority (-1 = high; the side and proposal streams keep theirs).
Question: the feature stream is the longest chain of a step (0.98 of 1.15 ms) and its MFMA kernels take 1.3-2.5 x their
solo time beside the geometry streams' kernels, which have slack -- does placing its workgroups first shorten the step?
usage: python profiles/prio_probe.py <priority | none> [bench.py arguments]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
prio = sys.argv[1]
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
import bench                      # noqa: E402  (sets DEBUG_CLR_GRAPH_PACKET_CAPTURE before the runtime starts)
import torch                      # noqa: E402

if prio != "none":
    print("priority range", torch.cuda.Stream.priority_range(), file=sys.stderr)
    torch.cuda.set_stream(torch.cuda.Stream(priority=int(prio)))
bench.main()
