"""Every PRCNN_* environment switch of the package, the C library and bench.py in ONE table ("a lot of surface").

kind:
  operational  sizes the runner / the host side for a deployment (streams, slots, worker processes): the ones a user may have to touch
  numerics     changes the association of a sum or routes a layer through a library GEMM: results move inside BASELINE's 1e-4 box
               tolerance; everything the parity tests pin runs with these at their defaults
  ab           A/B switch kept so that a measurement quoted in DESIGN.md / profiles/ can be repeated: SAME results, bit for bit, either
               way (each has a test or a shadow check behind it); the default is the measured-faster form
  tuning       a launch-geometry knob of one kernel (grid, tile or cell counts): same results
  debug, bench what the names say
All of them are read ONCE (module import / first call of the C entry that owns them): set them before the process starts.
`check_environment()` (called by the package's __init__) warns about PRCNN_* variables that are not in this table -- a typo would
otherwise be silently ignored; `python -m 3d_adapt_auto_driving_amd.switches` prints the table and what is set.
tests/test_host_logic.py::test_every_switch_is_registered keeps the table and the sources in step."""
import os
import warnings

# name: (kind, default, read by, meaning)
SWITCHES = {
    # ---- operational
    "PRCNN_GRAPHS": ("operational", "1", "eval_rcnn.py", "0: eager enqueue (PipelinedRunner) instead of hipGraph replay (GraphedRunner)"),
    "PRCNN_GRAPH_SLOTS": ("operational", "depth/group+1", "eval_rcnn.py", "group slots of the graphed runner (>= 2; 5.9 GB of HBM each at batch 8)"),
    "PRCNN_GEO_GROUP": ("operational", "4", "eval_rcnn.py", "batches per geometry chain"),
    "PRCNN_PAIR": ("operational", "2", "eval_rcnn.py", "batches of a geometry group that share the launches of the RPN / proposal / RCNN / final stages in the graphed runner (must divide the group; detections come back up to 2 x pair - 1 submits late)"),
    "PRCNN_GEO_DEPTH": ("operational", "3*group", "eval_rcnn.py", "batches the geometry runs ahead"),
    "PRCNN_SIDE_STREAMS": ("operational", "2", "eval_rcnn.py", "geometry side streams (with feature + proposal stream: the 4 hardware queues)"),
    "PRCNN_LOADER_THREADS": ("operational", "1", "eval_rcnn.py", "threads of the numeric libraries (BLAS / OpenMP) in every loader and writer process; 0: leave them alone (round 6: a loader's small numpy products fanned out over the whole host's thread pool)"),
    "PRCNN_RAW_SLOT_POINTS": ("operational", "200000", "eval_rcnn.py", "points per raw cloud a slot of the loaders' shared page-locked buffer holds with --device_input (slot = batch x points x 16 bytes; a larger cloud raises)"),
    "PRCNN_LOADER_WORKERS": ("operational", "budget", "eval_rcnn.py", "loader processes of eval_scenes (default: host_budget)"),
    "PRCNN_WRITER_PROCS": ("operational", "budget", "eval_rcnn.py", "KITTI result writer processes"),
    "PRCNN_LOADER_CONTEXT": ("operational", "forkserver/fork", "eval_rcnn.py", "multiprocessing start method of loaders and writers"),
    "PRCNN_NO_AFFINITY": ("operational", "unset", "eval_rcnn.py", "1: do not pin a rank to its share of the host cores"),
    "PRCNN_RESULT_LAG": ("operational", "3", "eval_rcnn.py", "batches between submitting a batch and reading its detections on the host"),
    "PRCNN_GRAPHS_FORCE": ("debug", "unset", "__init__.py", "1: replay graphs although DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 was not in place (profiles/graph_fault_probe.py)"),
    "PRCNN_GRAPH_DEBUG": ("debug", "0", "eval_rcnn.py", "bit mask: device syncs + prints around the graph replays"),
    # ---- numerics
    "PRCNN_SPLIT_BF16": ("numerics", "unset", "net/fast_infer.py", "1: EXPERIMENT -- the plain per-point layers (point_layer: FP modules' second layers, coarse products, RCNN heads) on the bf16 matrix cores with every operand split exactly into three bf16 pieces (csrc/split_bf16.hip): ~1e-7 relative to the f32 fma chain, not its bits; measured in profiles/r06_split_bf16.md, never the headline"),
    "PRCNN_NO_FP_LINEAR": ("numerics", "unset", "net/fast_infer.py", "FP layer 1 over the interpolated tensor (reference association) instead of interp(W f)"),
    "PRCNN_LIB_GEMM": ("numerics", "unset", "net/fast_infer.py", "per-point layers through torch (library GEMM) instead of csrc/packed_layer.hip"),
    "PRCNN_ALLOW_LIB_GEMM": ("numerics", "unset", "net/fast_infer.py", "1: permit a library GEMM for a shape the layer kernels do not cover (else: error)"),
    "PRCNN_TAIL_NARROW": ("numerics", "1", "csrc/rpn_tail.hip", "0: the RPN regression head's last layer as a zero-padded 128-column stage (rounds 2-4) instead of 64 columns on 32x32x2 + 16 on v_mfma_f32_16x16x4_f32 (another k order in columns 64..75, ~1e-7 relative; the oracle stand-in reads the same switch)"),
    "PRCNN_NO_PACK": ("numerics", "unset", "net/fast_infer.py", "all grouped rows instead of the distinct rows (prcnn_ball_pack) -- and, without the packed row lists, the per-point layers through the GEMM library (needs PRCNN_ALLOW_LIB_GEMM=1 on a covered network): the all-rows leg of bench.py"),
    "PRCNN_ROWS_GEMM": ("numerics", "unset", "net/fast_infer.py", "128-wide row layers through rows_gemm128 (round-1 form)"),
    # ---- scheduling A/B (same results)
    "PRCNN_EARLY_LEVELS": ("ab", "4", "net/fast_infer.py", "leading SA levels computed with the geometry"),
    "PRCNN_EARLY_FP": ("ab", "2", "net/fast_infer.py", "coarsest FP modules computed with the geometry (round 5: 2; 3 until then)"),
    "PRCNN_EARLY_G0": ("ab", "1", "net/fast_infer.py", "the finest FP module's coarse product computed with the geometry (round 4: +1.2 %)"),
    "PRCNN_NO_XYZ_EARLY": ("ab", "unset", "net/fast_infer.py", "1: no SA level rides with the geometry"),
    "PRCNN_NO_GROUP_SA": ("ab", "unset", "net/fast_infer.py", "1: early SA levels per batch instead of per geometry group"),
    "PRCNN_FINAL_ON_FEATURE": ("ab", "0", "eval_rcnn.py", "1: final stage behind the RCNN features on the feature stream (round 4's default); 0: on the proposal stream"),
    "PRCNN_NO_RCNN_SPLIT": ("ab", "unset", "eval_rcnn.py", "1: RCNN geometry on the feature stream"),
    # ---- engine formulations A/B (same results)
    "PRCNN_NO_POOL_DEDUP": ("ab", "unset", "net/fast_infer.py", "RCNN point MLP over all 512 pooled rows"),
    "PRCNN_NO_CENTRE_DEDUP": ("ab", "unset", "net/fast_infer.py", "no representative map over sampled centres"),
    "PRCNN_NO_CENTRE_SKIP": ("ab", "unset", "net/fast_infer.py", "copies of a centre keep rows of their own"),
    "PRCNN_NO_POOL_GROUPS": ("ab", "unset", "net/fast_infer.py", "RoI pooling sweeps all points (no spatial groups)"),
    "PRCNN_NO_POINT_MLP": ("ab", "unset", "net/fast_infer.py", "RCNN entrance as separate layers"),
    "PRCNN_NO_ROI_GEOMETRY": ("ab", "unset", "net/fast_infer.py", "RCNN sampling / ball queries as six launches"),
    "PRCNN_NO_CENTRE_ROWS": ("ab", "unset", "net/fast_infer.py", "the RCNN second level's per-point layer over all 128 level-1 centres of every RoI instead of the listed representatives"),
    "PRCNN_NO_POOLED_ROWS": ("ab", "unset", "net/fast_infer.py", "the RCNN entrance over whole 64-row tiles per RoI (prcnn_pooled_tiles) instead of the list of distinct pooled rows"),
    "PRCNN_NO_ROI_PACKS": ("ab", "unset", "net/fast_infer.py", "the RoI clouds' two row lists by prcnn_ball_pack_ex launches instead of inside prcnn_rcnn_roi_geometry_packs"),
    "PRCNN_NO_RPN_TAIL": ("ab", "unset", "net/fast_infer.py", "finest FP module and RPN heads layer by layer (the bits of the fused tail under PRCNN_NO_FP_LINEAR=1: the layer-by-layer form keeps the reference's association)"),
    "PRCNN_TAIL_DECODE": ("ab", "1", "net/fast_infer.py", "0: the fused RPN tail stores the (B, N, 76) regression rows and the proposal layer decodes them (rpn_decode_kernel) instead of decoding inside the tail kernel (round 5)"),
    "PRCNN_NO_SA2_BATCH": ("ab", "unset", "net/fast_infer.py", "set: the two 128-wide scales of an MSG level (RPN SA2) as two launches per stage instead of one (prcnn_sa_packed_mlp_batch, round 5)"),
    "PRCNN_NO_WIDE_FUSED3": ("ab", "unset", "net/fast_infer.py", "1: the GroupAll level's layer 1 as a per-point launch in front of csrc/sa_wide.hip instead of inside csrc/sa_wide3.hip"),
    # ---- kernel forms A/B (C library; same results)
    # ---- tuning
    "PRCNN_MFMA_GRID": ("tuning", "512", "csrc/capi.hip", "workgroups of a persistent MFMA launch"),
    "PRCNN_SA_GRID": ("tuning", "512", "csrc/sa_packed.hip", "workgroups of the packed SA kernels"),
    "PRCNN_SA_TILES": ("tuning", "0", "csrc/sa_mlp_fused.hip", "tiles per workgroup of the fused SA kernel (0 = tickets)"),
    "PRCNN_PL_STREAM_CAP": ("tuning", "512", "csrc/packed_layer.hip", "workgroups of the persistent layer kernels"),
    "PRCNN_PL_STREAM_MIN": ("tuning", "512", "csrc/packed_layer.hip", "items from which a K = 128 layer runs persistently"),
    "PRCNN_PL_PERSIST_MIN": ("tuning", "256", "csrc/packed_layer.hip", "items from which a K >= 256 layer runs persistently (and more than the cap)"),
    "PRCNN_FPS2_CAPACITY": ("tuning", "auto", "csrc/fps.hip", "co-resident fps_spec2_kernel workgroups the device is assumed to hold (default: occupancy query x CUs); a launch takes half of it, below 32 the sampling of 16384 < n <= 32768 points falls back to fps_generic_kernel (tests: 64 / 0)"),
    "PRCNN_TAIL_GRID": ("tuning", "256", "csrc/rpn_tail.hip", "workgroups of the fused RPN tail (one per CU: its waves hold a SIMD's whole register file, no other kernel shares a CU with it); 192-224 leave CUs to the other streams: +0.8-1.0 % at K = 100, level at K = 20 (round 6)"),
    "PRCNN_FPS_LDS_PAD": ("tuning", "84", "csrc/fps.hip", "KB of dynamic LDS an FPS workgroup claims (keeps its CU to itself)"),
    "PRCNN_TNN_CELLS": ("tuning", "2", "csrc/three_nn_grid.hip", "grid cells per known point"),
    "PRCNN_GROUP_CHUNKS": ("tuning", "auto", "csrc/ball_group.hip", "channel chunks of the grouping kernels"),
    "PRCNN_GROUP_ROWS": ("tuning", "auto", "csrc/ball_group.hip", "rows per workgroup of the LDS grouping kernels"),
    # ---- bench.py
    "PRCNN_BENCH_BATCH": ("bench", "8", "bench.py", "scenes per step (echoed in config.env_overrides)"),
    "PRCNN_BENCH_LAG": ("bench", "3", "bench.py", "host result lag of the timed loop"),
    "PRCNN_BENCH_SHARE_GPU": ("bench", "unset", "bench.py", "1: all ranks on GPU 0 (CPU-side rehearsal of N > 1)"),
    "PRCNN_BENCH_TRACE": ("bench", "unset", "bench.py", "1: print the per-step host timeline"),
    "PRCNN_TAIL_OVERLAP": ("bench", "1", "bench.py", "0: runner.step (no software pipeline) instead of submit / flush"),
}


def set_in_environment():
    return {k: v for k, v in os.environ.items() if k.startswith("PRCNN_")}


def check_environment():
    """Warn about PRCNN_* variables nobody reads (typos)."""
    unknown = sorted(k for k in set_in_environment() if k not in SWITCHES)
    if unknown:
        warnings.warn("unknown PRCNN_* environment variables (ignored): %s -- see python -m 3d_adapt_auto_driving_amd.switches"
                      % ", ".join(unknown), RuntimeWarning, stacklevel=2)
    return unknown


def table():
    # the last column names the test that RUNS the switch in its non-default form
    held = {"ab": "`tests/test_gpu_switches.py::test_every_ab_switch_gives_the_same_detections` (bit for bit)",
            "numerics": "`tests/test_gpu_switches.py::test_numerics_switches_stay_inside_the_box_tolerance` (boxes within 1e-4)"}
    special = {"PRCNN_ALLOW_LIB_GEMM": "set together with the GEMM-library switches in the numerics test",
               "PRCNN_FPS2_CAPACITY": "`tests/test_gpu_ops.py::test_fps_two_workgroups_respects_the_co_resident_capacity`",
               "PRCNN_BENCH_SHARE_GPU": "`tests/test_gpu_configs.py::test_bench_two_ranks_on_one_gpu_...`, `::test_bench_eight_ranks_on_one_gpu`",
               "PRCNN_GRAPHS": "`tests/test_gpu_graphs.py` (replay == eager enqueue, bit for bit); every child of the switch tests runs with 0"}
    rows = ["| switch | kind | default | read by | meaning | run in its other form by |", "|---|---|---|---|---|---|"]
    for name, (kind, default, where, what) in sorted(SWITCHES.items(), key=lambda kv: (kv[1][0], kv[0])):
        rows.append("| `%s` | %s | %s | `%s` | %s | %s |" % (name, kind, default, where, what, special.get(name, held.get(kind, "-"))))
    return "\n".join(rows)


if __name__ == "__main__":
    print(table())
    cur = set_in_environment()
    print("\nset in this environment: %s" % (", ".join("%s=%s" % kv for kv in sorted(cur.items())) or "none"))
    check_environment()
