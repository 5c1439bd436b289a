"""world_size-2 CPU test (gloo) of the multi-GPU path: scene sharding + the single all_gather of
padded detection tables (eval_rcnn.shard_scene_ids / pack_detections / all_gather_detections).
The same code runs over RCCL (backend "nccl") on the GPU node; only the backend string differs."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_scenes, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from conftest import pkg
        from test_host_logic import tiny_model
        from oracle import ext_cpu, oracle as O
        O.set_num_threads(2)
        E = pkg("eval_rcnn")
        model, cfg, _ = tiny_model()
        ids = E.shard_scene_ids(n_scenes, rank, world)
        with ext_cpu.patch_package():
            table, counts = E.eval_synthetic(model, cfg, "cpu", ids, batch_size=2, npoints=2048)
        table, counts = E.all_gather_detections(table, counts, torch.device("cpu"))
        if rank == 0:
            np.savez(os.path.join(out_dir, "gathered.npz"), table=table.numpy(), counts=counts.numpy())
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_sharded_eval_all_gather_equals_single_process(tmp_path, oracle):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import pkg
    from test_host_logic import tiny_model
    from oracle import ext_cpu
    E = pkg("eval_rcnn")
    n_scenes = 5                                   # odd: ranks get 3 and 2 scenes -> padded gather
    assert E.shard_scene_ids(5, 0, 2) == [0, 2, 4] and E.shard_scene_ids(5, 1, 2) == [1, 3]
    mp.spawn(_worker, args=(2, _free_port(), n_scenes, str(tmp_path)), nprocs=2, join=True)
    got = np.load(tmp_path / "gathered.npz")
    model, cfg, _ = tiny_model()
    with ext_cpu.patch_package():
        table, counts = E.eval_synthetic(model, cfg, "cpu", list(range(n_scenes)), batch_size=2, npoints=2048)
    # the gathered table lists rank 0's scenes then rank 1's: sort both by scene id (column 8)
    order = np.argsort(got["table"][:, 0, 8], kind="stable")
    assert got["table"].shape == (n_scenes, cfg.TEST.RPN_POST_NMS_TOP_N, 9)
    assert np.array_equal(got["table"][order][:, 0, 8], np.arange(n_scenes))
    assert np.array_equal(got["counts"][order], counts.numpy())
    np.testing.assert_allclose(got["table"][order], table.numpy(), rtol=0, atol=1e-6)
    assert counts.sum() > 0


def test_all_gather_is_identity_without_process_group():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import pkg
    E = pkg("eval_rcnn")
    t, c = torch.zeros((2, 4, 9)), torch.tensor([1, 0], dtype=torch.int32)
    t2, c2 = E.all_gather_detections(t, c, torch.device("cpu"))
    assert t2 is t and c2 is c
