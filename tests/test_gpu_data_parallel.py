"""Data-parallel update with REAL engines: two processes share the one MI355X of the test box, each owns an
engine with half of the minibatch, and exchange the flat gradient bucket (a device tensor) over gloo --
`grasp_rl.parallel.DataParallelSac`, i.e. bench.py's N > 1 path minus RCCL itself.  Must equal one engine
updating on the whole minibatch."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import parity_util as pu
from grasp_rl import _capi

pytestmark = pytest.mark.gpu
B, STEPS = 16, 3


def _case():
    return pu.make_case(extractor="augmented", kind="depth", B=B, n_replay=48, n_steps=STEPS)


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from grasp_rl.parallel import DataParallelSac
    case = _case()
    cfg = _capi.GrlConfig.from_buffer_copy(case["cfg"])
    cfg.batch_size = B // world
    case["cfg"] = cfg
    eng = pu.engine_setup(case)
    dp = DataParallelSac(eng, overlap=True)
    assert dp.staged and dp.overlap                # two-bucket schedule: the dense bucket travels on a second stream
    if rank != 0:                                  # replicas must start identical: perturb, then broadcast
        P = eng.get_parameters()
        P["model/pi/fc0/bias:0"] = P["model/pi/fc0/bias:0"] + 1.0
        eng.set_parameters(P)
    dp.broadcast_parameters(src=0)
    lo, hi = rank * (B // world), (rank + 1) * (B // world)
    dp.train(STEPS, case["idx"][:, lo:hi], case["eps"][:, lo:hi])
    eng.synchronize()
    P = eng.get_parameters()
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), **{k.replace("/", "|"): v for k, v in P.items()})
    eng.close()
    dist.destroy_process_group()


def test_two_ranks_on_one_device_equal_single_engine(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    case = _case()
    single = pu.engine_setup(case)
    single.train(STEPS, case["idx"], case["eps"])
    ref = single.get_parameters()
    single.close()
    r0 = np.load(os.path.join(str(tmp_path), "rank0.npz"))
    r1 = np.load(os.path.join(str(tmp_path), "rank1.npz"))
    lr = case["spec"].lr
    for k, v in ref.items():
        a, b = r0[k.replace("/", "|")], r1[k.replace("/", "|")]
        assert np.array_equal(a, b), "replicas diverged: " + k
        d = np.abs(a.astype(np.float64) - v)
        assert d.max() <= 0.3 * lr * STEPS + 1e-7 and d.mean() <= 0.02 * lr * STEPS + 1e-9, (k, d.max(), d.mean())


def _rccl_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    from grasp_rl.parallel import DataParallelSac
    from grasp_rl.engine import SacEngine
    case = _case()
    cfg = _capi.GrlConfig.from_buffer_copy(case["cfg"])
    cfg.batch_size = B // world
    case["cfg"] = cfg
    outs = []
    for overlap in (True, False):                  # both schedules, over RCCL on the engines' own streams
        eng = SacEngine(case["cfg"], device="cuda:%d" % rank)
        eng.set_parameters(case["params"])
        st = case["stats"]
        eng.set_obs_stats(st["mean"], st["var"], st["ret_var"])
        tr = case["tr"]
        eng.replay_add(tr["obs"], tr["act"], tr["rew"], tr["next_obs"], tr["done"])
        dp = DataParallelSac(eng, overlap=overlap)
        assert dp.overlap == overlap
        lo, hi = rank * (B // world), (rank + 1) * (B // world)
        dp.train(STEPS, case["idx"][:, lo:hi], case["eps"][:, lo:hi])
        eng.synchronize()
        outs.append(eng.get_parameters())
        eng.close()
    for k in outs[0]:
        assert np.array_equal(outs[0][k], outs[1][k]), "schedules differ: " + k
    np.savez(os.path.join(out_dir, "rccl%d.npz" % rank), **{k.replace("/", "|"): v for k, v in outs[0].items()})
    dist.destroy_process_group()


def _check_against_single(tmp_path, world, prefix):
    case = _case()
    single = pu.engine_setup(case)
    single.train(STEPS, case["idx"], case["eps"])
    ref = single.get_parameters()
    single.close()
    parts = [np.load(os.path.join(str(tmp_path), "%s%d.npz" % (prefix, r))) for r in range(world)]
    lr = case["spec"].lr
    for k, v in ref.items():
        for p in parts[1:]:
            assert np.array_equal(parts[0][k.replace("/", "|")], p[k.replace("/", "|")]), "replicas diverged: " + k
        d = np.abs(parts[0][k.replace("/", "|")].astype(np.float64) - v)
        assert d.max() <= 0.3 * lr * STEPS + 1e-7 and d.mean() <= 0.02 * lr * STEPS + 1e-9, (k, d.max(), d.mean())


def test_rccl_single_rank_group_runs_both_schedules(tmp_path):
    """backend "nccl" (= RCCL) with a one-rank group on this box's GPU: the collectives are issued on the engine /
    exchange streams exactly as on N GPUs; result must equal the plain single-engine update."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_rccl_worker, args=(1, port, str(tmp_path)), nprocs=1, join=True)
    _check_against_single(tmp_path, 1, "rccl")


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL over xGMI)")
def test_rccl_two_gpus_equal_single_engine(tmp_path):
    """Two ranks on two GPUs, gradients all-reduced by RCCL (two-bucket overlapped and single-bucket schedules)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_rccl_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    _check_against_single(tmp_path, 2, "rccl")


def _ingraph_worker(rank, world, port, out_dir, overlap=False):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from grasp_rl.parallel import DataParallelInGraph, DataParallelSac
    case = _case()
    cfg = _capi.GrlConfig.from_buffer_copy(case["cfg"])
    cfg.batch_size = B // world
    case["cfg"] = cfg
    lo, hi = rank * (B // world), (rank + 1) * (B // world)
    # reference: the same two ranks exchanging through gloo (single bucket: compute -> all_reduce -> apply)
    ref = pu.engine_setup(case)
    dpr = DataParallelSac(ref, overlap=False)
    dpr.train(STEPS, case["idx"][:, lo:hi], case["eps"][:, lo:hi])
    ref.synchronize()
    Pref = ref.get_parameters()
    ref.close()
    # the in-graph exchange over IPC-mapped buffers
    eng = pu.engine_setup(case)
    dp = DataParallelInGraph(eng, overlap=overlap)
    dp.train(STEPS, case["idx"][:, lo:hi], case["eps"][:, lo:hi])
    assert dp.check() == STEPS
    P = eng.get_parameters()
    for k in P:
        assert np.array_equal(P[k], Pref[k]), "in-graph exchange differs from the gloo exchange: " + k
    # and on the device RNG, several updates per call
    dp.train(5)
    assert dp.check() == STEPS + 5
    P = eng.get_parameters()
    np.savez(os.path.join(out_dir, "ig%d.npz" % rank), **{k.replace("/", "|"): v for k, v in P.items()})
    eng.close()
    dist.destroy_process_group()


def test_overlapped_in_graph_exchange_two_processes_on_one_gpu(tmp_path):
    """grl_allreduce_set_overlap: the staged plan with the dense bucket's exchange on a side lane of the graph (channel 0)
    and the convolution bucket's after it (channel 1).  Same sums: bit-identical to the gloo exchange and between the
    replicas, explicit minibatches and device RNG."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_ingraph_worker, args=(2, port, str(tmp_path), True), nprocs=2, join=True)
    r0 = np.load(os.path.join(str(tmp_path), "ig0.npz"))
    r1 = np.load(os.path.join(str(tmp_path), "ig1.npz"))
    for k in r0.files:
        assert np.array_equal(r0[k], r1[k]), "replicas diverged: " + k


def test_in_graph_exchange_two_processes_on_one_gpu(tmp_path):
    """Two processes on the box's one MI355X map each other's exchange buffer (hipIpc) and run the data-parallel update
    with the hand-written two-shot all-reduce inside the graph: bit-identical to the same two ranks exchanging through
    gloo, replicas bit-identical to each other (also after five more updates on the device RNG)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_ingraph_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = np.load(os.path.join(str(tmp_path), "ig0.npz"))
    r1 = np.load(os.path.join(str(tmp_path), "ig1.npz"))
    for k in r0.files:
        if not k.startswith("model|pi|") and not k.startswith("model|values_fn") and not k.startswith("target"):
            continue
    # the ranks sample DIFFERENT shards on the device RNG (seeds differ by rank in bench.py; here the same seed and the
    # same replay contents, so even the device-RNG updates must leave identical replicas)
    for k in r0.files:
        assert np.array_equal(r0[k], r1[k]), "replicas diverged: " + k
