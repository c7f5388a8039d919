"""Data-parallel update with REAL engines: two processes share the one MI355X of the test box, each owns an
engine with half of the minibatch, and exchange the flat gradient bucket (a device tensor) over gloo --
`grasp_rl.parallel.DataParallelSac`, i.e. bench.py's N > 1 path minus RCCL itself.  Must equal one engine
updating on the whole minibatch."""
import os
import socket
import time

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import parity_util as pu
from grasp_rl import _capi

pytestmark = pytest.mark.gpu
# The spawned workers build their cases with NumPy (orthogonal initialisation = an SVD of the 1024 x 512 dense kernels): with
# one BLAS thread per core of a 256-core box in each of 8 processes that took 97 s of a 115 s test (GRL_TEST_TIMING stamps);
# inherited by the children, no effect on this process (its NumPy is already loaded)
os.environ.setdefault("OMP_NUM_THREADS", "4")
os.environ.setdefault("OPENBLAS_NUM_THREADS", "4")
B, STEPS = 16, 3


def _case():
    return pu.make_case(extractor="augmented", kind="depth", B=B, n_replay=48, n_steps=STEPS)


def _init_gloo(rank, world, out_dir):
    """gloo over a FILE store: the TCP store's accept path looks every client's host name up, and on the GPU boxes (no resolver)
    each lookup waits for its time-out -- eight ranks spent 122 s of a 135 s test in `init_process_group` (stamps: GRL_TEST_TIMING)."""
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")      # (and gloo's own device: no lookup of the box's host name either)
    dist.init_process_group("gloo", init_method="file://" + os.path.join(out_dir, "gloo_store"), rank=rank, world_size=world)


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    _init_gloo(rank, world, out_dir)
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


# ---------------------------------------------------------------------------------------------------------------------
# the exchange inside the update graph (csrc/dp_kernels.h), W processes sharing the box's one MI355X
VARIANTS = (("oneshot", False), ("twoshot", False), ("twoshot", True))


def _rank_ordered_reference(case, lo, hi, world, group_steps):
    """The documented arithmetic of the exchange, formed on the host: every rank's gradient bucket (grl_compute_grads),
    added IN RANK ORDER in float32, applied with grad_scale 1 / world (grl_apply_grads) -- for two ranks also what a
    collective library's all-reduce gives (a + b), for more ranks its association differs."""
    ref = pu.engine_setup(case)
    for s in range(group_steps):
        ref.compute_grads(case["idx"][s:s + 1, lo:hi], case["eps"][s:s + 1, lo:hi])
        g = torch.from_numpy(ref.fetch("grads", (ref.n_trainable,)))
        parts = [torch.empty_like(g) for _ in range(world)]
        dist.all_gather(parts, g)
        total = parts[0].numpy().copy()
        for p in parts[1:]:
            total += p.numpy()
        ref.store("grads", total)
        ref.apply_grads(1.0 / world)
    ref.synchronize()
    P = ref.get_parameters()
    ref.close()
    return P


def _ingraph_worker(rank, world, port, out_dir):
    t_entry = time.time()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["GRL_TUNE"] = "dp_timeout_ms=20000"
    _init_gloo(rank, world, out_dir)
    from grasp_rl.parallel import DataParallelInGraph, DataParallelSac
    B = max(16, 4 * world)      # (a per-rank minibatch of at least 4 rows: below that the dense weight gradients leave the
    #                              vectorised kernel and the plan has no staged form to overlap)
    cases = {}
    # W = 2 and 4 run every variant on the CNN shape with the device-RNG / variant-switch extras; eight time-sliced processes
    # took two minutes for that, so W = 8 runs the CNN bucket (1 342 992 floats, 8 chunks of 167 874 -> rup 4) LIGHT: one-shot
    # and two-shot against the rank-ordered sum on the explicit minibatches only
    light = world >= 8
    slow = os.environ.get("GRL_SLOW_TESTS") == "1"
    # (eight time-sliced processes: 127 s for one-shot + two-shot over three updates -- GRL_SLOW_TESTS=1; the default suite runs
    # the CNN bucket at W = 8 with ONE variant, two-shot, over two updates: the shape the driver's 8-GPU run exchanges)
    n_steps = {"cnn": 2 if light and not slow else STEPS, "tiny": STEPS}
    cases["cnn"] = pu.make_case(extractor="augmented", kind="depth", B=B, n_replay=48, n_steps=n_steps["cnn"])
    if world >= 8:
        # the smallest SAC there is: a bucket of 84 floats -- not a multiple of 4 * world, rank 6's chunk short, rank 7's EMPTY
        # (capi.inl: chunk = rup(ceil(n / world), 4))
        cases["tiny"] = pu.make_case(extractor="mlp", obs_dim=1, act_dim=1, layers=(4,), B=B, n_replay=48, n_steps=STEPS)
    t_start = float(os.environ.get("GRL_TEST_T0", t_entry))      # (the parent's clock just before mp.spawn)
    stamp = (lambda what: print("[W=%d rank 0] %6.1f s  %s" % (world, time.time() - t_start, what), flush=True)) \
        if rank == 0 and os.environ.get("GRL_TEST_TIMING") == "1" else (lambda what: None)
    stamp("worker entered %.1f s after the spawn; process group + cases ready" % (t_entry - t_start))
    for cname, case in cases.items():
        stamp("case " + cname)
        cfg = _capi.GrlConfig.from_buffer_copy(case["cfg"])
        cfg.batch_size = B // world
        case["cfg"] = cfg
        lo, hi = rank * (B // world), (rank + 1) * (B // world)
        steps = n_steps[cname]
        Pref = _rank_ordered_reference(case, lo, hi, world, steps)
        stamp("rank-ordered reference done")
        if world == 2:      # ... and the same two ranks exchanging through gloo (single bucket: compute -> all_reduce -> apply)
            g = pu.engine_setup(case)
            DataParallelSac(g, overlap=False).train(STEPS, case["idx"][:, lo:hi], case["eps"][:, lo:hi])
            g.synchronize()
            Pg = g.get_parameters()
            g.close()
            for k in Pg:
                assert np.array_equal(Pg[k], Pref[k]), "rank-ordered sum differs from the gloo exchange: " + k
        finals = []
        for mode, overlap in VARIANTS:
            if overlap and (cname == "tiny" or light):
                continue                              # (vector observations have no staged plan)
            if light and cname == "cnn" and not slow and mode != "twoshot":
                continue
            eng = pu.engine_setup(case)
            stamp("%s: engine set up" % mode)
            dp = DataParallelInGraph(eng, overlap=overlap, mode=mode)
            stamp("%s: exchange connected" % mode)
            if cname == "tiny":
                assert eng.n_trainable == 84
            dp.train(steps, case["idx"][:, lo:hi], case["eps"][:, lo:hi])
            assert dp.check() == steps
            stamp("%s: %d explicit updates done" % (mode, steps))
            P = eng.get_parameters()
            for k in P:
                assert np.array_equal(P[k], Pref[k]), "%s / %s%s: differs from the rank-ordered sum: %s" % (cname, mode, "+overlap" if overlap else "", k)
            if light and cname == "cnn":
                finals.append(P)
                dp.close()
                eng.close()
                continue
            # and on the device RNG, several updates per call (the same seed and replay contents on every rank, so even
            # these must leave identical replicas), then a switch of the variant on the idle handle
            dp.train(5)
            dp.train(1)
            assert dp.check() == steps + 6
            dp.set_mode("oneshot" if mode == "twoshot" else "twoshot", False)
            dp.train(3)
            assert dp.check() == steps + 9
            finals.append(eng.get_parameters())
            dp.close()            # (collective: no rank frees its exchange memory while a peer may still read it)
            eng.close()
        for P in finals[1:]:
            for k in P:
                assert np.array_equal(P[k], finals[0][k]), "variants differ after the device-RNG updates: " + k
        np.savez(os.path.join(out_dir, "ig_%s_%d.npz" % (cname, rank)), **{k.replace("/", "|"): v for k, v in finals[0].items()})
    stamp("all cases done")
    dist.destroy_process_group()
    stamp("process group destroyed")


# ---------------------------------------------------------------------------------------------------------------------
# DQN / BDQ handles in the exchange (north_star: "SAC / BDQ / DQN update ... optionally sharded"): two-shot + gather, the
# plan's own clip_by_norm + Adam behind it with grad_scale 1 / W (the sum clipped at W * clip: csrc/plan_q.inl)
def _q_ingraph_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["GRL_TUNE"] = "dp_timeout_ms=20000"
    _init_gloo(rank, world, out_dir)
    import q_parity_util as qu
    from grasp_rl.engine import GrlError
    from grasp_rl.parallel import DataParallelInGraph
    Bq = 16 * world
    for name, clip in (("bdq_baseline_config3_uniform", 0.005), ("dqn_reference_shape", 10.0)):
        case = qu.make_q_case(**dict(qu.CASES[name], B=Bq), n_replay=3 * Bq, n_steps=STEPS)
        cfg = _capi.GrlConfig.from_buffer_copy(case["cfg"])
        cfg.batch_size, cfg.q_grad_clip = Bq // world, clip
        case["cfg"] = cfg
        lo, hi = rank * (Bq // world), (rank + 1) * (Bq // world)
        # reference: every rank's bucket added in rank order on the host, applied with grad_scale 1 / W
        ref = qu.q_engine_setup(case)
        for s in range(STEPS):
            ref.compute_grads(case["idx"][s:s + 1, lo:hi], case["weights"][s:s + 1, lo:hi])
            g = torch.from_numpy(ref.fetch("grads", (ref.n_trainable,)))
            parts = [torch.empty_like(g) for _ in range(world)]
            dist.all_gather(parts, g)
            total = parts[0].numpy().copy()
            for p in parts[1:]:
                total += p.numpy()
            if s == 0 and clip < 1.0:         # the clip is active on the MEAN gradient of at least one variable
                n_vars = len(ref.param_names(trainable_only=True))    # (sum of squared variable norms > n c^2: one exceeds c)
                assert float(np.sqrt((total.astype(np.float64) ** 2).sum())) / world > clip * np.sqrt(n_vars)
            ref.store("grads", total)
            ref.apply_grads(1.0 / world)
        ref.synchronize()
        Pref = ref.get_parameters()
        ref.close()
        eng = qu.q_engine_setup(case)
        dp = DataParallelInGraph(eng, mode="auto")
        dp.train(STEPS, case["idx"][:, lo:hi], case["weights"][:, lo:hi])
        assert dp.check() == STEPS
        P = eng.get_parameters()
        for k in P:
            assert np.array_equal(P[k], Pref[k]), "%s: differs from the rank-ordered sum: %s" % (name, k)
        with pytest.raises(GrlError):
            eng.allreduce_set_mode("oneshot")          # (the clipped apply reads the gathered sums: two-shot only)
        dp.train(4)                                     # device RNG: same seed and replay contents on every rank
        dp.train(1)
        assert dp.check() == STEPS + 5
        np.savez(os.path.join(out_dir, "qig_%s_%d.npz" % (name, rank)), **{k.replace("/", "|"): v for k, v in eng.get_parameters().items()})
        dp.close()
        eng.close()
    # ---- prioritised replay on a connected handle: every rank draws from ITS priority tree (different uniforms per rank),
    # the exchange is the same; reference = the drawn rows and weights replayed on a plain handle, rank-ordered sum, apply
    from oracle.per import PerOracle
    case = qu.make_q_case(**dict(qu.CASES["bdq_baseline_config3"], B=Bq), n_replay=3 * Bq, n_steps=STEPS)
    cfg = _capi.GrlConfig.from_buffer_copy(case["cfg"])
    cfg.batch_size, cfg.q_grad_clip = Bq // world, 0.005
    plain = _capi.GrlConfig.from_buffer_copy(cfg)
    cfg.q_per, cfg.q_per_alpha, cfg.q_per_eps, cfg.q_per_alpha64 = 1, 0.6, 1e-6, 0.6
    eng = qu.q_engine_setup(dict(case, cfg=cfg))
    ref = qu.q_engine_setup(dict(case, cfg=plain))
    dp = DataParallelInGraph(eng, mode="auto")
    orc = PerOracle(int(cfg.replay_capacity), 0.6, 1e-6)
    orc.add(3 * Bq)
    rng = np.random.default_rng(50 + rank)
    for s in range(STEPS):
        u = rng.random(Bq // world)
        dp.train_per(1, 0.5, u[None])
        idx, w = eng.sampled_indices(), eng.importance_weights()
        ref_idx, ref_w = orc.sample(u, 0.5)               # this rank's tree, walked by the restated segment trees
        assert np.array_equal(idx, ref_idx)
        assert np.allclose(w, ref_w.astype(np.float32), rtol=1e-6, atol=0)
        orc.update(idx, eng.priorities())
        assert np.array_equal(eng.stored_priorities(), orc.leaves), "leaves after update %d" % s
        ref.compute_grads(idx[None], w[None])
        g = torch.from_numpy(ref.fetch("grads", (ref.n_trainable,)))
        parts = [torch.empty_like(g) for _ in range(world)]
        dist.all_gather(parts, g)
        total = parts[0].numpy().copy()
        for p in parts[1:]:
            total += p.numpy()
        ref.store("grads", total)
        ref.apply_grads(1.0 / world)
        ref.synchronize()
        P, Pr = eng.get_parameters(), ref.get_parameters()
        for k in P:
            assert np.array_equal(P[k], Pr[k]), "prioritised: differs from the rank-ordered sum at update %d: %s" % (s, k)
    with pytest.raises(GrlError):
        eng.train_allreduce(1)                             # (uniform draws are refused on a prioritised handle)
    dp.train_per(3, 0.7)                                   # device RNG: per-rank draws, same exchange
    assert dp.check() == STEPS + 3
    np.savez(os.path.join(out_dir, "qig_per_%d.npz" % rank), **{k.replace("/", "|"): v for k, v in eng.get_parameters().items()})
    dp.close()
    eng.close()
    ref.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_in_graph_exchange_on_dqn_and_bdq_handles(tmp_path, world):
    """configs[2]'s BDQ (uniform replay, as gripper_grasp.yaml:106 selects) and the reference-shape DQN with W processes on the
    one MI355X: gradient sums exchanged inside the update graph, clipped per variable as the mean of the replicas, bit-identical
    to the rank-ordered host sum + grl_apply_grads(1 / W); replicas identical also after device-RNG updates.  Then configs[2]
    WITH prioritised replay: every rank draws from its own tree (indices and float64 leaves equal to the restated segment
    trees), same exchange, same bits as the rank-ordered sum of the drawn rows' gradients."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_q_ingraph_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for name in ("bdq_baseline_config3_uniform", "dqn_reference_shape", "per"):
        parts = [np.load(os.path.join(str(tmp_path), "qig_%s_%d.npz" % (name, r))) for r in range(world)]
        for p in parts[1:]:
            for k in parts[0].files:
                assert np.array_equal(parts[0][k], p[k]), "replicas diverged (%s): %s" % (name, k)


def _q_learn_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      GRL_DP_SAME_DEVICE="1", GRL_TUNE="dp_timeout_ms=20000")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    from fake_env import FakeGraspEnv
    from grasp_rl.parallel import DataParallelInGraph
    from grasp_rl.sb.callbacks import BaseCallback
    from grasp_rl.sb.dqn import BDQ
    from grasp_rl.sb.policies import BdqMlpActPolicy
    from grasp_rl.sb.vec_env import DummyVecEnv

    class StopAt(BaseCallback):
        calls = 0

        def _on_step(self):
            StopAt.calls += 1
            return self.num_timesteps < 300

    env = DummyVecEnv([lambda: FakeGraspEnv(vector_dim=101, act_dim=5, seed=5 + rank)])
    model = BDQ(BdqMlpActPolicy, env, batch_size=64, buffer_size=512, learning_starts=80, target_network_update_freq=50, seed=3,
                num_actions_pad=33, prioritized_replay=True, policy_kwargs={"layers": [[64, 64], [32], [32]]}, data_parallel=True)
    assert isinstance(model._dp, DataParallelInGraph) and model.engine.cfg.batch_size == 32
    model.learn(10_000, callback=StopAt())
    P = model.get_parameters()
    leaves = model.engine.stored_priorities()
    np.savez(os.path.join(out_dir, "qlearn%d.npz" % rank), steps=model.num_timesteps, updates=model.n_updates, calls=StopAt.calls,
             replay=model.engine.replay_size(), **{k.replace("/", "|"): v for k, v in P.items()})
    np.save(os.path.join(out_dir, "qleaves%d.npy" % rank), leaves)
    model.engine.close()
    dist.destroy_process_group()


def test_bdq_learn_two_replicas_with_prioritised_replay_on_one_gpu(tmp_path):
    """configs[2] as the bench runs it -- BDQ, 101-d observations, 5 x 33 bins, global batch 64, prioritised replay -- through
    BDQ(data_parallel=True).learn with two replicas sharing the MI355X: exchange inside the update graph, ONE priority tree per
    rank (the ranks' leaves differ, their parameters do not), job-level counters, rank 0's callback stopping both loops."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_q_learn_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = np.load(os.path.join(str(tmp_path), "qlearn0.npz"))
    r1 = np.load(os.path.join(str(tmp_path), "qlearn1.npz"))
    assert int(r0["calls"]) > 0 and int(r1["calls"]) == 0
    assert int(r0["steps"]) == int(r1["steps"]) == 300 and int(r0["replay"]) == int(r1["replay"]) == 149
    assert int(r0["updates"]) == int(r1["updates"]) == 2 * (149 - 40)        # job steps 82 .. 298: two updates each
    for k in r0.files:
        if k != "calls":
            assert np.array_equal(r0[k], r1[k]), "replicas diverged: " + k
    l0, l1 = np.load(os.path.join(str(tmp_path), "qleaves0.npy")), np.load(os.path.join(str(tmp_path), "qleaves1.npy"))
    assert (l0[:149] > 0).all() and (l1[:149] > 0).all() and not np.array_equal(l0, l1)      # trees of their own


@pytest.mark.parametrize("world", [2, 4, 8])
def test_in_graph_exchange_processes_on_one_gpu(tmp_path, world, monkeypatch):
    """W processes on the box's one MI355X map each other's exchange buffers (hipIpc) and run the data-parallel update with
    the hand-written all-reduces inside the graph -- one-shot, two-shot, two-shot with the dense bucket overlapped: each
    bit-identical to the rank-ordered float32 sum formed on the host (for W = 2 also to the gloo exchange), the replicas
    bit-identical to each other (also after updates on the device RNG and a switch of the variant).  W = 8 adds the
    smallest bucket there is: ragged chunks, the last one empty -- and the CNN bucket LIGHT (two-shot over two explicit
    minibatches; with GRL_SLOW_TESTS=1 one-shot and two-shot over three: 127 s of eight time-sliced processes)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ["GRL_TEST_T0"] = repr(time.time())
    # (the workers build their cases with NumPy: orthogonal initialisation = an SVD of the 1024 x 512 dense kernels.  Eight
    # processes x one BLAS thread per core of a 256-core box spent 97 of the test's 115 s there: GRL_TEST_TIMING stamps)
    monkeypatch.setenv("OMP_NUM_THREADS", "4")
    monkeypatch.setenv("OPENBLAS_NUM_THREADS", "4")
    mp.spawn(_ingraph_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    if os.environ.get("GRL_TEST_TIMING") == "1":
        print("[W=%d parent] %6.1f s  spawn joined" % (world, time.time() - float(os.environ["GRL_TEST_T0"])), flush=True)
    for cname in ("cnn", "tiny") if world >= 8 else ("cnn",):
        parts = [np.load(os.path.join(str(tmp_path), "ig_%s_%d.npz" % (cname, r))) for r in range(world)]
        for p in parts[1:]:
            for k in parts[0].files:
                assert np.array_equal(parts[0][k], p[k]), "replicas diverged (%s): %s" % (cname, k)


def _timeout_worker(rank, world, port, out_dir, via="tune"):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["GRL_TUNE"] = "dp_timeout_ms=1500" if via == "tune" else "dp_timeout_ms=100000"
    _init_gloo(rank, world, out_dir)
    from grasp_rl._capi import GrlError
    from grasp_rl.parallel import DataParallelInGraph
    case = _case()
    cfg = _capi.GrlConfig.from_buffer_copy(case["cfg"])
    cfg.batch_size = B // world
    case["cfg"] = cfg
    eng = pu.engine_setup(case)
    dp = DataParallelInGraph(eng, mode="twoshot")
    dp.train(2)
    assert dp.check() == 2
    before = eng.get_parameters()
    if via == "call":                 # grl_allreduce_set_timeout: the host's bound replaces the captured 100 s, graphs untouched
        eng.allreduce_set_timeout(1500)
    dist.barrier()
    raised = False
    t_wait = time.time()
    if rank == 0:
        dp.train(1)                   # the peer never arrives: the wait runs out, the channel is poisoned on EVERY rank
        try:
            dp.check()
        except GrlError:
            raised = True
        assert 1.0 < time.time() - t_wait < 30.0      # (gave up after the 1.5 s bound, not the 100 s one)
        dist.barrier()
    else:
        dist.barrier()                # (rank 0 has given up by now)
        try:
            dp.train(1)               # refused: the host mailbox / the poisoned flags say the replicas are out of step
            dp.check()
        except GrlError:
            raised = True
    assert raised, "rank %d did not see the time-out" % rank
    try:
        dp.train(1)
        later = False
    except GrlError:
        later = True
    assert later, "a poisoned channel must refuse further updates"
    after = eng.get_parameters()
    for k in before:                  # nobody applied a partial exchange
        assert np.array_equal(before[k], after[k]), k
    dp.close()
    eng.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("via", ["tune", "call"])
def test_a_missing_peer_poisons_the_exchange_on_every_rank(tmp_path, via):
    """Bounded waits (GRL_TUNE dp_timeout_ms at plan time, or grl_allreduce_set_timeout at any time -- bench.py verifies a fresh
    exchange under a 5 s bound and trains under the long one): a rank whose peer does not arrive gives up, raises `error` on every rank and in
    its host mailbox, and announces nothing further -- both ranks' next calls fail, no replica has applied a stale sum."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_timeout_worker, args=(2, port, str(tmp_path), via), nprocs=2, join=True)


# ---------------------------------------------------------------------------------------------------------------------
# data parallelism through the drop-in API: SAC(data_parallel=True).learn (sb_helper.py:175-177), two replicas on one GPU
def _learn_worker(rank, world, port, out_dir, device_norm):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      GRL_DP_SAME_DEVICE="1", GRL_TUNE="dp_timeout_ms=20000")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    from fake_env import FakeGraspEnv
    from grasp_rl.engine import SacEngine
    from grasp_rl.parallel import DataParallelInGraph
    from grasp_rl.sb import policies as pol
    from grasp_rl.sb.sac import SAC
    from grasp_rl.sb.vec_env import DummyVecEnv, VecNormalize
    from test_data_parallel_gloo import _augmented

    env = VecNormalize(DummyVecEnv([(lambda i=i: FakeGraspEnv("depth", seed=10 * rank + i)) for i in range(2)]),
                       training=True, norm_obs=True, norm_reward=True, clip_obs=10.0)
    model = SAC(pol.SacCnnPolicy, env, batch_size=16, buffer_size=64, learning_starts=8, seed=3, data_parallel=True,
                device_norm=device_norm, policy_kwargs={"cnn_extractor": _augmented(1)})
    assert isinstance(model._dp, DataParallelInGraph) and model.engine.cfg.batch_size == 8
    eng = model.engine
    P0 = eng.get_parameters()
    # record what learn() hands to the engine / the exchange: the same schedule is replayed below through the wrapper alone
    log = []
    for name in ("replay_add", "set_obs_stats", "set_ret_var", "norm_update", "set_running_stats", "observe", "replay_add_observed"):
        def rec(*a, _f=getattr(eng, name), _n=name, **kw):
            log.append((_n, tuple(np.array(x, copy=True) if isinstance(x, np.ndarray) else x for x in a), dict(kw)))
            return _f(*a, **kw)
        setattr(eng, name, rec)
    train = model._dp.train
    model._dp.train = lambda k: (log.append(("train", (k,), {})), train(k))[1]
    model.learn(48)
    assert model.num_timesteps == 48 and model.n_updates > 0
    P = model.get_parameters()
    mean, var, count = env.obs_rms.mean.copy(), env.obs_rms.var.copy(), env.obs_rms.count
    assert count == pytest.approx(1e-4 + 2 * 2 * 13)           # reset + 12 steps, both ranks' batches merged (host or device)
    # ---- the bench-style path: an engine of the same configuration driven by the wrapper alone
    eng2 = SacEngine(eng.cfg, device="cuda:0")
    eng2.set_parameters(P0)
    dp2 = DataParallelInGraph(eng2, group=model._dp_rt.ctrl)
    for name, a, kw in log:
        if name == "train":
            dp2.train(*a)
        else:
            getattr(eng2, name)(*a, **kw)
    if device_norm:
        assert any(name == "replay_add_observed" for name, _, _ in log)      # one upload per env step behind model.learn
    dp2.check()
    P2 = eng2.get_parameters()
    for k in P:
        assert np.array_equal(P[k], P2[k]), "model.learn and the wrapper-driven engine differ: " + k
    np.savez(os.path.join(out_dir, "learn%d.npz" % rank), obs_mean=mean, obs_var=var, obs_count=count, ret_var=env.ret_rms.var,
             **{k.replace("/", "|"): v for k, v in P.items()})
    dp2.close()
    model._dp.close()
    eng2.close()
    model.engine.close()
    dist.destroy_process_group()


def test_model_learn_two_replicas_on_one_gpu(tmp_path):
    """SAC(data_parallel=True).learn, two processes: the in-graph exchange behind `model.learn`, running statistics merged
    over the ranks on the host (share_running_stats) or on the device (grl_norm_update on a connected handle).  Each
    replica reaches exactly the parameters of an engine driven by the DataParallelInGraph wrapper alone on the recorded
    schedule (what bench.py does), and the replicas -- parameters and statistics -- are bit-identical.  The statistics the
    two paths end with are the same bits: both restate RunningMeanStd over the gathered batches."""
    stats = {}
    for device_norm in (False, True):
        out = tmp_path / ("dev" if device_norm else "host")
        out.mkdir()
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        mp.spawn(_learn_worker, args=(2, port, str(out), device_norm), nprocs=2, join=True)
        r0 = np.load(os.path.join(str(out), "learn0.npz"))
        r1 = np.load(os.path.join(str(out), "learn1.npz"))
        for k in r0.files:
            assert np.array_equal(r0[k], r1[k]), "replicas diverged (device_norm=%s): %s" % (device_norm, k)
        stats[device_norm] = (r0["obs_mean"], r0["obs_var"], float(r0["obs_count"]))
    assert np.array_equal(stats[False][0], stats[True][0]) and np.array_equal(stats[False][1], stats[True][1])
    assert stats[False][2] == stats[True][2]


# ---------------------------------------------------------------------------------------------------------------------
# the shape of `python -m grasp_rl.dp_run train_stable_baselines.py ...` with GRL_NUM_ENVS: W = 4 replicas, the job's 8
# environments = 2 worker processes per rank, each rank's env built as the reference's script builds it (ONE factory)
def _fanout_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      GRL_DP_SAME_DEVICE="1", GRL_TUNE="dp_timeout_ms=60000", GRL_NUM_ENVS="8", GRL_DATA_PARALLEL="auto")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import functools
    from fake_env import FakeGraspEnv
    from grasp_rl.parallel import DataParallelInGraph
    from grasp_rl.sb import policies as pol
    from grasp_rl.sb.monitor import Monitor
    from grasp_rl.sb.sac import SAC
    from grasp_rl.sb.vec_env import DummyVecEnv, VecNormalize
    from test_data_parallel_gloo import _augmented
    log_dir = os.path.join(out_dir, "rank%d" % rank)
    os.makedirs(log_dir)
    env = DummyVecEnv([functools.partial(_monitored_env, os.path.join(log_dir, "log_file"))])     # train_stable_baselines.py:54
    assert env.num_envs == 1 and env.envs[0].depth_obs                                            # sb_helper.py:86
    env = VecNormalize(env, norm_obs=True, norm_reward=True, clip_obs=10.0)                       # sb_helper.py:117-119
    model = SAC(pol.SacCnnPolicy, env, batch_size=16, buffer_size=128, learning_starts=16, seed=5,
                policy_kwargs={"cnn_extractor": _augmented(1)})                                   # data_parallel from the environment
    assert isinstance(model._dp, DataParallelInGraph) and model.engine.cfg.batch_size == 4
    assert env.num_envs == model.n_envs == model.engine.cfg.act_batch == 2 and env.ret.shape == (2,)
    model.learn(96)
    assert model.num_timesteps == 96                               # 12 vectorised steps x 2 envs x 4 ranks
    assert model.n_updates == 8 * (12 - 1)                         # one update per environment step of the job once 16 steps exist
    files = sorted(f for f in os.listdir(log_dir) if f.endswith("monitor.csv"))
    first = 2 * rank                                               # environment numbers are the job's: rank r owns 2r, 2r + 1
    assert files == sorted(["log_file.env%d.monitor.csv" % k if k else "log_file.monitor.csv" for k in (first, first + 1)]), files
    P = model.get_parameters()
    np.savez(os.path.join(out_dir, "fan%d.npz" % rank), obs_mean=env.obs_rms.mean, obs_count=env.obs_rms.count,
             **{k.replace("/", "|"): v for k, v in P.items()})
    env.close()
    model.engine.close()
    dist.destroy_process_group()


def _monitored_env(path):
    from fake_env import FakeGraspEnv
    from grasp_rl.sb.monitor import Monitor
    return Monitor(FakeGraspEnv("depth", seed=None, episode_len=4), path)


def test_model_learn_four_replicas_with_fanned_out_envs(tmp_path):
    """Row J3 under data parallelism (`dp_run` + GRL_NUM_ENVS, BASELINE configs[4] in small): four replicas on one MI355X,
    each fanning the single env it was handed out to 2 worker processes, global minibatch 16 = 4 rows per rank, gradients
    exchanged inside the update graph (two-shot at W = 4), one update per environment step of the job; replicas identical."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_fanout_worker, args=(4, port, str(tmp_path)), nprocs=4, join=True)
    parts = [np.load(os.path.join(str(tmp_path), "fan%d.npz" % r)) for r in range(4)]
    for p in parts[1:]:
        for k in parts[0].files:
            assert np.array_equal(parts[0][k], p[k]), "replicas diverged: " + k
    assert float(parts[0]["obs_count"]) == pytest.approx(1e-4 + 8 * 13)      # reset + 12 steps of all 8 environments


# ---------------------------------------------------------------------------------------------------------------------
def _setup_failure_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      GRL_DP_SAME_DEVICE="1", GRL_TUNE="dp_timeout_ms=20000")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    from grasp_rl.parallel import DataParallelInGraph, DataParallelRuntime, DataParallelSac, ExchangeSetupError
    rt = DataParallelRuntime()
    case = _case()
    cfg = _capi.GrlConfig.from_buffer_copy(case["cfg"])
    cfg.batch_size = B // world
    case["cfg"] = cfg
    lo, hi = rank * (B // world), (rank + 1) * (B // world)
    eng = pu.engine_setup(case)
    real = eng.allreduce_connect
    if rank == 1:                     # the peers' memory maps, then the set-up fails on this rank only
        def broken(handles):
            real(handles)
            raise RuntimeError("injected failure after hipIpcOpenMemHandle")
        eng.allreduce_connect = broken
    with pytest.raises(ExchangeSetupError):
        DataParallelInGraph(eng, group=rt.ctrl)
    # every rank released its mappings and its exchange memory; the handle trains alone again ...
    eng.train(1, case["idx"][:1, lo:hi], case["eps"][:1, lo:hi])
    eng.synchronize()
    # ... can be connected anew (the real set-up this time) ...
    eng.allreduce_connect = real
    dp = DataParallelInGraph(eng, group=rt.ctrl, mode="twoshot")
    dp.train(2)
    assert dp.check() == 2
    dp.close(disconnect=True)
    # ... and make_exchange lands every rank on the collective fallback together when the failure persists
    if rank == 1:
        eng.allreduce_connect = broken
    fb = rt.make_exchange(eng, prefer="ingraph")
    assert isinstance(fb, DataParallelSac) and isinstance(rt.ingraph_error, ExchangeSetupError)
    fb.broadcast_parameters(src=0)
    eng.reset_optimizer()             # (the solo update above left every rank its own Adam moments)
    fb.train(STEPS, case["idx"][:, lo:hi], case["eps"][:, lo:hi])
    eng.synchronize()
    P = eng.get_parameters()
    np.savez(os.path.join(out_dir, "sf%d.npz" % rank), **{k.replace("/", "|"): v for k, v in P.items()})
    eng.close()
    dist.destroy_process_group()


def test_exchange_setup_failure_releases_every_rank_and_falls_back_together(tmp_path):
    """a rank whose set-up fails AFTER the collectives of the set-up began (here: after mapping its peers)
    no longer leaves the others in `all_gather_object` / `barrier` -- every phase is voted on, every rank disconnects
    (grl_allreduce_disconnect), the handle works alone, can be connected again, and `make_exchange` falls back on all ranks."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_setup_failure_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = np.load(os.path.join(str(tmp_path), "sf0.npz"))
    r1 = np.load(os.path.join(str(tmp_path), "sf1.npz"))
    for k in r0.files:
        assert np.array_equal(r0[k], r1[k]), "replicas diverged: " + k
