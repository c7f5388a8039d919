"""Data-parallel path on CPU: 2 processes over gloo, each with its own engine (TEST-ONLY g++ emulation
build) holding half of the minibatch, one all-reduce of the flat gradient bucket per update
(grasp_rl.parallel.DataParallelSac).  Must equal one engine updating on the whole minibatch
(every SAC loss is a batch mean, so the mean of shard gradients is the global gradient)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import parity_util as pu
from grasp_rl import _capi
from grasp_rl.parallel import DataParallelSac
from hostemu_backend import NumpyHostBackend

B, STEPS = 8, 2


def _case():
    return pu.make_case(extractor="augmented", kind="depth", B=B, n_replay=32, n_steps=STEPS)


def _shard_cfg(cfg, batch):
    c = _capi.GrlConfig.from_buffer_copy(cfg)
    c.batch_size = batch
    return c


def _worker(rank, world, port, lib, out_dir):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    case = _case()
    case["cfg"] = _shard_cfg(case["cfg"], B // world)
    eng = pu.engine_setup(case, backend=NumpyHostBackend(), lib_path=lib)
    dp = DataParallelSac(eng, overlap=True)
    assert dp.staged and dp.overlap                # two-bucket schedule: dense ranges after stage 0, conv after stage 1
    r0, r1 = eng.grad_ranges(0), eng.grad_ranges(1)
    covered = sorted(r0 + r1)
    assert covered[0][0] == 0 and all(a[0] + a[1] == b[0] for a, b in zip(covered, covered[1:]))   # a partition ...
    assert covered[-1][0] + covered[-1][1] == eng.n_trainable                                      # ... of the bucket
    if rank != 0:                                  # replicas must start identical: perturb, then broadcast
        P = eng.get_parameters()
        P["model/pi/fc0/bias:0"] = P["model/pi/fc0/bias:0"] + 1.0
        eng.set_parameters(P)
    dp.broadcast_parameters(src=0)
    lo, hi = rank * (B // world), (rank + 1) * (B // world)
    dp.train(STEPS, case["idx"][:, lo:hi], case["eps"][:, lo:hi])
    P = eng.get_parameters()
    # the single-bucket schedule (one all-reduce after the whole gradient computation) gives the same bits
    eng1 = pu.engine_setup(case, backend=NumpyHostBackend(), lib_path=lib)
    dp1 = DataParallelSac(eng1, overlap=False)
    dp1.train(STEPS, case["idx"][:, lo:hi], case["eps"][:, lo:hi])
    P1 = eng1.get_parameters()
    for k in P:
        assert np.array_equal(P[k], P1[k]), "staged and single-bucket schedules differ: " + k
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), **{k.replace("/", "|"): v for k, v in P.items()})
    dist.destroy_process_group()


def test_two_rank_update_equals_single_engine(hostemu_lib, tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, hostemu_lib, str(tmp_path)), nprocs=2, join=True)
    case = _case()
    single = pu.engine_setup(case, backend=NumpyHostBackend(), lib_path=hostemu_lib)
    single.train(STEPS, case["idx"], case["eps"])
    ref = single.get_parameters()
    r0 = np.load(os.path.join(str(tmp_path), "rank0.npz"))
    r1 = np.load(os.path.join(str(tmp_path), "rank1.npz"))
    lr = case["spec"].lr
    for k, v in ref.items():
        a, b = r0[k.replace("/", "|")], r1[k.replace("/", "|")]
        assert np.array_equal(a, b), "replicas diverged: " + k
        d = np.abs(a.astype(np.float64) - v)
        assert d.max() <= 0.3 * lr * STEPS + 1e-7 and d.mean() <= 0.02 * lr * STEPS + 1e-9, (k, d.max(), d.mean())


def _stats_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from grasp_rl.parallel import share_running_stats
    from grasp_rl.sb.running_mean_std import RunningMeanStd

    class _VN:                                     # the two attributes share_running_stats touches
        obs_rms, ret_rms = RunningMeanStd(shape=(4, 3)), RunningMeanStd(shape=())
    vn = share_running_stats(_VN())
    rng = np.random.default_rng(100 + rank)
    for _ in range(3):
        vn.obs_rms.update(rng.normal(rank, 1.0 + rank, (5 + rank, 4, 3)))
        vn.ret_rms.update(rng.normal(size=5 + rank))
    import pickle
    assert "gather" not in pickle.loads(pickle.dumps(vn.obs_rms)).__dict__
    np.savez(os.path.join(out_dir, "stats%d.npz" % rank), mean=vn.obs_rms.mean, var=vn.obs_rms.var,
             count=vn.obs_rms.count, rmean=vn.ret_rms.mean, rvar=vn.ret_rms.var)
    dist.destroy_process_group()


def test_running_stats_are_merged_over_ranks(tmp_path):
    """VecNormalize moments under data parallelism: replicas stay identical and equal the rank-ordered merge."""
    from grasp_rl.sb.running_mean_std import RunningMeanStd
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_stats_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = np.load(os.path.join(str(tmp_path), "stats0.npz"))
    r1 = np.load(os.path.join(str(tmp_path), "stats1.npz"))
    for k in r0.files:
        assert np.array_equal(r0[k], r1[k]), k
    ref, rref = RunningMeanStd(shape=(4, 3)), RunningMeanStd(shape=())
    rngs = [np.random.default_rng(100), np.random.default_rng(101)]
    for _ in range(3):
        batches = [(rngs[r].normal(r, 1.0 + r, (5 + r, 4, 3)), rngs[r].normal(size=5 + r)) for r in range(2)]
        for o, ret in batches:
            ref.update(o)
            rref.update(ret)
    assert np.array_equal(ref.mean, r0["mean"]) and np.array_equal(ref.var, r0["var"]) and ref.count == float(r0["count"])
    assert np.array_equal(rref.var, r0["rvar"])
    # and the merge is the statistics of the union of all batches
    rngs = [np.random.default_rng(100), np.random.default_rng(101)]
    allobs = []
    for _ in range(3):
        for r in range(2):
            allobs.append(rngs[r].normal(r, 1.0 + r, (5 + r, 4, 3)))
            rngs[r].normal(size=5 + r)
    allobs = np.concatenate(allobs)
    assert np.allclose(r0["mean"], allobs.mean(0), atol=1e-4) and np.allclose(r0["var"], allobs.var(0), rtol=1e-3, atol=1e-4)


def test_runtime_for_modes_without_a_launch(monkeypatch):
    """`data_parallel=` of the models: off values and "auto" outside a torchrun launch give no runtime (single-process training);
    an insisting value without a launch raises instead of training a lone replica silently."""
    from grasp_rl.parallel import runtime_for
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    for mode in (None, False, "", "0", "off", "auto"):
        assert runtime_for(mode) is None
    with pytest.raises(RuntimeError):
        runtime_for(True)


def test_scale_and_bucket_helpers():
    from grasp_rl.parallel import allreduce_mean_scale
    assert allreduce_mean_scale(8) == 0.125


# ---------------------------------------------------------------------------------------------------------------------
# data parallelism through the drop-in API: SAC(data_parallel=...).learn (sb_helper.py:175-177 on an env built like
# train_stable_baselines.py:52-54), two replicas over gloo with the emulation engine
def _learn_worker(rank, world, port, lib, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from fake_env import FakeGraspEnv
    from grasp_rl.engine import SacEngine
    from grasp_rl.sb.callbacks import BaseCallback
    from grasp_rl.sb.sac import SAC
    from grasp_rl.sb.vec_env import DummyVecEnv, VecNormalize
    from grasp_rl.sb import policies as pol
    SAC._engine_factory = staticmethod(lambda cfg, device: SacEngine(cfg, backend=NumpyHostBackend(), lib_path=lib))

    class StopAt(BaseCallback):
        calls = 0

        def _on_step(self):
            StopAt.calls += 1
            return self.num_timesteps < 40          # rank 0 decides; every replica must leave the loop in the same iteration

    env = VecNormalize(DummyVecEnv([(lambda i=i: FakeGraspEnv("depth", seed=10 * rank + i)) for i in range(2)]),
                       training=True, norm_obs=True, norm_reward=True, clip_obs=10.0)
    model = SAC(pol.SacCnnPolicy, env, batch_size=8, buffer_size=64, learning_starts=8, seed=3, data_parallel=True,
                dp_exchange="collective", policy_kwargs={"cnn_extractor": _augmented(1)})
    assert model.engine.cfg.batch_size == 4                      # the global minibatch dealt over the ranks
    cb = StopAt()
    model.learn(10_000, callback=cb)
    P = model.get_parameters()
    np.savez(os.path.join(out_dir, "learn%d.npz" % rank), steps=model.num_timesteps, updates=model.n_updates, calls=StopAt.calls,
             obs_mean=env.obs_rms.mean, obs_var=env.obs_rms.var, obs_count=env.obs_rms.count, ret_var=env.ret_rms.var,
             replay=model.engine.replay_size(), **{k.replace("/", "|"): v for k, v in P.items()})
    dist.destroy_process_group()


def _augmented(n):
    def augmented_nature_cnn(scaled_images, **kwargs):
        raise AssertionError("the TF extractor must not be called")
    def bound(scaled_images, **kwargs):
        return augmented_nature_cnn(scaled_images, n=n, **kwargs)
    bound.__name__ = "augmented_nature_cnn"
    return bound


def test_model_learn_as_one_replica_of_two(hostemu_lib, tmp_path):
    """SAC(data_parallel=True).learn with two replicas: global minibatch split over the ranks, parameters from rank 0,
    gradients exchanged every update, VecNormalize statistics (observations and returns) merged over the ranks at every
    env step, timesteps counted over all ranks, callbacks on rank 0 only and its stop request ending both loops in the
    same iteration -- the replicas end bit-identical."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_learn_worker, args=(2, port, hostemu_lib, str(tmp_path)), nprocs=2, join=True)
    r0 = np.load(os.path.join(str(tmp_path), "learn0.npz"))
    r1 = np.load(os.path.join(str(tmp_path), "learn1.npz"))
    assert int(r0["calls"]) > 0 and int(r1["calls"]) == 0
    assert int(r0["steps"]) == int(r1["steps"]) == 40                  # 2 ranks x 2 envs per vectorised step
    assert int(r0["replay"]) == int(r1["replay"]) == 18                # (the stopping step is not stored)
    # gradient_steps=None = one update per ENVIRONMENT step of the job: 2 ranks x 2 envs = 4 updates (each on the global
    # minibatch) per vectorised step, from the step that reaches learning_starts = 8 to the one before the stop
    assert int(r0["updates"]) == int(r1["updates"]) == 4 * 8
    for k in r0.files:
        if k != "calls":
            assert np.array_equal(r0[k], r1[k]), "replicas diverged: " + k
    assert float(r0["obs_count"]) == pytest.approx(1e-4 + 2 * 2 * 11)   # reset + 10 steps, both ranks' batches merged


def _device_norm_vote_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from grasp_rl.parallel import DataParallelRuntime
    from grasp_rl.sb.callbacks import CheckpointCallback
    from grasp_rl.sb.sac import resolve_device_norm
    rt = DataParallelRuntime()

    class InGraph:                      # what SAC sees of a connected in-graph exchange
        def check(self):
            pass

    def fn_callback(_locals, _globals):     # a plain function: wrapped in ConvertCallback, may read locals['new_obs']
        return True
    got = {
        # the user's callback is judged BEFORE the rank filter: the same function callback on both ranks -> host statistics
        "fn_both": resolve_device_norm("auto", fn_callback, rt, InGraph()),
        # a script that hands only rank 0 a callback: rank 1 alone would choose the device -- the vote says host on both
        "fn_rank0_only": resolve_device_norm("auto", fn_callback if rank == 0 else None, rt, InGraph()),
        # nothing reads the observations anywhere: the device on both
        "none": resolve_device_norm("auto", None, rt, InGraph()),
        "known": resolve_device_norm("auto", CheckpointCallback(10, "/tmp/x") if rank == 0 else None, rt, InGraph()),
        # explicit settings that disagree across ranks still end the same on both
        "explicit_mixed": resolve_device_norm(rank == 0, None, rt, InGraph()),
        # the collective fallback has no channel for the device-side merge
        "fallback": resolve_device_norm(True, None, rt, object()),
    }
    np.savez(os.path.join(out_dir, "vote%d.npz" % rank), **{k: np.array(v) for k, v in got.items()})
    dist.destroy_process_group()


def test_device_norm_auto_is_decided_collectively(tmp_path):
    """`device_norm='auto'` was resolved per rank after the rank filter on the callback, so rank 0 with a
    function callback took the host-statistics path (an all_gather per env step) while ranks > 0 took the device path (an
    in-kernel merge that waits for every peer): two different exchanges waiting for each other."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_device_norm_vote_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = np.load(os.path.join(str(tmp_path), "vote0.npz"))
    r1 = np.load(os.path.join(str(tmp_path), "vote1.npz"))
    want = {"fn_both": False, "fn_rank0_only": False, "none": True, "known": True, "explicit_mixed": False, "fallback": False}
    for k, v in want.items():
        assert bool(r0[k]) == bool(r1[k]) == v, (k, bool(r0[k]), bool(r1[k]))


# ---------------------------------------------------------------------------------------------------------------------
# set-up of the in-graph exchange is collective and fail-safe: whatever fails on whichever rank, every rank executes the
# same collectives, releases what it had set up and takes the fallback together (ADVICE r4 / VERDICT r4 "Next 1b")
def _fallback_worker(rank, world, port, lib, out_dir, fail_phase, fail_rank):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from grasp_rl.parallel import DataParallelInGraph, DataParallelRuntime, ExchangeSetupError
    rt = DataParallelRuntime()                     # initialises gloo from the launcher variables
    case = _case()
    case["cfg"] = _shard_cfg(case["cfg"], B // world)
    eng = pu.engine_setup(case, backend=NumpyHostBackend(), lib_path=lib)
    calls = []
    if rank == fail_rank:
        real = getattr(eng, fail_phase)

        def broken(*a, **k):
            calls.append(fail_phase)
            if fail_phase == "allreduce_connect":
                real(*a, **k)                      # the mapping itself worked; what follows it does not
            raise RuntimeError("injected: %s fails on rank %d" % (fail_phase, rank))
        setattr(eng, fail_phase, broken)
    # the constructor raises on EVERY rank, from the same place
    with pytest.raises(ExchangeSetupError) as ei:
        DataParallelInGraph(eng, group=rt.ctrl)
    assert ("this rank" in str(ei.value)) == (rank == fail_rank)
    # ... and left a plain handle behind on every rank: a fresh initialisation is accepted, device statistics do not wait for peers
    if rank == fail_rank:
        setattr(eng, fail_phase, real)
    eng.allreduce_init(rank, world)
    eng.allreduce_disconnect()
    eng.allreduce_disconnect()                     # idempotent
    # make_exchange: everyone lands on the collective fallback and trains in step
    if rank == fail_rank:
        setattr(eng, fail_phase, broken)
    dp = rt.make_exchange(eng, prefer="ingraph")
    assert isinstance(dp, DataParallelSac) and isinstance(rt.ingraph_error, ExchangeSetupError)
    dp.broadcast_parameters(src=0)
    lo, hi = rank * (B // world), (rank + 1) * (B // world)
    dp.train(STEPS, case["idx"][:, lo:hi], case["eps"][:, lo:hi])
    P = eng.get_parameters()
    np.savez(os.path.join(out_dir, "fb%d.npz" % rank), **{k.replace("/", "|"): v for k, v in P.items()})
    dist.destroy_process_group()


@pytest.mark.parametrize("fail_phase,fail_rank", [("allreduce_init", 1), ("allreduce_connect", 0)])
def test_exchange_setup_failure_on_one_rank_falls_back_on_all(hostemu_lib, tmp_path, fail_phase, fail_rank):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_fallback_worker, args=(2, port, hostemu_lib, str(tmp_path), fail_phase, fail_rank), nprocs=2, join=True)
    r0 = np.load(os.path.join(str(tmp_path), "fb0.npz"))
    r1 = np.load(os.path.join(str(tmp_path), "fb1.npz"))
    for k in r0.files:
        assert np.array_equal(r0[k], r1[k]), "replicas diverged: " + k


# ---------------------------------------------------------------------------------------------------------------------
# DQN / BDQ handles under data parallelism (north_star: "SAC / BDQ / DQN update ... optionally sharded").  The TD loss is a batch
# mean, so the mean of the shard gradients is the global gradient; clip_by_norm acts on that mean -- the all-reduced SUM is
# clipped at W * clip and Adam's 1 / W brings it back (csrc/plan_q.inl) -- with a threshold the gradients really exceed here.
Q_B, Q_STEPS, Q_CLIP = 8, 3, 0.05


def _q_case(batch, clip=Q_CLIP):
    import q_parity_util as qu
    case = qu.make_q_case(**dict(qu.CASES["bdq_5_branches"], B=Q_B), n_steps=Q_STEPS, lr=1e-2)
    cfg = _capi.GrlConfig.from_buffer_copy(case["cfg"])
    cfg.batch_size, cfg.q_grad_clip = batch, clip
    case["cfg"] = cfg
    return case


def _q_worker(rank, world, port, lib, out_dir):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    import q_parity_util as qu
    case = _q_case(Q_B // world)
    eng = qu.q_engine_setup(case, backend=NumpyHostBackend(), lib_path=lib)
    dp = DataParallelSac(eng)
    assert not dp.staged
    dp.broadcast_parameters(src=0)
    lo, hi = rank * (Q_B // world), (rank + 1) * (Q_B // world)
    dp.train(Q_STEPS, case["idx"][:, lo:hi], case["weights"][:, lo:hi])
    P = eng.get_parameters()
    np.savez(os.path.join(out_dir, "q%d.npz" % rank), **{k.replace("/", "|"): v for k, v in P.items()})
    dist.destroy_process_group()


def test_two_rank_bdq_update_with_clipping_equals_single_engine(hostemu_lib, tmp_path):
    import q_parity_util as qu
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_q_worker, args=(2, port, hostemu_lib, str(tmp_path)), nprocs=2, join=True)
    case = _q_case(Q_B)
    single = qu.q_engine_setup(case, backend=NumpyHostBackend(), lib_path=hostemu_lib)
    single.compute_grads(case["idx"][:1], case["weights"][:1])
    G = single.get_gradients()
    assert max(float(np.sqrt((g.astype(np.float64) ** 2).sum())) for g in G.values()) > 2 * Q_CLIP     # the clip is active
    single.apply_grads(1.0)
    single.train(Q_STEPS - 1, case["idx"][1:], case["weights"][1:])
    ref = single.get_parameters()
    # ... and it matters: without it the parameters move visibly further
    unclipped = qu.q_engine_setup(_q_case(Q_B, clip=0.0), backend=NumpyHostBackend(), lib_path=hostemu_lib)
    unclipped.train(Q_STEPS, case["idx"], case["weights"])
    Pu = unclipped.get_parameters()
    r0 = np.load(os.path.join(str(tmp_path), "q0.npz"))
    r1 = np.load(os.path.join(str(tmp_path), "q1.npz"))
    lr = 1e-2
    moved = 0
    for k, v in ref.items():
        a, b = r0[k.replace("/", "|")], r1[k.replace("/", "|")]
        assert np.array_equal(a, b), "replicas diverged: " + k
        d = np.abs(a.astype(np.float64) - v)
        assert d.max() <= 0.3 * lr * Q_STEPS + 1e-7 and d.mean() <= 0.02 * lr * Q_STEPS + 1e-9, (k, d.max(), d.mean())
        moved = max(moved, float(np.abs(Pu[k].astype(np.float64) - v).max()))
    assert moved > 0


def _q_learn_worker(rank, world, port, lib, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from fake_env import FakeGraspEnv
    from grasp_rl.engine import QEngine
    from grasp_rl.sb.callbacks import BaseCallback
    from grasp_rl.sb.dqn import BDQ
    from grasp_rl.sb.vec_env import DummyVecEnv
    BDQ._engine_factory = staticmethod(lambda cfg, device: QEngine(cfg, backend=NumpyHostBackend(), lib_path=lib))

    class StopAt(BaseCallback):
        calls = 0

        def _on_step(self):
            StopAt.calls += 1
            return self.num_timesteps < 60

    from grasp_rl.sb.policies import BdqMlpActPolicy as MlpActPolicy
    env = DummyVecEnv([lambda: FakeGraspEnv(vector_dim=12, act_dim=3, seed=5 + rank)])
    model = BDQ(MlpActPolicy, env, batch_size=8, buffer_size=64, learning_starts=16, target_network_update_freq=20, seed=3, num_actions_pad=5,
                policy_kwargs={"layers": [[16, 16], [8], [8]]}, data_parallel=True, dp_exchange="collective")
    assert model.engine.cfg.batch_size == 4
    model.learn(10_000, callback=StopAt())
    P = model.get_parameters()
    np.savez(os.path.join(out_dir, "ql%d.npz" % rank), steps=model.num_timesteps, updates=model.n_updates, calls=StopAt.calls,
             replay=model.engine.replay_size(), **{k.replace("/", "|"): v for k, v in P.items()})
    dist.destroy_process_group()


def test_bdq_learn_as_one_replica_of_two(hostemu_lib, tmp_path):
    """BDQ(data_parallel=True).learn with two replicas over gloo: every rank steps its own environment, counters count the
    job's environment steps (2 per iteration), one update on the global minibatch per step of the job, hard target copies in
    step, rank 0's callback stops both loops in the same iteration; the replicas end bit-identical, targets included."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_q_learn_worker, args=(2, port, hostemu_lib, str(tmp_path)), nprocs=2, join=True)
    r0 = np.load(os.path.join(str(tmp_path), "ql0.npz"))
    r1 = np.load(os.path.join(str(tmp_path), "ql1.npz"))
    assert int(r0["calls"]) > 0 and int(r1["calls"]) == 0
    assert int(r0["steps"]) == int(r1["steps"]) == 60
    assert int(r0["replay"]) == int(r1["replay"]) == 29                 # (the stopping step is not stored)
    # steps 18, 20, ..., 58 of the job (> learning_starts, before the stop): two updates each
    assert int(r0["updates"]) == int(r1["updates"]) == 2 * 21
    for k in r0.files:
        if k != "calls":
            assert np.array_equal(r0[k], r1[k]), "replicas diverged: " + k
    moved = [k for k in r0.files if "target_q_func" in k and "weights" in k]
    assert moved
