"""ctypes binding of libgrl.so (include/grl.h).  No torch / numpy dependency here."""
import ctypes as C
import os

GRL_MAX_LAYERS = 4
EXTRACTOR_MLP, EXTRACTOR_AUGMENTED, EXTRACTOR_NATURE = 0, 1, 2

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(_HERE, "libgrl.so")


class GrlConfig(C.Structure):
    _fields_ = [
        ("extractor", C.c_int32), ("img_hw", C.c_int32), ("obs_channels", C.c_int32),
        ("n_direct", C.c_int32), ("obs_dim", C.c_int32), ("act_dim", C.c_int32),
        ("n_layers", C.c_int32), ("layers", C.c_int32 * GRL_MAX_LAYERS),
        ("batch_size", C.c_int32), ("act_batch", C.c_int32), ("replay_capacity", C.c_int64),
        ("normalize", C.c_int32), ("gamma", C.c_float), ("lr", C.c_float), ("tau", C.c_float),
        ("clip_obs", C.c_float), ("clip_reward", C.c_float), ("norm_eps", C.c_float),
        ("target_entropy", C.c_float), ("seed", C.c_uint64),
        ("algo", C.c_int32), ("q_branches", C.c_int32), ("q_bins", C.c_int32),
        ("q_n_common", C.c_int32), ("q_common", C.c_int32 * GRL_MAX_LAYERS),
        ("q_n_branch", C.c_int32), ("q_branch", C.c_int32 * GRL_MAX_LAYERS),
        ("q_n_value", C.c_int32), ("q_value", C.c_int32 * GRL_MAX_LAYERS),
        ("q_huber", C.c_int32), ("q_double", C.c_int32), ("q_grad_clip", C.c_float), ("q_trunk_scale", C.c_float),
        ("q_per", C.c_int32), ("q_per_alpha", C.c_float), ("q_per_eps", C.c_float),
        ("replay_rgb_u8", C.c_int32), ("q_per_stratified", C.c_int32), ("q_per_alpha64", C.c_double),
        ("q_loss_sum_branches", C.c_int32),
    ]


class GrlSizes(C.Structure):
    _fields_ = [("state_bytes", C.c_size_t), ("grads_bytes", C.c_size_t), ("work_bytes", C.c_size_t),
                ("replay_bytes", C.c_size_t), ("n_params", C.c_int64), ("n_trainable", C.c_int64)]


class GrlBuffers(C.Structure):
    _fields_ = [("state", C.c_void_p), ("grads", C.c_void_p), ("work", C.c_void_p), ("replay", C.c_void_p)]


class GrlMetrics(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("policy_loss", "qf1_loss", "qf2_loss", "value_loss", "ent_coef_loss",
                                          "ent_coef", "entropy", "mean_qf1", "mean_v")]


EXPORTS = [
    "grl_last_error", "grl_version", "grl_query_sizes", "grl_create", "grl_destroy", "grl_set_stream",
    "grl_param_count", "grl_param_info", "grl_reset_optimizer", "grl_set_obs_stats", "grl_replay_add",
    "grl_replay_add_device", "grl_replay_size", "grl_train_step", "grl_compute_grads", "grl_apply_grads",
    "grl_get_metrics", "grl_act", "grl_encoder_load", "grl_encode", "grl_debug_fetch",
    "grl_profile_enable", "grl_profile_query", "grl_profile_dump", "grl_q_update_target", "grl_train_step_per",
    "grl_ae_train_step",
    "grl_ae_reconstruct", "grl_debug_store", "grl_set_learning_rate", "grl_compute_grads_staged", "grl_grad_ranges",
    "grl_observe", "grl_replay_add_observed",
    "grl_norm_update", "grl_set_running_stats", "grl_set_ret_var", "grl_get_obs_stats",
    "grl_allreduce_init", "grl_allreduce_connect", "grl_train_step_allreduce", "grl_allreduce_status", "grl_allreduce_set_overlap", "grl_allreduce_set_mode", "grl_allreduce_set_timeout",
    "grl_allreduce_disconnect",
]


class GrlError(RuntimeError):
    pass


def load_library(path=None):
    """Load libgrl.so and declare prototypes.  Raises loudly when the library is missing."""
    path = path or os.environ.get("GRL_LIBRARY") or DEFAULT_LIB
    if not os.path.exists(path):
        raise GrlError("libgrl.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; "
                       "g.build()'` (hipcc --offload-arch=gfx950); there is no CPU fallback" % path)
    try:   # PyTorch-ROCm bundles its own HIP runtime: load it first so that libgrl.so binds to the
        import torch  # noqa: F401  same libamdhip64 (two runtimes in one process do not see the GPU)
    except ImportError:
        pass
    lib = C.CDLL(path)
    vp, i32, i64, f32p, dp = C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_void_p
    lib.grl_last_error.restype = C.c_char_p
    lib.grl_version.restype = i32
    lib.grl_query_sizes.argtypes = [C.POINTER(GrlConfig), C.POINTER(GrlSizes)]
    lib.grl_create.argtypes = [C.POINTER(GrlConfig), C.POINTER(GrlBuffers), C.POINTER(vp)]
    lib.grl_destroy.argtypes = [vp]
    lib.grl_set_stream.argtypes = [vp, vp]
    lib.grl_param_count.argtypes = [vp]
    lib.grl_param_info.argtypes = [vp, i32, C.c_char_p, i32, C.POINTER(i64), C.POINTER(i64),
                                   C.POINTER(C.c_int32), C.POINTER(i64 * 4), C.POINTER(C.c_int32)]
    lib.grl_reset_optimizer.argtypes = [vp]
    lib.grl_set_obs_stats.argtypes = [vp, dp, dp, C.c_double]
    lib.grl_set_learning_rate.argtypes = [vp, C.c_float]
    lib.grl_replay_add.argtypes = [vp, f32p, f32p, f32p, f32p, f32p, i32]
    lib.grl_replay_add_device.argtypes = [vp, f32p, f32p, f32p, f32p, f32p, i32]
    lib.grl_replay_size.argtypes = [vp]
    lib.grl_replay_size.restype = i64
    lib.grl_train_step.argtypes = [vp, i32, vp, vp]
    lib.grl_compute_grads.argtypes = [vp, vp, vp]
    lib.grl_apply_grads.argtypes = [vp, C.c_float]
    lib.grl_compute_grads_staged.argtypes = [vp, i32, vp, vp]
    lib.grl_grad_ranges.argtypes = [vp, i32, i32, C.POINTER(i64), C.POINTER(i64)]
    lib.grl_q_update_target.argtypes = [vp]
    lib.grl_train_step_per.argtypes = [vp, C.c_int, C.c_double, vp]
    lib.grl_allreduce_init.argtypes = [vp, i32, i32, vp]
    lib.grl_allreduce_connect.argtypes = [vp, vp]
    lib.grl_allreduce_disconnect.argtypes = [vp]
    lib.grl_allreduce_set_overlap.argtypes = [vp, i32]
    lib.grl_allreduce_set_mode.argtypes = [vp, i32]
    lib.grl_allreduce_set_timeout.argtypes = [vp, i32]
    lib.grl_train_step_allreduce.argtypes = [vp, i32, vp, vp]
    lib.grl_allreduce_status.argtypes = [vp, C.POINTER(i64), C.POINTER(C.c_int)]
    lib.grl_norm_update.argtypes = [vp, f32p, i32]
    lib.grl_observe.argtypes = [vp, f32p, i32, i32]
    lib.grl_replay_add_observed.argtypes = [vp, f32p, f32p, f32p, i32, vp, f32p, i32]
    lib.grl_set_running_stats.argtypes = [vp, dp, dp, C.c_double]
    lib.grl_set_ret_var.argtypes = [vp, C.c_double]
    lib.grl_get_obs_stats.argtypes = [vp, dp, dp, C.POINTER(C.c_double)]
    lib.grl_ae_train_step.argtypes = [vp, vp, C.c_int]
    lib.grl_ae_reconstruct.argtypes = [vp, vp, vp]
    lib.grl_get_metrics.argtypes = [vp, C.POINTER(GrlMetrics)]
    lib.grl_act.argtypes = [vp, f32p, i32, i32, f32p, f32p]
    lib.grl_encoder_load.argtypes = [vp, C.POINTER(vp), C.POINTER(i64), i32]
    lib.grl_encode.argtypes = [vp, f32p, i32, f32p]
    lib.grl_debug_fetch.argtypes = [vp, C.c_char_p, f32p, i64]
    lib.grl_debug_fetch.restype = i64
    lib.grl_debug_store.argtypes = [vp, C.c_char_p, f32p, i64]
    lib.grl_debug_store.restype = i64
    lib.grl_profile_enable.argtypes = [vp, i32]
    lib.grl_profile_query.argtypes = [vp, C.c_char_p, C.POINTER(C.c_double), C.POINTER(i64)]
    lib.grl_profile_dump.argtypes = [vp, C.c_char_p, i32]
    return lib


def check(lib, rc):
    if rc < 0:
        raise GrlError("libgrl: %s (code %d)" % (lib.grl_last_error().decode(), rc))
    return rc


def norm_mode(normalize):
    """grl_config.normalize: 0 off, 1 observations and rewards, 2 observations only, 3 rewards only.
    Accepts a bool, one of those integers, "obs" / "reward", or an object with norm_obs / norm_reward
    (a VecNormalize wrapper)."""
    if hasattr(normalize, "norm_obs"):
        return {(True, True): 1, (True, False): 2, (False, True): 3, (False, False): 0}[
            (bool(normalize.norm_obs), bool(normalize.norm_reward))]
    if normalize in ("obs", "reward"):
        return {"obs": 2, "reward": 3}[normalize]
    mode = int(normalize)
    if mode not in (0, 1, 2, 3):
        raise GrlError("normalize must be 0..3, a bool, 'obs' or 'reward'")
    return mode


def make_config(extractor, obs_channels=2, n_direct=1, obs_dim=0, act_dim=5, layers=(64, 64), batch_size=64,
                act_batch=1, replay_capacity=50000, normalize=True, gamma=0.99, lr=3e-4, tau=0.005,
                clip_obs=10.0, clip_reward=10.0, norm_eps=1e-8, target_entropy=None, seed=0, img_hw=64,
                replay_rgb_u8=False):
    cfg = GrlConfig()
    cfg.extractor = {"mlp": 0, "augmented": 1, "nature": 2}.get(extractor, extractor)
    cfg.img_hw, cfg.obs_channels, cfg.n_direct, cfg.obs_dim, cfg.act_dim = img_hw, obs_channels, n_direct, obs_dim, act_dim
    if len(layers) > GRL_MAX_LAYERS:
        raise GrlError("at most %d hidden layers are supported" % GRL_MAX_LAYERS)
    cfg.n_layers = len(layers)
    for i, h in enumerate(layers):
        cfg.layers[i] = int(h)
    cfg.batch_size, cfg.act_batch, cfg.replay_capacity = batch_size, act_batch, replay_capacity
    cfg.normalize = norm_mode(normalize)
    cfg.gamma, cfg.lr, cfg.tau = gamma, lr, tau
    cfg.clip_obs, cfg.clip_reward, cfg.norm_eps = clip_obs, clip_reward, norm_eps
    cfg.target_entropy = -float(act_dim) if target_entropy is None else target_entropy
    cfg.seed = seed
    cfg.replay_rgb_u8 = 1 if replay_rgb_u8 else 0
    return cfg


def param_table(lib, handle):
    """[(name, offset_floats, numel, shape, trainable)] in TF creation order."""
    out = []
    for i in range(check(lib, lib.grl_param_count(handle))):
        name = C.create_string_buffer(256)
        off, numel, ndim, tr = C.c_int64(), C.c_int64(), C.c_int32(), C.c_int32()
        shape = (C.c_int64 * 4)()
        check(lib, lib.grl_param_info(handle, i, name, 256, C.byref(off), C.byref(numel), C.byref(ndim),
                                      C.byref(shape), C.byref(tr)))
        out.append((name.value.decode(), off.value, numel.value, tuple(shape[k] for k in range(ndim.value)),
                    bool(tr.value)))
    return out


def make_ae_config(batch_size=128, lr=2e-4, act_batch=16):
    """Depth auto-encoder training handle (config/encoder.yaml: batch 128, lr 2e-4)."""
    cfg = make_config("mlp", obs_dim=4096, act_dim=1, layers=(1,), batch_size=batch_size, act_batch=act_batch,
                      replay_capacity=1, normalize=False, lr=lr)
    cfg.algo = 3
    return cfg


def make_q_config(algo, obs_dim, n_branches, n_bins, common=(), branch_hidden=(64, 64), value_hidden=(64, 64),
                  batch_size=32, act_batch=1, replay_capacity=50000, normalize=False, gamma=0.99, lr=5e-4,
                  double_q=True, grad_clip=10.0, clip_obs=10.0, clip_reward=10.0, norm_eps=1e-8, seed=0,
                  prioritized=False, per_alpha=0.6, per_eps=1e-6, per_stratified=False, loss_sum_branches=False,
                  trunk_rescale=True):
    """DQN (algo='dqn': separate dueling towers) / BDQ (algo='bdq': shared trunk + branches).
    per_stratified: False = stable-baselines 2.10.x sampler (mass = random(batch) * total), True = the stratified
    sampler of OpenAI baselines / stable-baselines < 2.10 (include/grl.h, grl_config.q_per_stratified)."""
    cfg = make_config("mlp", obs_dim=obs_dim, act_dim=n_branches, layers=(1,), batch_size=batch_size,
                      act_batch=act_batch, replay_capacity=replay_capacity, normalize=normalize, gamma=gamma, lr=lr,
                      clip_obs=clip_obs, clip_reward=clip_reward, norm_eps=norm_eps, seed=seed)
    cfg.algo = {"dqn": 1, "bdq": 2}[algo]
    cfg.q_branches, cfg.q_bins = n_branches, n_bins
    for name, vals in (("common", common), ("branch", branch_hidden), ("value", value_hidden)):
        if len(vals) > GRL_MAX_LAYERS:
            raise GrlError("at most %d layers per tower" % GRL_MAX_LAYERS)
        setattr(cfg, "q_n_" + name, len(vals))
        arr = getattr(cfg, "q_" + name)
        for i, h in enumerate(vals):
            arr[i] = int(h)
    cfg.q_huber = 1 if algo == "dqn" else 0
    cfg.q_double = 1 if double_q else 0
    cfg.q_grad_clip = grad_clip
    cfg.q_trunk_scale = 1.0 / (n_branches + 1) if (algo == "bdq" and len(common) > 0 and trunk_rescale) else 1.0
    cfg.q_loss_sum_branches = 1 if (algo == "bdq" and loss_sum_branches) else 0
    cfg.q_per, cfg.q_per_alpha, cfg.q_per_eps = (1 if prioritized else 0), per_alpha, per_eps
    cfg.q_per_alpha64, cfg.q_per_stratified = float(per_alpha), (1 if per_stratified else 0)
    return cfg
