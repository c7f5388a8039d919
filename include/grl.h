/*
 * grl.h -- C ABI of libgrl.so, the MI355X (gfx950) implementation of the SAC / DQN / BDQ
 * update hot path of BarisYazici/deep-rl-grasping.
 *
 * The reference has no FFI of its own: its hot path is the stable-baselines model object it
 * drives from Python.  Each entry point below names the reference call it stands behind
 * (paths relative to /root/reference):
 *
 *   grl_create / grl_destroy      sb.SAC(policy, env, policy_kwargs=..., gamma=..., buffer_size=...,
 *                                 batch_size=..., learning_rate=...)
 *                                 manipulation_main/training/sb_helper.py:104-112,120-128
 *   grl_param_count/info          model.get_parameters() / load_parameters()  sb_helper.py:114-115
 *   grl_replay_add                SAC.learn -> replay_buffer.add(obs, action, reward, new_obs, done)
 *                                 (stable-baselines learn loop entered at sb_helper.py:175-177)
 *   grl_set_obs_stats             VecNormalize(env, norm_obs=True, norm_reward=True, clip_obs=10.)
 *                                 statistics read at replay-sample time   sb_helper.py:117-119
 *   grl_train_step                SAC._train_step + target_update_op, once per env step
 *                                 (train_freq=1, gradient_steps=1)         sb_helper.py:175-177
 *   grl_compute_grads/apply_grads the same step split around the data-parallel gradient all-reduce
 *   grl_act                       model.predict(obs, deterministic)        manipulation_main/utils.py:71,
 *                                 training/base_callbacks.py:84-88
 *   grl_encode                    SimpleAutoEncoder.encode(imgs)           gripperEnv/encoders.py:59-61,
 *                                 call site gripperEnv/sensor.py:220-222
 *
 * Conventions
 *   - plain C types only; device memory is passed as raw pointers.  All device memory is OWNED BY
 *     THE CALLER (PyTorch-ROCm tensors in the Python host); the library allocates no device memory.
 *     Sizes of the arenas the caller must provide come from grl_query_sizes().
 *   - every call enqueues its work on the HIP stream given by grl_set_stream() (default: the null
 *     stream) and returns without synchronising, except the calls documented as "host" which copy
 *     results to host memory and synchronise that stream.
 *   - return value 0 = OK, negative = error; grl_last_error() returns a thread-local message.
 *   - one handle per GPU / process; a handle is not re-entrant.
 */
#ifndef GRL_H_
#define GRL_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GRL_OK 0
#define GRL_ERR_INVALID (-1)
#define GRL_ERR_HIP (-2)
#define GRL_ERR_STATE (-3)

#define GRL_MAX_LAYERS 4

/* feature extractor selected by sb_helper.py:85-96 */
#define GRL_EXTRACTOR_MLP 0        /* sacMlp on vector observations (auto-encoder features)      */
#define GRL_EXTRACTOR_AUGMENTED 1  /* sacCnn + custom_obs_policy.create_augmented_nature_cnn(n)  */
#define GRL_EXTRACTOR_NATURE 2     /* sacCnn with the default nature_cnn over all channels       */

#define GRL_ALGO_SAC 0   /* sb.SAC  sb_helper.py:104-128 */
#define GRL_ALGO_DQN 1   /* sb.DQN  sb_helper.py:159-165 */
#define GRL_ALGO_BDQ 2   /* sb.BDQ  sb_helper.py:210-224 */
#define GRL_ALGO_AE 3    /* depth auto-encoder training: gripperEnv/encoders.py:40-50,70-136, config/encoder.yaml */

typedef struct grl_config {
  int32_t extractor;      /* GRL_EXTRACTOR_*                                                    */
  int32_t img_hw;         /* 64 (config/camera_info.yaml:1-2)                                   */
  int32_t obs_channels;   /* channels of the env observation: 2 depth, 5 RGB-D (robot.py:223-228) */
  int32_t n_direct;       /* direct features carried in the last channel (augmented only)       */
  int32_t obs_dim;        /* vector observation size (MLP extractor only)                       */
  int32_t act_dim;        /* 5 full / 3 simplified (actuator.py:60-73)                           */
  int32_t n_layers;       /* len(config[algo]['layers'])                                        */
  int32_t layers[GRL_MAX_LAYERS];
  int32_t batch_size;     /* per-GPU minibatch                                                  */
  int32_t act_batch;      /* max observations per grl_act call (vectorised envs)                */
  int64_t replay_capacity;
  int32_t normalize;      /* config['normalize'] -> VecNormalize at sample time: 0 off, 1 observations and
                             rewards (sb_helper.py:117-119), 2 observations only (norm_reward=False),
                             3 rewards only (norm_obs=False)                                     */
  float gamma, lr, tau;
  float clip_obs, clip_reward, norm_eps;
  float target_entropy;   /* -act_dim for ent_coef='auto'                                       */
  uint64_t seed;          /* device RNG (replay indices, policy noise) when none are supplied   */
  /* ---- DQN / BDQ (algo != GRL_ALGO_SAC): vector observations, extractor must be GRL_EXTRACTOR_MLP,
     act_dim = q_branches (the replay stores one bin index per action dimension)              */
  int32_t algo;           /* GRL_ALGO_*                                                         */
  int32_t q_branches;     /* 1 for DQN; action dimensions for BDQ                               */
  int32_t q_bins;         /* discrete actions (DQN) / num_actions_pad per dimension (BDQ)       */
  int32_t q_n_common;  int32_t q_common[GRL_MAX_LAYERS];   /* BDQ layers[0] (shared trunk)     */
  int32_t q_n_branch;  int32_t q_branch[GRL_MAX_LAYERS];   /* hidden layers of each branch     */
  int32_t q_n_value;   int32_t q_value[GRL_MAX_LAYERS];    /* hidden layers of the value tower */
  int32_t q_huber;        /* 1: Huber loss (DQN), 0: squared TD (BDQ)                           */
  int32_t q_double;       /* double-Q action selection                                          */
  float q_grad_clip;      /* per-variable clip_by_norm (10 in stable-baselines), 0 = off        */
  float q_trunk_scale;    /* gradient scale entering the shared trunk (BDQ: 1/(D+1))            */
  /* prioritised replay on the device (`prioritized_replay: True`, gripper_grasp.yaml:102;
     stable-baselines PrioritizedReplayBuffer semantics, csrc/per_kernels.h)                       */
  int32_t q_per;          /* 1: keep priorities in the replay arena, enable grl_train_step_per  */
  float q_per_alpha;      /* prioritized_replay_alpha (0.6)                                      */
  float q_per_eps;        /* prioritized_replay_eps (1e-6)                                       */
  /* RGB-D observations (obs_channels 5 = R, G, B, depth, pad; robot.py:201-202): keep the three colour channels of
     a stored transition as bytes (the camera delivers uint8, sensor.py:126-145) -- one packed dword + one float32
     depth per pixel, 32 KB per observation instead of 64 KB, so that the reference's 1 M-transition buffer
     (full_depth_obs.yaml / SAC_full_rgbd) is 65.5 GB of HBM.  Lossless iff the colour values are integers in
     [0, 255]; the caller vouches for that.  Requires 4 image channels (augmented extractor on 5-channel obs). */
  int32_t replay_rgb_u8;
  /* prioritised replay, continued.  q_per_stratified: 0 = stable-baselines 2.10.x `_sample_proportional`
     (mass = np.random.random(size=batch) * total; the version setup.py:7-12 pins), 1 = the stratified form of
     OpenAI baselines / stable-baselines < 2.10 (mass_k = random() * L + k * L, L = total / batch).
     q_per_alpha64: prioritized_replay_alpha as the Python float the reference passes (the exponent of
     `max_priority ** alpha` at add time; q_per_alpha is its float32 rounding, the exponent NumPy uses for the
     float32 priorities of an update); 0 = use q_per_alpha. */
  int32_t q_per_stratified;
  double q_per_alpha64;
  /* BDQ: the `bdq_sb` fork's source is not available (.gitmodules:1-3, no pinned commit), so two aggregation choices of
     its update are switches rather than citations (oracle/dqn.py): q_loss_sum_branches = 1 sums the squared TD errors
     over the action branches instead of averaging them (default 0: mean, Tavakoli et al. 2018 eq. 6);
     q_trunk_scale above = 1 disables the 1/(D+1) rescaling of the gradient entering the shared trunk. */
  int32_t q_loss_sum_branches;
} grl_config;

/* byte sizes of the four caller-provided device arenas */
typedef struct grl_sizes {
  size_t state_bytes;   /* parameters (incl. target net), Adam moments, scalars                 */
  size_t grads_bytes;   /* flat fp32 gradient bucket (the data-parallel all-reduce payload)     */
  size_t work_bytes;    /* activations, gradient workspaces, descriptor tables                  */
  size_t replay_bytes;  /* replay ring: obs, next_obs, action, reward, done                     */
  int64_t n_params;     /* floats in the parameter block (trainable + target, incl. padding)    */
  int64_t n_trainable;  /* floats covered by the gradient bucket                                */
} grl_sizes;

typedef struct grl_buffers {
  void* state;
  void* grads;
  void* work;
  void* replay;
} grl_buffers;

/* losses / diagnostics of one update, as logged by SB (logs.csv columns) */
typedef struct grl_metrics {
  float policy_loss, qf1_loss, qf2_loss, value_loss, ent_coef_loss, ent_coef, entropy;
  float mean_qf1, mean_v;
} grl_metrics;

typedef struct grl_ctx* grl_handle;

const char* grl_last_error(void);
int grl_version(void);

int grl_query_sizes(const grl_config* cfg, grl_sizes* out);
int grl_create(const grl_config* cfg, const grl_buffers* bufs, grl_handle* out);
int grl_destroy(grl_handle h);
int grl_set_stream(grl_handle h, void* hip_stream);

/* parameter layout: TF variable names (with ":0") in TF creation order, as in the SB zips */
int grl_param_count(grl_handle h);
int grl_param_info(grl_handle h, int index, char* name, int name_cap, int64_t* offset_floats,
                   int64_t* numel, int32_t* ndim, int64_t shape[4], int32_t* trainable);
/* re-initialise Adam moments / beta powers (after load_parameters) */
int grl_reset_optimizer(grl_handle h);

/* learning rate of the following updates (stable-baselines evaluates `learning_rate(progress)` before every
   update when a schedule is given; a constant needs no call: updates start with cfg.lr).  Takes effect in
   stream order; captured graphs stay valid (the value lives in device memory). */
int grl_set_learning_rate(grl_handle h, float lr);

/* VecNormalize statistics (host float64, HWC layout of the observation space or [obs_dim]);
   obs_var/ret_var are variances, epsilon is added inside.  Copied before returning. */
int grl_set_obs_stats(grl_handle h, const double* obs_mean, const double* obs_var, double ret_var);
/* The same statistics maintained ON THE DEVICE (SAC handles).  VecNormalize.step_wait -> obs_rms.update(obs)
   (stable-baselines RunningMeanStd.update / update_from_moments, wrapper created at sb_helper.py:117-119): merges the
   batch moments of the n raw observations of one env step (HOST pointer, env layout [n, 64, 64, C+1] or [n, obs_dim],
   n <= max(act_batch, 64)) into the running mean / variance / count in device memory -- float32 batch moments like
   NumPy forms them from the float32 observations, float64 Chan merge -- and refreshes the sample-time statistics;
   stream-ordered, no synchronisation.  grl_set_running_stats loads a starting point for them (the mean / var / count of a
   vecnormalize.pkl; RunningMeanStd starts at 0 / 1 / 1e-4) -- grl_set_obs_stats only loads what sampling reads --
   grl_set_ret_var pushes the host-side return variance alone, and grl_get_obs_stats (host, synchronises) copies
   mean / var [env layout] and count out for pickling.
   On a handle connected for data parallelism (grl_allreduce_connect) grl_norm_update merges the batch moments of ALL
   ranks, in rank order, into every replica (csrc/dp_kernels.h) -- the arithmetic of
   grasp_rl.parallel.share_running_stats on the host; all ranks must call it the same number of times. */
int grl_norm_update(grl_handle h, const float* obs, int n);
int grl_set_running_stats(grl_handle h, const double* obs_mean, const double* obs_var, double count);
int grl_set_ret_var(grl_handle h, double ret_var);
int grl_get_obs_stats(grl_handle h, double* obs_mean, double* obs_var, double* count);

/* append n raw (un-normalised) transitions; host pointers, obs in env layout [n,H,W,C] or [n,D] */
int grl_replay_add(grl_handle h, const float* obs, const float* act, const float* rew,
                   const float* next_obs, const float* done, int n);
/* same, device pointers already on the stream */
int grl_replay_add_device(grl_handle h, const float* obs, const float* act, const float* rew,
                          const float* next_obs, const float* done, int n);
int64_t grl_replay_size(grl_handle h);

/* n_steps updates.  idx [n_steps*batch] int64 replay indices and eps [n_steps*batch*act_dim]
   standard-normal noise are DEVICE pointers; pass NULL to draw them on the device (Philox).
   DQN / BDQ handles: `eps` carries the prioritised-replay importance weights [n_steps*batch]
   (NULL with idx == NULL: uniform sampling on the device, weights 1). */
int grl_train_step(grl_handle h, int n_steps, const int64_t* idx, const float* eps);
/* DQN / BDQ with q_per (stable-baselines DQN.learn with prioritized_replay=True: replay_buffer.sample(batch_size,
   beta=...) -> _train_step(importance weights) -> update_priorities(|td| + eps)): n_steps updates on minibatches drawn
   proportionally to the float64 leaves priority**alpha (csrc/per_kernels.h restates the segment-tree arithmetic),
   importance weights (N p)^-beta / max, then leaves <- float32(|td| + eps) ** float32(alpha).  beta > 0, at least two
   stored transitions.  u_or_null: DEVICE pointer to [n_steps*batch] float64 uniforms in [0,1) (what np.random.random
   delivers; parity tests); NULL = device Philox, 53 bits per draw. */
int grl_train_step_per(grl_handle h, int n_steps, double beta, const double* u_or_null);
/* DQN / BDQ: hard copy of the online network into the target network (target_network_update_freq) */
int grl_q_update_target(grl_handle h);
/* split form for data parallelism: grads -> (caller all-reduces the grads arena) -> apply */
int grl_compute_grads(grl_handle h, const int64_t* idx, const float* eps);
int grl_apply_grads(grl_handle h, float grad_scale);
/* The same gradient computation in two stages, for overlapping the exchange with compute: after
   stage 0 the ranges of bucket 0 (fully-connected + head layers: ~90 % of the bytes) are final in the grads arena,
   after stage 1 those of bucket 1 (convolutions, entropy coefficient).  Typical use: stage 0 -> start all-reduce of
   bucket 0 on a second stream -> stage 1 -> all-reduce bucket 1 -> grl_apply_grads.  idx / eps as for
   grl_compute_grads (stage 0 only).  Handles without a staged plan (vector observations, DQN / BDQ) do everything
   in stage 0 and report one range covering the whole bucket. */
int grl_compute_grads_staged(grl_handle h, int stage, const int64_t* idx, const float* eps);
/* The exchange step inside the library (csrc/dp_kernels.h): all-reduces written for this path over exchange
   buffers that every rank exports with hipIpcGetMemHandle and maps from its peers (xGMI between the GPUs of a node).
   The gradient computation's last launch publishes the sums it forms; then either TWO-SHOT -- reduce-scatter by the owner
   of each 1/world chunk in rank order, all-gather by pull inside the Adam kernel -- or ONE-SHOT (small worlds) -- every
   rank adds all contributions in rank order inside the Adam kernel.
     grl_allreduce_init     allocates this rank's exchange memory -- a few hundred bytes of flags and a
                            buffer of three bucket-sized arrays + two moment blocks, both fine-grained (the only device
                            memory the library allocates itself: IPC export needs allocations of its own) -- and writes
                            their two IPC handles, GRL_ALLREDUCE_HANDLE_BYTES = 128 bytes, to handle_out; the host
                            exchanges the handles of all ranks out of band (any transport: files, MPI,
                            torch.distributed.all_gather_object);
     grl_allreduce_connect  handles = world x GRL_ALLREDUCE_HANDLE_BYTES in rank order; maps the peers' memory;
     grl_allreduce_disconnect   host (synchronises this handle's stream): unmaps the peers, frees the exchange memory, and
                            the handle is a single-process handle again (grl_norm_update / grl_observe no longer merge over
                            ranks; grl_allreduce_init may be called anew).  For a set-up that failed on SOME rank -- every
                            rank then releases what it had and the job falls back together -- and for an orderly end of
                            training.  The caller guarantees that no exchange is in flight on any rank (drain, barrier).
                            No-op on a handle that was never initialised;
     grl_train_step_allreduce   n_steps data-parallel updates, each ONE graph: minibatch from this rank's replay shard,
                            gradients, exchange, Adam + Polyak with the mean gradient -- what grl_compute_grads ->
                            all-reduce -> grl_apply_grads(1 / world) does with a collective library in between.  Every
                            replica receives bit-identical sums.  All ranks must call it the same number of times.  A rank
                            WAITS for a peer that is late (GRL_TUNE dp_timeout_ms, default 120 s); when a wait runs out the
                            rank poisons the channel on every rank: this and every later call fail with GRL_ERR_STATE on
                            all of them (no replica keeps training on a partial exchange);
     grl_allreduce_status   host (synchronises): exchanges begun; error != 0 (and a negative return) after a time-out;
     grl_allreduce_set_mode 0 (default): one-shot for world <= 2, else two-shot; 1: two-shot; 2: one-shot.  All ranks must
                            choose alike, while no exchange is in flight on any rank;
     grl_allreduce_set_timeout   host, any time: bound of every wait for a peer in milliseconds from now on (0 = back to the
                            default, GRL_TUNE dp_timeout_ms or 120 s) -- short while a freshly set-up exchange is being
                            verified (a channel that cannot work should cost seconds, not minutes: bench.py's
                            make_data_parallel), long while training (a peer running an evaluation callback is merely late).
                            Takes effect for waits already in progress; values below 200 ms act as 200 ms;
     grl_allreduce_set_overlap   on = 1: grl_train_step_allreduce runs the staged plan (grl_compute_grads_staged) with BOTH
                            exchanges in its graph -- bucket 0 (dense layers, ~90 % of the bytes) is reduced and exchanged on
                            a side lane of the graph while the backward through the convolutions runs, bucket 1 follows,
                            Adam waits for both (SURVEY.md 8e "overlapped with the backward"; two-shot).  Same sums,
                            bit-identical parameters.  All ranks must choose alike.  GRL_ERR_STATE when the handle has no
                            staged plan (vector observations); off by default.
   DQN / BDQ handles (uniform replay) take part the same way: tf.clip_by_norm per variable sits between the all-reduce and
   Adam there, so their exchange is two-shot + a pull of every sum into the gradient bucket, followed by the plan's own clip +
   Adam with grad_scale 1 / world -- the SUM is clipped at world x clip, i.e. the MEAN of the replicas is clipped as one
   gradient (the same holds for grl_compute_grads -> all-reduce -> grl_apply_grads(1 / world)).  Mode 2 and the overlapped
   form are GRL_ERR_STATE on them.  Prioritised handles keep per-rank priority trees: on a connected handle
   grl_train_step_per draws from this rank's tree (importance weights against its own total and minimum), exchanges the
   gradient sums the same way and writes the priorities of its own rows back; grl_train_step_allreduce (uniform draws) is
   GRL_ERR_STATE on them.  All ranks call alike. */
#define GRL_ALLREDUCE_HANDLE_BYTES 128
int grl_allreduce_init(grl_handle h, int rank, int world, void* handle_out);
int grl_allreduce_connect(grl_handle h, const void* handles);
int grl_allreduce_disconnect(grl_handle h);
int grl_allreduce_set_overlap(grl_handle h, int on);
int grl_allreduce_set_mode(grl_handle h, int mode);
int grl_allreduce_set_timeout(grl_handle h, int ms);
int grl_train_step_allreduce(grl_handle h, int n_steps, const int64_t* idx, const float* eps);
int grl_allreduce_status(grl_handle h, int64_t* exchanges, int* error);
/* contiguous ranges (float offsets into the grads arena) of bucket 0 / 1; returns their number (<= cap) or < 0 */
int grl_grad_ranges(grl_handle h, int bucket, int cap, int64_t* offsets, int64_t* numels);
/* host: metrics of the most recent update (synchronises the stream) */
int grl_get_metrics(grl_handle h, grl_metrics* out);

/* host: actor forward for n <= act_batch observations (env layout, host ptr); `flags` is a mask of the bits below
   (any other bit: GRL_ERR_INVALID).
     GRL_ACT_DETERMINISTIC  tanh(mu) instead of a sample;
     GRL_ACT_RAW_OBS        (SAC handles) the observations are RAW and VecNormalize.normalize_obs is applied on the device
                            with the statistics grl_norm_update / grl_observe maintain (otherwise they are already
                            normalised, as VecNormalize hands them out);
     GRL_ACT_OBSERVED       (SAC handles) act on the n observations the last grl_observe uploaded; `obs` is ignored
                            (may be NULL) and nothing but eps and the actions crosses the bus.
   eps_or_null: [n,act_dim] noise for stochastic actions (host).  Synchronises the stream.
   DQN / BDQ handles: out receives the dueling Q-values [n, q_branches*q_bins]. */
#define GRL_ACT_DETERMINISTIC 1
#define GRL_ACT_RAW_OBS 2
#define GRL_ACT_OBSERVED 4
int grl_act(grl_handle h, const float* obs, int n, int flags, const float* eps_or_null,
            float* out_actions);

/* One env step's observations uploaded ONCE (SAC handles).  In SAC.learn the observations an env step returns are used
   three times -- RunningMeanStd.update (VecNormalize.step_wait), the next model.predict-style action, and two
   replay_buffer.add rows (new_obs of this step, obs of the next; stable-baselines' learn loop entered at
   sb_helper.py:175-177).  grl_observe copies obs [n, env layout] (host ptr, n <= max(act_batch, 64); copied before the
   call returns) to the device, keeps the previous call's observations beside them and, with GRL_OBSERVE_UPDATE_STATS,
   folds them into the running statistics exactly as grl_norm_update does.  grl_act(GRL_ACT_OBSERVED) then acts on them;
   grl_replay_add_observed appends the n transitions (previous observations, act, rew, newest observations, done) --
   GRL_ERR_STATE unless the last two grl_observe calls both held n observations.  Rows whose episode ended store the
   terminal observation instead: term_rows [n_term] names them, term_obs [n_term, env layout] holds their observations
   (the auto-resetting VecEnv has already put the next episode's first observation in the observed row).
   All stream-ordered; nothing synchronises. */
#define GRL_OBSERVE_UPDATE_STATS 1
int grl_observe(grl_handle h, const float* obs, int n, int flags);
int grl_replay_add_observed(grl_handle h, const float* act, const float* rew, const float* done, int n,
                            const int32_t* term_rows, const float* term_obs, int n_term);

/* GRL_ALGO_AE handles: n_steps minibatch updates of the depth auto-encoder (forward, mean-squared
   reconstruction error, backward, Keras-Adam; encoders.py:40-50,127-136).  imgs: DEVICE pointer to
   [n_steps*batch_size, 64, 64, 1] float32 depth images (train_encoder.py:19-27 preprocessing is the
   caller's).  Parameters are the 16 Keras tensors of model.h5 (grl_param_info); grl_get_metrics reports
   the reconstruction loss of the last minibatch in policy_loss; grl_encode works on the handle. */
int grl_ae_train_step(grl_handle h, const float* imgs, int n_steps);

/* GRL_ALGO_AE handles: forward pass only (Keras Model.predict, encoders.py:52-57 `test` / `predict`): the
   reconstruction of one minibatch.  imgs, out: DEVICE pointers to [batch_size, 64, 64, 1] float32.
   Parameters and optimiser state are untouched. */
int grl_ae_reconstruct(grl_handle h, const float* imgs, float* out);

/* Keras depth auto-encoder (encoder half).  weights: 8 host arrays in Keras order
   conv2d_1..3 kernel(HWIO)/bias, dense_1 kernel [2048,100]/bias; copied into the work arena. */
int grl_encoder_load(grl_handle h, const float* const* weights, const int64_t* numels, int n_arrays);
/* host: depth [n,64,64,1] float32 -> out [n,100]; n <= act_batch.  Synchronises the stream. */
int grl_encode(grl_handle h, const float* depth, int n, float* out_feat);

/* debugging / parity: copy a named internal activation of the last update to the host.
   Names: "x_obs","x_next","feat_pi","feat_vf","feat_tgt","mu","log_std","pi","logp","qf1","qf2",
   "v","v_tgt","qf1_pi","qf2_pi","rew","done".  Returns the number of floats written or <0. */
int64_t grl_debug_fetch(grl_handle h, const char* name, float* out, int64_t cap);
/* debugging / parity: overwrite the first n floats of a named internal tensor from host memory (the parity
   tests place arbitrary stored priorities "per_p" or optimiser moments "adam_m" / "adam_v" this way; no
   reference call stands behind it).  Synchronises the stream.  Returns n or <0. */
int64_t grl_debug_store(grl_handle h, const char* name, const float* in, int64_t n);

/* wall-clock free kernel timing: enables hipEvent timing of tagged kernels in later steps */
int grl_profile_enable(grl_handle h, int on);
/* host: average ms per launch of the kernel tagged `name` since enable; "" lists tags into name_out */
int grl_profile_query(grl_handle h, const char* name, double* avg_ms, int64_t* launches);
/* host: one "tag:avg_ms:launches:flops_per_launch:bytes_per_launch:executed_flops_per_launch" line per profiled tag
   (flops = algorithmic 2*M*N*K over taps that exist; executed >= flops only for the masked backward-data form) */
int grl_profile_dump(grl_handle h, char* buf, int cap);

#ifdef __cplusplus
}
#endif
#endif /* GRL_H_ */
