// engine.hip -- host side of libgrl.so: grl_ctx (parameter layout, addressing tables, problem builders, launch and
// graph machinery); the launch plans and the C ABI of include/grl.h are included parts of this translation unit:
//   plan_sac.inl (SAC update, act, encoder forward)   plan_q.inl (DQN / BDQ, prioritised replay)
//   plan_ae.inl  (auto-encoder training)              capi.inl  (extern "C" entry points)
// The element-wise, replay, prioritised-sampling and exchange kernels are compiled here; the GEMM and head kernels in
// gemm_fwd.hip / gemm_bwd.hip / gemm_wgrad.hip / heads.hip behind the launchers of launch.h.
//
// What is built here is the stable-baselines SAC update the reference drives through
// manipulation_main/training/sb_helper.py:104-128 (policy / extractor selection :85-96, extractor
// custom_obs_policy.py:15-43), restated in SURVEY.md Appendix A.  All device memory belongs to the
// caller; this file only plans where things live inside the caller's arenas and enqueues kernels.
#ifdef GRL_HOSTEMU
#include "hostemu.h"
#else
#include <hip/hip_runtime.h>
#endif

#include <algorithm>
#include <climits>
#include <cmath>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <map>
#include <string>
#include <vector>

#include "../../include/grl.h"
// the GEMM and head kernels are instantiated in their own translation units (launch.h): descriptors / argument blocks only here
#define GRL_GEMM_TYPES_ONLY
#define GRL_HEADS_TYPES_ONLY
#include "launch.h"
#include "elem_kernels.h"
#include "per_kernels.h"
#include "ae_kernels.h"
#include "q_apply_kernels.h"
#include "dp_kernels.h"

namespace grl {

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
#define HIPCHK(expr)                                                                      \
  do {                                                                                    \
    hipError_t e_ = (expr);                                                               \
    if (e_ != hipSuccess)                                                                 \
      return fail(GRL_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));        \
  } while (0)

static inline int64_t rup(int64_t x, int64_t a) { return (x + a - 1) / a * a; }

// GRL_TUNE="key=value,key=value,...": the measurement / tuning knobs of scripts/ behind ONE variable (workgroup shape per
// launch tag `i2cfg_<tag>`, `sk_wgs`, `wg_split` "a/b/c", `l0_split`, `graph_updates`, `dp_blocks` "a/b/c", `dp_coarse`,
// `dp_timeout_ms`, `heads_stamps`) and the switches that select a tested alternative launch list (`fused_adam=0`,
// `gather_prefetch=0`, `fused_q=0`, `fused_qapply=0`, `q_mfma=0`, `q_l0_chain=0`, `per_inc=0`).  None of them changes what is
// computed; `q_mfma` / `fused_q` pick stage kernels with a different summation order (each checked against the oracle).
static bool tune_str(const char* key, std::string* out) {
  const char* e = getenv("GRL_TUNE");
  if (!e) return false;
  const std::string s(e), k(key);
  size_t pos = 0;
  while (pos < s.size()) {
    size_t end = s.find(',', pos);
    if (end == std::string::npos) end = s.size();
    const std::string item = s.substr(pos, end - pos);
    const size_t eq = item.find('=');
    if (eq != std::string::npos && item.substr(0, eq) == k) { *out = item.substr(eq + 1); return true; }
    pos = end + 1;
  }
  return false;
}
static int tune_int(const char* key, int dflt) {
  std::string v;
  return tune_str(key, &v) ? atoi(v.c_str()) : dflt;
}
static int tune_int3(const char* key, int v[3]) {     // "a/b/c"
  std::string t;
  return tune_str(key, &t) ? sscanf(t.c_str(), "%d/%d/%d", &v[0], &v[1], &v[2]) : 0;
}

// ------------------------------------------------------------------------------------------------
struct Var {
  std::string name;
  int64_t off, numel;
  int ndim;
  int64_t shape[4];
  bool trainable;
};

struct Arena {
  char* base = nullptr;
  size_t off = 0;
  void* take(size_t bytes, size_t align = 256) {
    off = (size_t)rup((int64_t)off, (int64_t)align);
    void* p = base + off;   // base == nullptr: dry run, the value is only an offset
    off += bytes;
    return p;
  }
  float* f32(int64_t n) { return (float*)take((size_t)n * 4); }
};

struct Upload {
  void* dst;
  std::vector<uint8_t> bytes;
};

struct ConvGeom {
  int H, W, C, KH, KW, S, pad, OH, OW, Cout;
  int ldx = 0, ldy = 0;   // pixel strides of the input / output tensors when wider than C / Cout (0: dense) -- the
                          // layer-1 activations of the two trained networks sit side by side in one buffer
  int K() const { return KH * KW * C; }
  int LX() const { return ldx ? ldx : C; }
  int LY() const { return ldy ? ldy : Cout; }
};

struct ConvFwdTabs {
  int32_t* tab_i = nullptr;   // [M]
  int32_t* tab_r = nullptr;   // [K+1] (extra entry for the bias row of the weight gradient)
  uint64_t* vmask = nullptr;  // [M] when pad > 0
  uint8_t* tap = nullptr;     // [K+1]
  int M = 0;
  bool vec4 = false;          // 16-byte runs along the kernel index, 16-byte aligned pixel offsets
};

struct ConvBwdClass {
  int32_t *tab_i, *tab_r, *q_tab_r, *c_tab_i;
  uint64_t* vmask;
  uint8_t* tap;
  int M, K;
  bool vec4_p, vec4_q;
  float alg_frac = 1.f;   // existing (row, tap) pairs / (M * tap positions): what the masked form multiplies that is not zero
  std::vector<int32_t> ct_host;   // host copy of c_tab_i (plans that derive a second row table from it: IgemmProb.m_tab_i)
};

struct Launch {
  int variant;   // 0: P along r, Q along j   1: P along r, Q along r   2: P along i, Q along j
  int pm = 0, qm = 0, np = 1;   // addressing modes / part count (igemm.h), uniform over a launch
  bool v2 = false;              // igemm2_kernel (vectorised staging) instead of igemm_kernel
  int cfg = 0;                  // igemm2 workgroup shape
  int flags = 0;                // igemm2 instantiation flags (I2F_*)
  int sk = 0;                   // igemm_sk_kernel (streaming short-K forward): the reduction length, or 0
  std::vector<IgemmProb> probs;
  IgemmProb* d_probs = nullptr;
  int4* d_tiles = nullptr;
  int n_tiles = 0;
  Launch* filler = nullptr;     // tiles of a second instantiation carried by the same launch (igemm2_pair_kernel)
};

struct Op {
  std::string tag;
  // graph capture places an op on the main lane (0) or the side lane (1).  `fork`: a side-lane op first
  // waits for everything enqueued on the main lane so far; `join`: a main-lane op first waits for the
  // side lane.  Eager execution (profiling, GRL_NO_GRAPH) runs the list in order on one stream.
  int lane = 0;
  bool fork = false, join = false;
  std::function<void(hipStream_t)> run;
  double flops = 0;   // algorithmic FLOPs of one launch (2 * M * N * K over the taps / rows that exist)
  double flops_exec = 0;   // FLOPs the launch's MFMAs execute (>= flops: masked taps of the parity-class backward-data form)
  double bytes = 0;   // algorithmic HBM bytes of one launch
};

struct ProfAcc {
  double ms = 0;
  int64_t n = 0;
  double flops = 0, flops_exec = 0, bytes = 0;
};

struct ExtractorP {
  int64_t w[3], b[3], fw, fb;
};
struct MlpP {
  int in_dim = 0;
  int64_t w[GRL_MAX_LAYERS], b[GRL_MAX_LAYERS];
  int n_out = 0;
  int64_t ow[2], ob[2];
  int out_dim = 0;
};
struct HeadAct {
  float* z[GRL_MAX_LAYERS];
  float* out[2];
};
struct HeadGrad {
  float* g[GRL_MAX_LAYERS];
  int ld0 = 0;   // row stride of g[0] (fused heads: the critics' g[0] are column blocks of one buffer)
};

}  // namespace grl

using namespace grl;

struct grl_ctx {
  grl_config cfg;
  hipStream_t stream = nullptr;
  bool dry = false;

  // derived
  bool cnn = false;
  int C_img = 0, hw = 0, img_elems = 0, F = 0, Fc = 0, ldf = 0, A = 0, L = 0, B = 0, NA = 0;
  int hid[GRL_MAX_LAYERS];

  // parameter layout
  std::vector<Var> vars;
  int64_t n_params = 0, n_train = 0, tgt_off = 0, vf_off = 0, n_polyak = 0, ent_off = 0;
  ExtractorP ex[3];   // 0 pi, 1 values_fn, 2 target
  MlpP m_pi, m_vf, m_qf1, m_qf2, m_tgt;

  // arenas
  Arena st, gr, wk, rp;
  float *params = nullptr, *adam_m = nullptr, *adam_v = nullptr, *grads = nullptr;
  DevScalars* sc = nullptr;
  double *s_mean = nullptr, *s_std = nullptr, *s_dmean = nullptr, *s_dstd = nullptr, *s_ret = nullptr;
  // VecNormalize running statistics kept on the device (grl_norm_update): mean / var over the env-layout observation,
  // count double buffered; n_stage receives the raw observations of one env step
  double *n_mean = nullptr, *n_var = nullptr, *n_count = nullptr;
  float* n_stage = nullptr;
  int64_t n_elems = 0;
  int n_parity = 0;
  std::vector<Op> ops_act_in[4];  // entry launch of the SAC act path: [(observed by grl_observe) << 1 | (raw: VecNormalize applied on the device)]
  // observations uploaded once per env step (grl_observe): newest, the one before, terminal rows, [act | rew | done | next_row]
  float *ob_latest = nullptr, *ob_prev = nullptr, *ob_term = nullptr;
  int64_t ob_elems = 0;
  int ob_n = 0, ob_n_prev = 0;     // rows held by ob_latest / ob_prev (0 = nothing observed yet)
  float* pin_ob[2] = {nullptr, nullptr};            // page-locked staging of grl_observe / grl_replay_add_observed, alternating
  hipEvent_t pin_ob_ev[2] = {nullptr, nullptr};
  bool pin_ob_used[2] = {false, false};
  size_t pin_ob_n = 0;
  int pin_ob_next = 0;
  // replay
  float *rp_obs, *rp_next, *rp_dobs, *rp_dnext, *rp_act, *rp_rew, *rp_done;
  int64_t rp_pos = 0, rp_size = 0;
  // staging for host replay_add / act / encode
  float *stg_obs, *stg_next, *stg_act, *stg_rew, *stg_done;
  int stg_n = 0;

  // workspace (train)
  int64_t* idx_buf;
  float* eps_buf;
  float *x_obs, *x_next;
  float *a1[3], *a2[3], *a3[3], *feat[3];
  float *act, *rew, *done;
  HeadAct hPI, hVF, hQF1, hQF2, hTGT, hQF1PI, hQF2PI;
  float *pi_a, *logp, *ent;
  float *d_qf1, *d_qf2, *d_v, *d_qf1pi;
  HeadGrad gPI, gVF, gQF1, gQF2, gQF1PI;
  float *da_pi, *dmu, *dls;
  float *dfeat[2], *g3[2], *g2[2], *g1[2];
  int ld1 = 64;   // pixel stride of a1 / g1: the two trained networks side by side
  float* act_p = nullptr;            // [B, Ap] row-padded copy of the minibatch actions (Ap = rup(A, 4))
  int Ap = 0, ld_d = 1, ld_dm = 0;   // strides of the packed output-gradient buffers (fused heads)
  bool fused_heads = false;          // heads_kernels.h path (row-local chains) instead of per-layer GEMMs
  bool heads_mfma = false;           // heads_mfma.h: forward + backward of all heads as ONE launch on 16x16x4 MFMAs
  float *u_l0[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};   // layer-0 feature partials: pi, vf, qf1, qf2, target
  int l0_split = 1;
  float* g0cat = nullptr;            // [B, 3*H0]: layer-0 gradients of vf | qf1 | qf2
  // act path
  float *ax, *aa1, *aa2, *aa3, *afeat, *a_eps, *a_out;
  float* act_io_host = nullptr;   // SAC: a_eps | a_out live in page-locked host memory (plan_sac.inl)
  float* q_act_host = nullptr;         // DQN / BDQ: the Q-values of the act path in coherent host memory (plan_q.inl)
  unsigned* act_done_host = nullptr;   // SAC: completion counter of the act path's last launch (coherent host memory), polled by grl_act
  unsigned act_done_wgs = 0, act_done_seen = 0;    // increments per call (0: no counter, synchronise the stream); expected value
  HeadAct ahPI;
  // encoder path
  float *enc_w[8];
  bool enc_loaded = false;
  float *ex_in, *ec1, *ec2, *ec3, *eout;

  std::vector<Upload> uploads;
  std::vector<std::pair<void*, size_t>> zero_once;   // regions cleared once at creation
  std::vector<Launch*> launches;
  std::vector<ReduceDesc> reduces;
  ReduceDesc* d_reduces = nullptr;

  std::vector<Op> ops_rng, ops_gather, ops_grads, ops_apply, ops_act, ops_act_det, ops_act_sto, ops_enc, wgrad_ops;   // ops_rng: gather with device RNG; ops_gather: gather of explicit indices
  LossArgs loss_args;              // SAC: batch reductions appended to the reduce_slabs launch (fused heads)
  std::vector<Op> ops_grads_apply; // SAC: ops_grads with Adam + Polyak fused into the slab-reduction launch (full updates)
  // SAC, calls of several updates on the device RNG: the next minibatch is gathered inside the last launch of an update
  std::vector<Op> ops_pf_first, ops_pf_mid, ops_pf_last;
  // "gather_ride": the same three kinds of update with the minibatch IMAGES double buffered (x_obs / x_obs_b).  Flavour f reads
  // buffer f; its head launch carries the image gather of update t+1 into the other buffer as extra workgroups (heads_mfma.h),
  // its reduction launch the per-row extras of update t+1 (GatherArgs.parts)
  std::vector<Op> ops_ride_first, ops_ride_mid[2], ops_ride_last[2];
  bool ride_ok = false;
  float* x_obs_b = nullptr;
  Op wgrad_conv_alt;                    // the merged weight-gradient launch with conv1's operand in x_obs_b
  bool have_wgrad_conv_alt = false;
  ConvStackArgs ride_conv_args;         // arguments of the forward stack as planned (reading x_obs)
  const HeadsFusedArgs* ride_heads_args = nullptr;   // [3] argument blocks of the head launch (plain, first, later updates)
  int ride_heads_shape = 0, ride_heads_nblk = 0;
  LossArgs ride_lk;                     // loss arguments / extras gather the riding reductions carry (data-parallel connect)
  GatherArgs ride_g2;
  Op pf_heads[2];
  float* ae_gp4 = nullptr;               // auto-encoder step: the output gradient's four sub-position planes (plan_ae, MseArgs.gp4)
  GatherArgs pf_ga;
  int pf_gx = 0;
  bool prefetch_ok = false;
  std::vector<Op> ops_grads_apply_per;   // DQN / BDQ with prioritised replay: ... and the priority write-back
  // ... multi-update calls on the device RNG, FOUR launches per update: write-back + block-sum refresh ride on the trunk launch,
  // the sampler of the next update on the apply launch (plan_q.inl "per_pf"); the forward launch opens the update
  std::vector<Op> ops_per_pf_first, ops_per_pf_mid, ops_per_pf_last;
  // ... and uniform replay: the next update's index draw + gather ride on the apply launch (plan_q.inl "q_pf"), four launches per update
  std::vector<Op> ops_q_pf_first, ops_q_pf_mid, ops_q_pf_last;
  bool q_pf_ok = false;
  bool per_pf_ok = false;
  Op q_fwd_tick_op, q_bwd_wb_op;
  bool have_q_fwd_tick = false, have_q_bwd_wb = false;
  // data parallel, staged (grl_compute_grads_staged): stage 0 ends with the dense (fc + head) gradients final in the
  // bucket, stage 1 is the convolution backward + its weight gradients + the loss reductions
  std::vector<Op> ops_stage0, ops_stage1;
  bool staged_ok = false;
  bool conv_stack = false;                // conv1 -> conv2 -> conv3 as one sample-local launch (conv_stack.h)
  bool conv_stack_bwd = false;            // ... and the backward-data of conv3 -> conv2
  bool loss_in_reduce = false;
  float grad_scale = 1.f;   // read by the apply op

  // data parallel without a host round trip per update (grl_allreduce_*, csrc/dp_kernels.h)
  DpArgs dp;
  bool dp_on = false;
  void* dp_buf = nullptr;                // this rank's exchange data (src | red) and flags: the only device allocations the
  void* dp_flags = nullptr;              // library makes itself (IPC export needs allocations of their own)
  void* dp_peer[2 * DP_MAX_WORLD] = {nullptr};
  // the reductions that end the gradient computation, kept for the exchange step (which publishes from inside them):
  // whole bucket / dense pieces / convolution pieces (+ the loss workgroup where has_loss)
  struct ReducePlan { const int2* tiles = nullptr; int n = 0; int has_loss = 0; };
  ReducePlan red_all, red_dense, red_conv;
  AdamArgs adam_base;
  std::vector<Op> ops_dp;                // two-shot: reduce + publish | reduce-scatter | pull + Adam + Polyak (replaces the last op of ops_grads)
  std::vector<Op> ops_dp1;               // one-shot: reduce + publish | sum of all ranks + Adam + Polyak
  std::vector<Op> ops_pfdp_first, ops_pfdp_mid, ops_pfdp_last;      // prefetching sequences ending in the two-shot exchange (built at connect)
  std::vector<Op> ops_pfdp1_first, ops_pfdp1_mid, ops_pfdp1_last;   // ... in the one-shot exchange
  // ... and the "gather_ride" sequences ending in the exchange: [0] two-shot, [1] one-shot; flavour = image buffer the update reads
  std::vector<Op> ops_ridedp_first[2], ops_ridedp_mid[2][2], ops_ridedp_last[2][2];
  LossArgs pf_lk;                        // loss arguments / gather of the NEXT update as the prefetching reductions carry them (plan_sac)
  GatherArgs pf_g2;
  std::vector<Op> dp_body;               // ops_grads without its final reduction (the exchange's first kernel forms the sums)
  std::vector<Op> dp_body_per;           // DQN / BDQ with prioritised replay: the same without the gather (the sampler gathers its rows)
  int dp_mode = 0;                       // 0 auto (one-shot for world <= 2), 1 two-shot, 2 one-shot
  hipEvent_t copy_ev = nullptr;          // copy_from_caller: completion of a copy whose source is page-locked caller memory
  uint32_t* dp_err_host = nullptr;       // page-locked mailbox: a kernel that gave up waiting for a peer sets it
  DpNormArgs dp_norm;                    // running-statistics merge over the ranks (grl_norm_update on a connected handle)
  std::vector<Op> ops_dp_overlap;        // the whole overlapped update: staged gradients, two exchanges (one on a side lane), Adam
  bool dp_overlap = false;

  // graphs
  std::map<std::string, hipGraphExec_t> graphs;   // captured launch sequences, keyed by what they contain
  hipStream_t side = nullptr;                     // second capture lane (independent weight-gradient launches)
  std::vector<hipEvent_t> lane_ev;
  float apply_graph_scale = 0.f;
  bool use_graph = true;

  // profiling
  bool prof = false;
  std::map<std::string, ProfAcc> prof_acc;
  std::vector<hipEvent_t> ev;

  std::map<std::string, std::pair<const float*, int64_t>> dbg;

  ~grl_ctx() {
    if (pin_in) hipHostFree(pin_in);
    if (pin_out) hipHostFree(pin_out);
    if (act_io_host) hipHostFree(act_io_host);
    if (act_done_host) hipHostFree(act_done_host);
    if (q_act_host) hipHostFree(q_act_host);
    for (int k = 0; k < 2; ++k) {
      if (pin_ob[k]) hipHostFree(pin_ob[k]);
      if (pin_ob_ev[k]) hipEventDestroy(pin_ob_ev[k]);
    }
    for (int k = 0; k < 2; ++k) {
      if (pin_stats[k]) hipHostFree(pin_stats[k]);
      if (pin_stats_ev[k]) hipEventDestroy(pin_stats_ev[k]);
    }
    for (int p = 0; p < 2 * DP_MAX_WORLD; ++p)
      if (dp_peer[p]) (void)hipIpcCloseMemHandle(dp_peer[p]);
    if (dp_buf) (void)hipFree(dp_buf);
    if (dp_flags) (void)hipFree(dp_flags);
    if (dp_err_host) (void)hipHostFree(dp_err_host);
    if (copy_ev) (void)hipEventDestroy(copy_ev);
    for (auto* l : launches) delete l;
    for (auto e : ev) hipEventDestroy(e);
    for (auto e : lane_ev) hipEventDestroy(e);
    if (side) hipStreamDestroy(side);
    drop_graphs();
  }
  void drop_graphs() {
    for (auto& kv : graphs) (void)hipGraphExecDestroy(kv.second);
    graphs.clear();
  }
  bool graphs_on() const { return use_graph && !prof && stream != nullptr; }   // the null stream cannot be captured
  // run a sequence of op lists either eagerly or as a cached hipGraph
  int run_seq(const std::string& key, std::vector<std::vector<Op>*> seq) {
    if (!graphs_on()) {
      for (auto* ops : seq)
        if (int e = run_ops(*ops)) return e;
      return GRL_OK;
    }
    auto it = graphs.find(key);
    if (it == graphs.end()) {
      hipGraphExec_t g = nullptr;
      if (int e = capture(seq, &g)) return e;
      it = graphs.emplace(key, g).first;
    }
    hipError_t e_ = hipGraphLaunch(it->second, stream);
    if (e_ != hipSuccess) return fail(GRL_ERR_HIP, std::string("hipGraphLaunch: ") + hipGetErrorString(e_));
    return GRL_OK;
  }

  // `count` identical updates (device RNG: nothing changes on the host between them): groups of up to `graph_updates`
  // (GRL_TUNE; default 16, powers of two) go out as ONE graph -- no graph boundary between the updates of a group (measured on
  // MI355X, SAC depth B = 256: 5 090 -> 5 194 / 5 227 / 5 232 updates/s at 4 / 8 / 16 per graph)
  int run_repeated(const std::string& key, const std::vector<std::vector<Op>*>& one, int count) {
    const int max_group = std::max(1, std::min(64, tune_int("graph_updates", 16)));
    while (count > 0) {
      int group = 1;
      while (2 * group <= max_group && 2 * group <= count) group *= 2;
      if (group == 1 || !graphs_on()) {
        if (int e = run_seq(key, one)) return e;
        count -= 1;
        continue;
      }
      std::vector<std::vector<Op>*> seq;
      for (int g = 0; g < group; ++g) seq.insert(seq.end(), one.begin(), one.end());
      if (int e = run_seq(key + "_x" + std::to_string(group), seq)) return e;
      count -= group;
    }
    return GRL_OK;
  }

  // A call of n >= 2 updates with double-buffered minibatch images (plan_sac "gather_ride"): update j reads image buffer j % 2
  // and its head launch gathers the images of update j + 1 into the other one.  first: the call's first update (flavour 0);
  // mid[f] / last[f]: later updates reading buffer f.
  int run_ride(const std::string& key, std::vector<Op>* first, std::vector<Op>* mid, std::vector<Op>* last, int n_steps) {
    const int max_group = std::max(1, std::min(64, tune_int("graph_updates", 16)));
    if (n_steps <= 32 && max_group != 1) {        // short calls (SAC.learn: n = number of environments): ONE graph, cached per n
      std::vector<std::vector<Op>*> seq;
      seq.push_back(first);
      for (int j = 1; j + 1 < n_steps; ++j) seq.push_back(&mid[j & 1]);
      seq.push_back(&last[(n_steps - 1) & 1]);
      return run_seq(key + "_call_" + std::to_string(n_steps), seq);
    }
    if (int e = run_seq(key + "_first", {first})) return e;
    int j = 1;                                     // index of the next update within the call
    while (j + 1 < n_steps) {
      const int left = n_steps - 1 - j;
      int group = 1;
      while (2 * group <= max_group && 2 * group <= left) group *= 2;
      std::vector<std::vector<Op>*> seq;
      for (int g = 0; g < group; ++g) seq.push_back(&mid[(j + g) & 1]);
      // (groups are powers of two: every group of two or more starts at the parity of its first update and ends on the other)
      if (int e = run_seq(key + "_mid_p" + std::to_string(j & 1) + "_x" + std::to_string(group), seq)) return e;
      j += group;
    }
    return run_seq(key + "_last_p" + std::to_string((n_steps - 1) & 1), {&last[(n_steps - 1) & 1]});
  }

  // ---------------------------------------------------------------- helpers
  template <class T>
  T* upload_vec(Arena& a, const std::vector<T>& v) {
    T* d = (T*)a.take(std::max<size_t>(v.size(), 1) * sizeof(T));
    if (!dry && !v.empty()) {
      Upload u;
      u.dst = d;
      u.bytes.resize(v.size() * sizeof(T));
      memcpy(u.bytes.data(), v.data(), u.bytes.size());
      uploads.push_back(std::move(u));
    }
    return d;
  }

  int64_t add_var(const std::string& name, std::initializer_list<int64_t> shape, bool trainable) {
    Var v;
    v.name = name;
    v.ndim = (int)shape.size();
    v.numel = 1;
    int k = 0;
    for (auto s : shape) {
      v.shape[k++] = s;
      v.numel *= s;
    }
    for (; k < 4; ++k) v.shape[k] = 1;
    v.off = n_params;
    v.trainable = trainable;
    n_params += rup(v.numel, 4);   // keep every tensor 16-byte aligned
    vars.push_back(v);
    return v.off;
  }

  void add_extractor(const std::string& scope, ExtractorP& e, bool trainable) {
    if (!cnn) return;
    const bool aug = cfg.extractor == GRL_EXTRACTOR_AUGMENTED;
    const char* n[4] = {aug ? "cnn1" : "c1", aug ? "cnn2" : "c2", aug ? "cnn3" : "c3",
                        aug ? "cnn_fc1" : "fc1"};
    const int64_t cs[3][4] = {{8, 8, C_img, 32}, {4, 4, 32, 64}, {3, 3, 64, 64}};
    for (int l = 0; l < 3; ++l) {
      e.w[l] = add_var(scope + "/" + n[l] + "/w:0", {cs[l][0], cs[l][1], cs[l][2], cs[l][3]}, trainable);
      e.b[l] = add_var(scope + "/" + n[l] + "/b:0", {1, cs[l][3], 1, 1}, trainable);
    }
    e.fw = add_var(scope + "/" + n[3] + "/w:0", {1024, 512}, trainable);
    e.fb = add_var(scope + "/" + n[3] + "/b:0", {512}, trainable);
  }

  void add_mlp(const std::string& scope, MlpP& m, int in_dim, std::vector<std::string> outs, int out_dim,
               bool trainable) {
    m.in_dim = in_dim;
    int d = in_dim;
    for (int l = 0; l < L; ++l) {
      m.w[l] = add_var(scope + "/fc" + std::to_string(l) + "/kernel:0", {d, hid[l]}, trainable);
      m.b[l] = add_var(scope + "/fc" + std::to_string(l) + "/bias:0", {hid[l]}, trainable);
      d = hid[l];
    }
    m.n_out = (int)outs.size();
    m.out_dim = out_dim;
    for (int k = 0; k < m.n_out; ++k) {
      m.ow[k] = add_var(scope + "/" + outs[k] + "/kernel:0", {d, out_dim}, trainable);
      m.ob[k] = add_var(scope + "/" + outs[k] + "/bias:0", {out_dim}, trainable);
    }
  }

  void build_layout() {   // TF creation order == order of the shipped zips
    add_extractor("model/pi", ex[0], true);
    add_mlp("model/pi", m_pi, F, {"dense", "dense_1"}, A, true);
    vf_off = n_params;
    add_extractor("model/values_fn", ex[1], true);
    add_mlp("model/values_fn/vf", m_vf, F, {"vf"}, 1, true);
    n_polyak = n_params - vf_off;   // extractor + vf: what target_update_op averages (A.4)
    add_mlp("model/values_fn/qf1", m_qf1, F + A, {"qf1"}, 1, true);
    add_mlp("model/values_fn/qf2", m_qf2, F + A, {"qf2"}, 1, true);
    ent_off = add_var("model/log_ent_coef:0", {}, true);
    n_train = n_params;
    tgt_off = n_params;
    add_extractor("target/values_fn", ex[2], false);
    add_mlp("target/values_fn/vf", m_tgt, F, {"vf"}, 1, false);
  }

  // ---------------------------------------------------------------- conv tables
  // igemm2 fetches 4 consecutive table-addressed elements with one 16-byte load: that needs the
  // offsets of a quad to be consecutive and the first one (plus every row term) a multiple of 4
  template <class T>
  static bool runs4(const std::vector<T>& v, size_t n, bool aligned) {
    if (n % 4) return false;
    for (size_t q = 0; q < n; q += 4) {
      if (aligned && (((long)v[q]) & 3)) return false;
      for (int e = 1; e < 4; ++e)
        if ((long)v[q + e] != (long)v[q] + e) return false;
    }
    return true;
  }
  template <class T>
  static bool const4(const std::vector<T>& v, size_t n) {
    if (n % 4) return false;
    for (size_t q = 0; q < n; q += 4)
      for (int e = 1; e < 4; ++e)
        if (v[q + e] != v[q]) return false;
    return true;
  }
  template <class T>
  static bool all_mod4(const std::vector<T>& v) {
    for (auto x : v)
      if (((long)x) & 3) return false;
    return true;
  }

  ConvFwdTabs conv_fwd_tabs(const ConvGeom& g, int Bn) {
    ConvFwdTabs t;
    t.M = Bn * g.OH * g.OW;
    const int K = g.K();
    std::vector<int32_t> ti(t.M), tr(K + 1, 0);
    std::vector<uint64_t> vm(t.M, ~0ull);
    std::vector<uint8_t> tp(K + 1, 0);
    for (int b = 0; b < Bn; ++b)
      for (int oh = 0; oh < g.OH; ++oh)
        for (int ow = 0; ow < g.OW; ++ow) {
          const int m = (b * g.OH + oh) * g.OW + ow;
          const int ih0 = oh * g.S - g.pad, iw0 = ow * g.S - g.pad;
          ti[m] = ((b * g.H + ih0) * g.W + iw0) * g.LX();
          uint64_t bits = 0;
          for (int kh = 0; kh < g.KH; ++kh)
            for (int kw = 0; kw < g.KW; ++kw)
              if (ih0 + kh >= 0 && ih0 + kh < g.H && iw0 + kw >= 0 && iw0 + kw < g.W)
                bits |= 1ull << (kh * g.KW + kw);
          vm[m] = bits;
        }
    for (int kh = 0; kh < g.KH; ++kh)
      for (int kw = 0; kw < g.KW; ++kw)
        for (int c = 0; c < g.C; ++c) {
          const int r = (kh * g.KW + kw) * g.C + c;
          tr[r] = (kh * g.W + kw) * g.LX() + c;
          tp[r] = (uint8_t)(kh * g.KW + kw);
        }
    t.tab_i = upload_vec(wk, ti);
    t.tab_r = upload_vec(wk, tr);
    t.vec4 = all_mod4(ti) && runs4(tr, (size_t)K, true);
    const bool taps4 = const4(tp, (size_t)K);   // a masked quad must share one tap
    bool need_mask = false;   // TF 'SAME' padding may be one-sided (lo = 0, hi = 1)
    const uint64_t full = (g.KH * g.KW >= 64) ? ~0ull : ((1ull << (g.KH * g.KW)) - 1);
    for (auto b : vm) need_mask = need_mask || (b != full);
    if (need_mask) {
      t.vmask = upload_vec(wk, vm);
      t.tap = upload_vec(wk, tp);
      t.vec4 = t.vec4 && taps4;
    }
    return t;
  }

  // transposed gather of a strided conv, decomposed by input-position parity:
  // dX[b, S*i'+ph, S*j'+pw, cin] = sum_{jj,ll,co} dY[b, i'-jj, j'-ll, co] * W[ph+S*jj, pw+S*ll, cin, co]
  std::vector<ConvBwdClass> conv_bwd_tabs(const ConvGeom& g, int Bn) {
    std::vector<ConvBwdClass> out;
    // padded input coordinate ih' = ih + pad = S*i + ph; only the low padding shifts indices ('SAME' may pad
    // more on the high side, which the validity masks cover)
    const int IHc = (g.H + g.pad + g.S - 1) / g.S, IWc = (g.W + g.pad + g.S - 1) / g.S;
    const int TJ = (g.KH + g.S - 1) / g.S, TL = (g.KW + g.S - 1) / g.S;
    for (int ph = 0; ph < g.S; ++ph)
      for (int pw = 0; pw < g.S; ++pw) {
        ConvBwdClass c;
        c.M = Bn * IHc * IWc;
        c.K = TJ * TL * g.Cout;
        std::vector<int32_t> ti(c.M), tr(c.K), qt(c.K), ct(c.M);
        std::vector<uint64_t> vm(c.M);
        std::vector<uint8_t> tp(c.K);
        double live_taps = 0;
        for (int b = 0; b < Bn; ++b)
          for (int i = 0; i < IHc; ++i)
            for (int j = 0; j < IWc; ++j) {
              const int m = (b * IHc + i) * IWc + j;
              ti[m] = ((b * g.OH + i) * g.OW + j) * g.LY();
              uint64_t bits = 0;
              for (int jj = 0; jj < TJ; ++jj)
                for (int ll = 0; ll < TL; ++ll) {
                  const int oh = i - jj, ow = j - ll;
                  const int kh = ph + g.S * jj, kw = pw + g.S * ll;
                  if (oh >= 0 && oh < g.OH && ow >= 0 && ow < g.OW && kh < g.KH && kw < g.KW)
                    bits |= 1ull << (jj * TL + ll);
                }
              vm[m] = bits;
              const int ih = g.S * i + ph - g.pad, iw = g.S * j + pw - g.pad;
              ct[m] = (ih >= 0 && iw >= 0 && ih < g.H && iw < g.W) ? ((b * g.H + ih) * g.W + iw) * g.LX() : -1;
              if (ct[m] >= 0) live_taps += __builtin_popcountll(bits);
            }
        c.alg_frac = (float)(live_taps / ((double)c.M * TJ * TL));
        for (int jj = 0; jj < TJ; ++jj)
          for (int ll = 0; ll < TL; ++ll)
            for (int co = 0; co < g.Cout; ++co) {
              const int r = (jj * TL + ll) * g.Cout + co;
              tr[r] = -(jj * g.OW + ll) * g.LY() + co;
              tp[r] = (uint8_t)(jj * TL + ll);
              const int kh = std::min(ph + g.S * jj, g.KH - 1), kw = std::min(pw + g.S * ll, g.KW - 1);
              qt[r] = ((kh * g.KW + kw) * g.C) * g.Cout + co;
            }
        c.vec4_p = all_mod4(ti) && runs4(tr, tr.size(), true) && const4(tp, tp.size());
        c.vec4_q = runs4(qt, qt.size(), true) && (g.Cout % 4 == 0);
        c.tab_i = upload_vec(wk, ti);
        c.tab_r = upload_vec(wk, tr);
        c.q_tab_r = upload_vec(wk, qt);
        c.c_tab_i = upload_vec(wk, ct);
        c.ct_host = ct;
        c.vmask = upload_vec(wk, vm);
        c.tap = upload_vec(wk, tp);
        out.push_back(c);
      }
    return out;
  }

  // The same transposed gather with EXACT taps.  In the parity form above every class reduces over all TJ x TL tap
  // positions and border pixels multiply zeros for the taps that fall outside the output (3x3 s1 over 6x6: 4 of 9
  // taps exist on average, 2.25x the MACs; 4x4 s2 over 15x15 padded to 16x16: 1.78x).  Here the input pixels of a
  // parity class are split further by their valid-tap set per dimension -- rows i' of one (ph, tap-set) group times
  // columns j' of one (pw, tap-set) group form a problem whose reduction covers exactly the taps that exist, K =
  // nh * nw * Cout, no validity masks (PM_TABLE).  Taps keep the order of the parity form and a dropped tap is a run
  // of 2 * 32 exact zeros there, so the sums are bit-identical.  Pixels no output touches (conv2: row / column 14)
  // belong to no problem: `untouched` lists them (offsets of their first channel) and the caller keeps them zero.
  std::vector<ConvBwdClass> conv_bwd_tabs_exact(const ConvGeom& g, int Bn, std::vector<int32_t>* untouched) {
    std::vector<ConvBwdClass> out;
    struct Grp { int par; std::vector<int> taps; std::vector<int> pos, img; };   // parity, tap indices jj, i' values, image coords
    auto groups = [&](int Himg, int KHh, int OHh) {
      std::vector<Grp> gs;
      const int IHc = (Himg + g.pad + g.S - 1) / g.S, TJ = (KHh + g.S - 1) / g.S;
      for (int ph = 0; ph < g.S; ++ph)
        for (int i = 0; i < IHc; ++i) {
          const int ih = g.S * i + ph - g.pad;
          if (ih < 0 || ih >= Himg) continue;
          std::vector<int> taps;
          for (int jj = 0; jj < TJ; ++jj)
            if (i - jj >= 0 && i - jj < OHh && ph + g.S * jj < KHh) taps.push_back(jj);
          Grp* at = nullptr;
          for (auto& q : gs)
            if (q.par == ph && q.taps == taps) at = &q;
          if (!at) { gs.push_back(Grp{ph, taps, {}, {}}); at = &gs.back(); }
          at->pos.push_back(i);
          at->img.push_back(ih);
        }
      return gs;
    };
    const std::vector<Grp> gh = groups(g.H, g.KH, g.OH), gw = groups(g.W, g.KW, g.OW);
    for (const Grp& a : gh)
      for (const Grp& b : gw) {
        if (a.taps.empty() || b.taps.empty()) {
          if (untouched)
            for (int bb = 0; bb < Bn; ++bb)
              for (int ih : a.img)
                for (int iw : b.img) untouched->push_back(((bb * g.H + ih) * g.W + iw) * g.LX());
          continue;
        }
        ConvBwdClass c;
        const int nh = (int)a.taps.size(), nw = (int)b.taps.size();
        c.M = Bn * (int)a.pos.size() * (int)b.pos.size();
        c.K = nh * nw * g.Cout;
        std::vector<int32_t> ti(c.M), tr(c.K), qt(c.K), ct(c.M);
        int m = 0;
        for (int bb = 0; bb < Bn; ++bb)
          for (size_t x = 0; x < a.pos.size(); ++x)
            for (size_t y = 0; y < b.pos.size(); ++y, ++m) {
              ti[m] = ((bb * g.OH + a.pos[x]) * g.OW + b.pos[y]) * g.LY();
              ct[m] = ((bb * g.H + a.img[x]) * g.W + b.img[y]) * g.LX();
            }
        for (int x = 0; x < nh; ++x)
          for (int y = 0; y < nw; ++y)
            for (int co = 0; co < g.Cout; ++co) {
              const int r = (x * nw + y) * g.Cout + co;
              const int jj = a.taps[x], ll = b.taps[y];
              tr[r] = -(jj * g.OW + ll) * g.LY() + co;
              qt[r] = (((a.par + g.S * jj) * g.KW + (b.par + g.S * ll)) * g.C) * g.Cout + co;
            }
        c.vec4_p = all_mod4(ti) && runs4(tr, tr.size(), true);
        c.vec4_q = runs4(qt, qt.size(), true) && (g.Cout % 4 == 0);
        c.tab_i = upload_vec(wk, ti);
        c.tab_r = upload_vec(wk, tr);
        c.q_tab_r = upload_vec(wk, qt);
        c.c_tab_i = upload_vec(wk, ct);
        c.ct_host = ct;
        c.vmask = nullptr;
        c.tap = nullptr;
        out.push_back(c);
      }
    return out;
  }

  // ---------------------------------------------------------------- problem builders
  static IgemmProb blank() {
    IgemmProb p;
    memset(&p, 0, sizeof(p));
    p.p_ones_i = -1;
    p.split = 1;
    return p;
  }
  static void set_split(IgemmProb& p, int split_target) {
    const int tiles_r = (p.K + 31) / 32;
    int s = std::max(1, std::min(split_target, tiles_r));
    const int per = (tiles_r + s - 1) / s;
    p.k_chunk = per * 32;
    p.split = (p.K + p.k_chunk - 1) / p.k_chunk;
    p.slab_stride = (int64_t)p.M * p.N;
  }
  static void single_part(IgemmProb& p) { p.p_k0 = p.p_k1 = INT_MAX; }

  static IgemmProb dense_fwd(const float* x0, int ld0, int K0, const float* x1, int ld1, int K1, int M,
                             const float* w, int N, const float* bias, float* y, int ldy, int act) {
    IgemmProb p = blank();
    p.M = M; p.N = N; p.K = K0 + K1;
    p.p_base[0] = x0; p.p_ld_i[0] = ld0; p.p_ld_r[0] = 1;
    p.p_base[1] = x1; p.p_ld_i[1] = ld1; p.p_ld_r[1] = 1;
    p.p_k0 = K1 > 0 ? K0 : INT_MAX; p.p_k1 = INT_MAX;
    p.q_base[0] = w; p.q_ld_r[0] = N; p.q_ld_j[0] = 1;
    p.q_base[1] = w + (int64_t)K0 * N; p.q_ld_r[1] = N; p.q_ld_j[1] = 1;
    p.c = y; p.ldc = ldy; p.bias = bias; p.act = act;
    set_split(p, 1);
    return p;
  }

  struct BwdPart {
    const float* g; int ldg; int Nl; const float* w;
  };
  // dx[m, col0 + j] = sum_parts sum_n g_p[m, n] * W_p[col0 + j, n]   (optionally masked by mask > 0)
  static IgemmProb dense_bwd(const std::vector<BwdPart>& parts, int M, int col0, int ncols, float* dx,
                             int lddx, const float* mask) {
    IgemmProb p = blank();
    p.M = M; p.N = ncols; p.K = 0;
    int bounds[3] = {INT_MAX, INT_MAX, INT_MAX};
    for (size_t k = 0; k < parts.size(); ++k) {
      p.p_base[k] = parts[k].g; p.p_ld_i[k] = parts[k].ldg; p.p_ld_r[k] = 1;
      p.q_base[k] = parts[k].w + (int64_t)col0 * parts[k].Nl; p.q_ld_r[k] = 1; p.q_ld_j[k] = parts[k].Nl;
      p.K += parts[k].Nl;
      bounds[k] = p.K;
    }
    p.p_k0 = parts.size() > 1 ? bounds[0] : INT_MAX;
    p.p_k1 = parts.size() > 2 ? bounds[1] : INT_MAX;
    p.c = dx; p.ldc = lddx; p.relu_mask = mask;
    set_split(p, 1);
    return p;
  }

  // slab[(Kin + ones), N] = [x^T ; 1^T] * g     (weight + bias gradient of a dense layer)
  // Kin_pad >= Kin: x columns [Kin, Kin_pad) exist (row padding of the buffer, kept zero) and are carried
  // along so that the row count is a multiple of 4 (igemm2 fetches 4 rows of x^T per load); their
  // slab rows are never reduced into the gradient bucket.
  static IgemmProb dense_wgrad(const float* x, int ldx, int Kin, bool ones, const float* g, int ldg, int N,
                               int rows, float* slab, int split_target, int Kin_pad = 0) {
    IgemmProb p = blank();
    const int Kp = std::max(Kin, Kin_pad);
    p.M = Kp + (ones ? 1 : 0); p.N = N; p.K = rows;
    p.p_base[0] = x; p.p_ld_i[0] = 1; p.p_ld_r[0] = ldx; single_part(p);
    p.p_ones_i = ones ? Kp : -1;
    p.q_base[0] = g; p.q_ld_r[0] = ldg; p.q_ld_j[0] = 1;
    p.c = slab; p.ldc = N;
    set_split(p, split_target);
    return p;
  }

  static IgemmProb conv_fwd(const float* x, const ConvFwdTabs& t, const ConvGeom& g, const float* w,
                            const float* bias, float* y, int act, float alpha) {
    IgemmProb p = blank();
    p.M = t.M; p.N = g.Cout; p.K = g.K();
    p.p_base[0] = x; p.p_tab_i = t.tab_i; p.p_tab_r = t.tab_r; p.p_vmask_i = t.vmask; p.p_tap_r = t.tap;
    single_part(p);
    p.q_base[0] = w; p.q_ld_r[0] = g.Cout; p.q_ld_j[0] = 1;
    p.c = y; p.ldc = g.LY(); p.bias = bias; p.act = act; p.act_alpha = alpha;
    p.vflags = t.vec4 ? VF_P_TABS : 0;
    set_split(p, 1);
    return p;
  }

  static IgemmProb conv_bwd(const float* gy, const ConvBwdClass& c, const ConvGeom& g, const float* w,
                            float* dx, const float* mask) {
    IgemmProb p = blank();
    p.M = c.M; p.N = g.C; p.K = c.K;
    p.p_base[0] = gy; p.p_tab_i = c.tab_i; p.p_tab_r = c.tab_r; p.p_vmask_i = c.vmask; p.p_tap_r = c.tap;
    single_part(p);
    p.q_base[0] = w; p.q_tab_r = c.q_tab_r; p.q_ld_j[0] = g.Cout;
    p.c = dx; p.c_tab_i = c.c_tab_i; p.relu_mask = mask;
    p.vflags = (c.vec4_p ? VF_P_TABS : 0) | (c.vec4_q ? VF_Q_TAB : 0) | ((g.C % 4 == 0) ? VF_CT4 : 0);
    p.alg_frac = c.alg_frac;
    set_split(p, 1);
    return p;
  }

  // (n_side > 1: the output gradients of n_side networks side by side in gy's rows -- one problem, N = n_side * Cout)
  static IgemmProb conv_wgrad(const float* x, const ConvFwdTabs& t, const ConvGeom& g, const float* gy,
                              float* slab, int split_target, int n_side = 1) {
    IgemmProb p = blank();
    p.M = g.K() + 1; p.N = n_side * g.Cout; p.K = t.M;
    p.p_base[0] = x; p.p_tab_i = t.tab_r; p.p_tab_r = t.tab_i; single_part(p);
    if (t.vmask) { p.p_vmask_i = t.vmask; p.p_tap_r = t.tap; p.p_mask_swap = 1; }   // padded conv: skip out-of-image taps
    p.p_ones_i = g.K();
    p.q_base[0] = gy; p.q_ld_r[0] = g.LY(); p.q_ld_j[0] = 1;
    p.c = slab; p.ldc = p.N;
    p.vflags = t.vec4 ? VF_P_TABS : 0;
    set_split(p, split_target);
    return p;
  }

  // ---------------------------------------------------------------- igemm2 eligibility / shape choice
  static bool al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }
  static bool v2_prob_ok(const IgemmProb& p, int variant) {
    if (p.p_k0 < p.K) return false;                       // multi-part operands stay on igemm_kernel
    if (!al16(p.p_base[0]) || !al16(p.q_base[0])) return false;
    const bool ptab = p.p_tab_i != nullptr, qtab = p.q_tab_r != nullptr;
    const int Meff = p.p_ones_i >= 0 ? p.M - 1 : p.M;
    if (p.p_ones_i >= 0 && (p.p_ones_i != p.M - 1 || variant != 2)) return false;
    if (variant == 2) {                                   // P along i
      if (Meff % 4) return false;
      if (ptab) { if (!(p.vflags & VF_P_TABS) || p.p_vmask_i) return false; }
      else if (p.p_ld_i[0] != 1 || (p.p_ld_r[0] % 4)) return false;
    } else {                                              // P along r
      if (ptab) { if (!(p.vflags & VF_P_TABS) || (p.K % 4)) return false; }
      else if (p.p_ld_r[0] != 1 || (p.p_ld_i[0] % 4) || p.p_ld_i[0] < rup(p.K, 4)) return false;
    }
    if (variant == 1) {                                   // Q along r
      if (p.K % 4) return false;
      if (qtab) { if (!(p.vflags & VF_Q_TAB) || (p.q_ld_j[0] % 4)) return false; }
      else if (p.q_ld_r[0] != 1 || (p.q_ld_j[0] % 4)) return false;
    } else {                                              // Q along j
      if (qtab) return false;
      // a quad straddling N reads row padding (never stored): the row stride must cover it
      if (p.q_ld_j[0] != 1 || (p.q_ld_r[0] % 4) || p.q_ld_r[0] < rup(p.N, 4)) return false;
    }
    return true;
  }
  // workgroup shape of a v2 launch: narrow outputs -> 128x32; few 64x64 tiles -> 32x64 with the reduction
  // split over wave pairs; otherwise 64x64 (shape 2, a 4-way split, stays selectable by GRL_TUNE i2cfg_<tag>)
  static int v2_pick_cfg(const std::vector<IgemmProb>& probs, int variant, const std::string& tag) {
    const int forced = tune_int(("i2cfg_" + tag).c_str(), -1);
    if (forced >= 0) return forced;
    int maxN = 0;
    long tiles64 = 0;
    bool ones = false, longk = true;
    for (auto& p : probs) {
      maxN = std::max(maxN, p.N);
      tiles64 += (long)p.split * ((p.M + 63) / 64) * ((p.N + 63) / 64);
      ones = ones || p.p_ones_i >= 0;
      longk = longk && std::min(p.K, p.k_chunk) >= 256;
    }
    if (maxN <= 32) return 1;
    // measured on MI355X (scripts/cfg_sweep.sh): below ~1.5 64x64 tiles per CU the 32x64 shape with a
    // 2-way reduction split wins (twice the workgroups, 48 KB of LDS so that three share a CU)
    (void)longk;
    if (!ones && variant != 2 && tiles64 < 400) return 3;
    return 0;
  }

  // igemm2 instantiation key of a launch (see the dispatch switch in add_launch)
  static int v2_key(const Launch* l) { return l->variant * 10000 + l->pm * 1000 + l->qm * 100 + l->cfg * 10 + l->flags; }
  // pairs of instantiations igemm2_pair_kernel is built for: a backward-data stage + weight-gradient fillers
  static bool pair_ok(const Launch* a, const Launch* b) {
    if (!a->v2 || !b->v2 || a->sk || b->sk || v2_key(b) != 21001) return false;
    const int ka = v2_key(a);
    return ka == 11130;     // conv3_bwd (exact taps), 32x64 shape
  }

  // finish a launch: tile list (heaviest reductions first), upload, wrap as an Op.  `filler`: independent problems
  // of a second instantiation (variant `fvariant`, workgroup shape `fcfg`) whose tiles follow the launch's own in
  // the same grid; when the pair is not one igemm2_pair_kernel is built for they become a launch of their own.
  void add_launch(std::vector<Op>& ops, const std::string& tag, int variant, std::vector<IgemmProb> probs,
                  const std::string& ftag = "", int fvariant = 0, std::vector<IgemmProb> fprobs = {}, int force_cfg = -1) {
    if (probs.empty()) {
      if (!fprobs.empty()) add_launch(ops, ftag, fvariant, fprobs, "", 0, {}, 0);
      return;
    }
    if (!fprobs.empty()) {
      std::vector<Op> tmp_a, tmp_b;
      {   // the fillers will occupy list positions n_a..: let the placement of the launch's own tiles count them in
        lpt_extra_tiles = 0; lpt_extra_w = 0;
        for (auto& p : fprobs) {
          const int Mt = p.p_ones_i >= 0 ? p.M - 1 : p.M;
          lpt_extra_tiles += p.split * ((Mt + 63) / 64) * ((p.N + 63) / 64);
          lpt_extra_w = std::max(lpt_extra_w, (double)((std::min(p.K, p.k_chunk) + 31) / 32));
        }
      }
      add_launch(tmp_a, tag, variant, probs);
      lpt_extra_tiles = 0; lpt_extra_w = 0;
      Launch* la = launches.back();
      add_launch(tmp_b, ftag, fvariant, fprobs, "", 0, {}, 0);      // fillers keep the 64x64 shape of the merged launch
      Launch* lb = launches.back();
      if (tmp_a.size() == 1 && tmp_b.size() == 1 && pair_ok(la, lb)) {
        la->filler = lb;
        Op op = tmp_a[0];
        op.tag = tag;
        op.flops += tmp_b[0].flops;
        op.flops_exec += tmp_b[0].flops_exec;
        grl_ctx* self = this;
        const std::string t2 = tag;
        op.run = [la, lb, t2](hipStream_t s) {
          launch_igemm2_pair(v2_key(la), la->n_tiles, lb->n_tiles, s, la->d_probs, la->d_tiles, lb->d_probs, lb->d_tiles, t2.c_str());
        };
        (void)self;
        if (getenv("GRL_PLAN_DUMP"))
          fprintf(stderr, "grl plan: %-14s carries %d filler tiles of '%s' behind its own %d\n", tag.c_str(), lb->n_tiles,
                  ftag.c_str(), la->n_tiles);
        ops.push_back(std::move(op));
      } else {
        for (auto& o : tmp_a) ops.push_back(o);
        for (auto& o : tmp_b) ops.push_back(o);
      }
      return;
    }
    Launch* l = new Launch();
    l->variant = variant;
    l->probs = std::move(probs);
    // addressing modes are compile-time in the kernels: derive them and insist they are uniform
    auto pm_of = [](const IgemmProb& p) { return p.p_tab_i ? (p.p_vmask_i ? PM_TABLE_MASK : PM_TABLE) : PM_AFFINE; };
    auto qm_of = [](const IgemmProb& p) { return p.q_tab_r ? QM_TABLE : QM_AFFINE; };
    l->pm = pm_of(l->probs[0]);
    l->qm = qm_of(l->probs[0]);
    for (auto& p : l->probs)
      if (pm_of(p) != l->pm || qm_of(p) != l->qm || ((p.p_tab_i != nullptr) != (p.p_tab_r != nullptr))) {
        fprintf(stderr, "grl: launch '%s' mixes addressing modes\n", tag.c_str());
        abort();
      }
    for (auto& p : l->probs)
      if (p.p_k0 < p.K) l->np = 3;
    const char* nv2 = getenv("GRL_NO_V2");
    l->v2 = !(nv2 && nv2[0] == '1');
    for (auto& p : l->probs) l->v2 = l->v2 && v2_prob_ok(p, variant);
    // ones rows / K tails select a kernel instantiation: they must be uniform over the launch
    {
      bool any_ones = false, all_ones = true, ktail = false;
      for (auto& p : l->probs) {
        any_ones = any_ones || p.p_ones_i >= 0;
        all_ones = all_ones && p.p_ones_i >= 0;
        ktail = ktail || (p.K % 4) != 0;
      }
      if (any_ones != all_ones) l->v2 = false;
      if (ktail && !(variant == 0 && l->pm == PM_AFFINE && l->qm == QM_AFFINE)) l->v2 = false;
      l->flags = (any_ones ? I2F_ONES : 0) | (ktail ? I2F_KTAIL : 0);
    }
    l->cfg = l->v2 ? (force_cfg >= 0 ? force_cfg : v2_pick_cfg(l->probs, variant, tag)) : 0;
    if (l->v2)   // outputs that can leave as 16-byte stores (igemm2 wide epilogue)
      for (auto& p : l->probs)
        if (al16(p.c) && (p.ldc % 4) == 0 && (p.N % 4) == 0 && (p.slab_stride % 4) == 0 &&
            (!p.c_tab_i || (p.vflags & VF_CT4)) && (!p.m_tab_i || (p.vflags & VF_CT4)) && (!p.relu_mask || al16(p.relu_mask)) &&
            (!p.bias || al16(p.bias)))
          p.vflags |= VF_C_VEC;
    // short reductions with a narrow output (first convolution of the extractors): streaming kernel, igemm_sk.h
    {
      bool ok = l->v2 && variant == 0 && l->pm == PM_TABLE && l->qm == QM_AFFINE && l->flags == 0;
      for (auto& p : l->probs)
        ok = ok && (p.K == 64 || p.K == 32) && p.K == l->probs[0].K && p.N <= 32 && (p.N % 4) == 0 && p.split == 1 &&
             (p.vflags & VF_P_TABS) && (p.vflags & VF_C_VEC) && !p.c_tab_i && !p.relu_mask && !p.accumulate &&
             p.q_ld_j[0] == 1 && (p.q_ld_r[0] % 4) == 0 && p.M >= 256;
      l->sk = ok ? l->probs[0].K : 0;
    }
    if (l->sk) {
      long total = 0;
      for (auto& p : l->probs) total += (p.M + 127) / 128;
      long slots = 512;                                                // ~2 workgroups per CU, each streams `per` tiles
      slots = std::max(1, tune_int("sk_wgs", (int)slots));
      const int per = (int)std::max<long>(1, (total + slots - 1) / slots);
      std::vector<int4> work;
      double flops = 0;
      for (size_t pi = 0; pi < l->probs.size(); ++pi) {
        const IgemmProb& p = l->probs[pi];
        flops += 2.0 * p.M * p.N * p.K;
        const int nt = (p.M + 127) / 128;
        for (int t0 = 0; t0 < nt; t0 += per) work.push_back(make_int4((int)pi, t0, std::min(per, nt - t0), 0));
      }
      l->n_tiles = (int)work.size();
      l->d_probs = upload_vec(wk, per_tile_descs(l->probs, work));
      l->d_tiles = upload_vec(wk, work);
      launches.push_back(l);
      if (getenv("GRL_PLAN_DUMP"))
        fprintf(stderr, "grl plan: %-14s streaming short-K kernel K %d  probs %zu  row tiles per workgroup %d  tiles %d\n",
                tag.c_str(), l->sk, l->probs.size(), per, l->n_tiles);
      Op op;
      op.tag = tag;
      op.flops = op.flops_exec = flops;
      op.run = [l](hipStream_t s) { launch_igemm_sk(l->sk, l->n_tiles, s, l->d_probs, l->d_tiles); };
      ops.push_back(op);
      return;
    }
    const int BMt = l->v2 ? i2_bm(l->cfg) : 64, BNt = l->v2 ? i2_bn(l->cfg) : 64;

    double flops = 0, flops_alg = 0;
    for (auto& p : l->probs) {
      flops += 2.0 * p.M * p.N * p.K;
      flops_alg += 2.0 * p.M * p.N * p.K * (p.alg_frac > 0.f ? (double)p.alg_frac : 1.0);
    }
    std::vector<int4> tiles = tile_list(l->probs, l->v2, BMt, BNt);
    if (variant == 2) tiles = xcd_order(tiles, l->probs, BMt, BNt);
    if (variant == 1 && l->v2) tiles = lpt_order(tiles, l->probs, BMt, BNt, lpt_extra_tiles, lpt_extra_w, tag);
    l->n_tiles = (int)tiles.size();
    if (getenv("GRL_PLAN_DUMP"))
      fprintf(stderr, "grl plan: %-14s variant %d pm %d qm %d np %d  %s cfg %d flags %d  probs %zu  tiles %d\n", tag.c_str(),
              variant, l->pm, l->qm, l->np, l->v2 ? "v2" : "v1", l->cfg, l->flags, l->probs.size(), l->n_tiles);
    {
      std::vector<IgemmProb> descs = per_tile_descs(l->probs, tiles);
#ifdef GRL_TILE_TRACE
      unsigned long long* tr = (unsigned long long*)wk.take(descs.size() * 48);   // measurement build: one 6-word record per workgroup
      zero_once.push_back({tr, descs.size() * 48});
      for (size_t i = 0; i < descs.size(); ++i) descs[i].dbg_t = tr + 6 * i;
      static int trace_seq = 0;
      dbg["trace" + std::to_string(trace_seq++) + "_" + tag] = {(const float*)tr, (int64_t)descs.size() * 12};
#endif
      l->d_probs = upload_vec(wk, descs);
    }
    l->d_tiles = upload_vec(wk, tiles);
    launches.push_back(l);
    Op op;
    op.tag = tag;
    op.flops = flops_alg;
    op.flops_exec = flops;
    op.run = [l, tag](hipStream_t s) {
      if (l->v2) {
        const int key = v2_key(l);
        if (l->variant == 0) launch_igemm2_fwd(key, l->n_tiles, s, l->d_probs, l->d_tiles, tag.c_str());
        else if (l->variant == 1) launch_igemm2_bwd(key, l->n_tiles, s, l->d_probs, l->d_tiles, tag.c_str());
        else launch_igemm2_wgrad(key, l->n_tiles, s, l->d_probs, l->d_tiles, tag.c_str());
        return;
      }
      launch_igemm(l->np * 1000 + l->pm * 100 + l->qm * 10 + l->variant, l->n_tiles, s, l->d_probs, l->d_tiles, tag.c_str());
    };
    ops.push_back(std::move(op));
  }

  // work list of a launch: {problem, reduction chunk, row tile, column tile}, longest reduction chunks first
  static std::vector<int4> tile_list(const std::vector<IgemmProb>& probs, bool v2, int BMt, int BNt) {
    std::vector<int4> tiles;
    std::vector<int> order(probs.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = (int)i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
      return std::min(probs[a].K, probs[a].k_chunk) > std::min(probs[b].K, probs[b].k_chunk);
    });
    for (int pi : order) {
      const IgemmProb& p = probs[pi];
      const int Mt = (v2 && p.p_ones_i >= 0) ? p.M - 1 : p.M;   // v2: the ones row rides on row tile 0
      for (int s = 0; s < p.split; ++s)
        for (int ti = 0; ti < (Mt + BMt - 1) / BMt; ++ti)
          for (int tj = 0; tj < (p.N + BNt - 1) / BNt; ++tj) tiles.push_back(make_int4(pi, s, ti, tj));
    }
    return tiles;
  }

  // Reduction splits of the three convolution weight gradients, chosen with the placement model of xcd_order: the
  // merged launch lasts as long as its fullest CU works (0.48 us per 32-deep slab it holds, scripts/tile_trace.py) and
  // every extra split adds a slab the reduction pass has to write and read back (~4 TB/s effective).  The dense
  // problems of the same launch enter with their shapes only.  Measured at the headline shape: hand-tuned 72/12/6
  // (fullest CU 56 slabs) 4 560 updates/s, model's choice (49 slabs) 4 590.
  void pick_wgrad_splits(const ConvGeom* cg, const ConvFwdTabs* ft, int n_side, int wsplit[3], int rider_budget) {
    auto shape = [](int M, int N, int K, int split) {
      IgemmProb p = blank();
      p.M = M; p.N = N; p.K = K; p.p_ones_i = M - 1;
      set_split(p, split);
      return p;
    };
    std::vector<IgemmProb> light;
    for (int n = 0; n < 2; ++n) light.push_back(shape(1024 + 1, 512, B, 1));
    const MlpP* ms[4] = {&m_pi, &m_vf, &m_qf1, &m_qf2};
    for (const MlpP* m : ms) {
      int d = m->in_dim;
      for (int l = 0; l < L; ++l) { light.push_back(shape(d + 1, hid[l], B, 1)); d = hid[l]; }
      for (int k = 0; k < m->n_out; ++k) light.push_back(shape(d + 1, m->out_dim, B, 1));
    }
    take_riders(light, rider_budget);   // the dense problems that will ride on conv3_bwd's launch (same rule as the plan)
    double best = 1e30;
    const int rows[3] = {(ft[0].M + 31) / 32, (ft[1].M + 31) / 32, (ft[2].M + 31) / 32};   // 32-deep slabs of each reduction
    auto cands = [](int r) {   // splits that give chunks of 12 .. 32 slabs
      std::vector<int> v;
      int last = -1;
      for (int per = 32; per >= 12; --per) {
        const int s = std::max(1, (r + per - 1) / per);
        if (s != last) v.push_back(s);
        last = s;
      }
      return v;
    };
    for (int s1 : cands(rows[0]))
      for (int s2 : cands(rows[1]))
        for (int s3 : cands(rows[2])) {
          std::vector<IgemmProb> pr;
          double slab_bytes = 0;
          const int sp[3] = {s1, s2, s3};
          for (int l = 2; l >= 0; --l) {
            const int reps = (l == 0 && n_side > 1) ? 1 : 2;
            const int N = (l == 0 && n_side > 1) ? n_side * cg[l].Cout : cg[l].Cout;
            for (int n = 0; n < reps; ++n) {
              pr.push_back(shape(cg[l].K() + 1, N, ft[l].M, sp[l]));
              slab_bytes += 4.0 * pr.back().M * N * pr.back().split;
            }
          }
          pr.insert(pr.end(), light.begin(), light.end());
          double mx = 0;
          xcd_order(tile_list(pr, true, 64, 64), pr, 64, 64, &mx);
          // us: the fullest CU's slabs + the reduction pass: slab bytes written and read back, and its per-thread chain of
          // load batches, which grows with the largest split (measured 13.5 / 15.0 / 19.3 us at 57 / 86-113 / 225 splits)
          const double cost = 0.48 * mx + 2.0 * slab_bytes / 4e6 + 0.02 * std::max(s1, std::max(s2, s3));
          if (cost < best) { best = cost; wsplit[0] = s1; wsplit[1] = s2; wsplit[2] = s3; }
        }
    if (getenv("GRL_PLAN_DUMP"))
      fprintf(stderr, "grl plan: weight-gradient reduction splits %d / %d / %d (model cost %.1f us)\n", wsplit[0], wsplit[1], wsplit[2], best);
  }

  // tiles a launch of these problems will have, and its workgroup shape (mirrors add_launch; -1: not on igemm2_kernel)
  static int planned_tiles(const std::vector<IgemmProb>& probs, int variant, const std::string& tag, int* cfg_out) {
    const char* nv2 = getenv("GRL_NO_V2");
    bool v2 = !(nv2 && nv2[0] == '1');
    for (auto& p : probs) v2 = v2 && v2_prob_ok(p, variant) && (p.K % 4) == 0 && p.p_ones_i < 0;
    if (!v2 || probs.empty()) return -1;
    const int cfg = v2_pick_cfg(probs, variant, tag);
    if (cfg_out) *cfg_out = cfg;
    return (int)tile_list(probs, true, i2_bm(cfg), i2_bn(cfg)).size();
  }
  // Slots the last dispatch round of a launch leaves empty (workgroup b lands on XCD b % 8, CU (b / 8) % 32: a launch of
  // T tiles fills T / 256 rounds completely and T % 256 CUs once more).  Only for launches a CU holds completely.
  static int free_slots(int tiles, int max_per_cu) {
    if (tiles <= 0) return 0;
    const int rounds = (tiles + 255) / 256;
    return rounds <= max_per_cu ? rounds * 256 - tiles : 0;
  }
  // problems of `from` (in order) whose 64x64 tiles fit into `budget` slots; they are removed from `from`
  static std::vector<IgemmProb> take_riders(std::vector<IgemmProb>& from, int budget) {
    std::vector<IgemmProb> out, rest;
    for (auto& p : from) {
      const int Mt = p.p_ones_i >= 0 ? p.M - 1 : p.M;
      const int t = p.split * ((Mt + 63) / 64) * ((p.N + 63) / 64);
      if (t <= budget) { out.push_back(p); budget -= t; } else rest.push_back(p);
    }
    from = rest;
    return out;
  }

  // One descriptor COPY per workgroup, in work-list order.  A workgroup's first operand load sits at the end of a chain
  // of dependent memory round trips (~1.2 us each on MI355X): work-list entry -> problem descriptor -> address tables ->
  // data.  Indexing the descriptors by blockIdx like the work list itself lets the first two travel together.
  static std::vector<IgemmProb> per_tile_descs(const std::vector<IgemmProb>& probs, const std::vector<int4>& tiles) {
    std::vector<IgemmProb> out;
    out.reserve(tiles.size());
    for (const int4& t : tiles) out.push_back(probs[t.x]);
    return out;
  }

  // XCD-aware order of a weight-gradient work list.  The tiles of one reduction chunk read the same rows of the output
  // gradient and overlapping input patches (conv2: 8 row tiles over one 768-pixel chunk; the dense layer: every tile of
  // a row band reads the whole batch of d feat), but the 8 XCDs have private L2s and workgroup b runs on XCD b % 8
  // (MI355X_MICROARCH.md, "Workgroup dispatch"; an observation, used for speed only): in list order those tiles land on
  // 8 different L2s and each fetches its operands from HBM -- 158 MB per launch where ~50 MB are distinct.  Here the
  // tiles are grouped by chunk, the groups dealt to 8 queues (least work first, heavy groups first) and the list is
  // re-emitted so that position i comes from queue i % 8.  Same tiles, same arithmetic.
  static std::vector<int4> xcd_order(const std::vector<int4>& tiles, const std::vector<IgemmProb>& probs, int BMt, int BNt,
                                     double* model_max = nullptr) {
    constexpr int NX = 8;
    struct Grp { std::vector<int4> t; double w = 0; };
    std::vector<Grp> groups;
    std::map<std::tuple<int, int, int>, size_t> at;
    for (const int4& t : tiles) {
      const IgemmProb& p = probs[t.x];
      const int ntj = (p.N + BNt - 1) / BNt;
      const int band = std::max(1, 16 / ntj);                         // row tiles per group: at most ~16 tiles share an L2 set
      const auto key = std::make_tuple(t.x, t.y, t.z / band);
      auto it = at.find(key);
      if (it == at.end()) { it = at.emplace(key, groups.size()).first; groups.emplace_back(); }
      Grp& g = groups[it->second];
      g.t.push_back(t);
      g.w += (double)std::min(p.K, p.k_chunk);
    }
    (void)BMt;
    std::vector<std::vector<int4>> q(NX);
    double load[NX] = {0};
    for (const Grp& g : groups) {                                      // groups arrive heaviest problem first
      int best = 0;
      for (int x = 1; x < NX; ++x)
        if (load[x] < load[best]) best = x;
      q[best].insert(q[best].end(), g.t.begin(), g.t.end());
      load[best] += g.w;
    }
    // Inside an XCD the workgroups go to its 32 CUs round-robin and (at <= 4 workgroups per CU) all at once: list
    // position j of a queue lands on CU j % 32 (scripts/tile_trace.py shows the schedule as the hardware ran it).  A CU
    // then works through the SUM of the reductions it holds, so each queue is arranged longest-processing-time-first
    // over 32 bins -- bin c owns positions c, c + 32, ... -- instead of heavy-first order (57 vs 47 slabs on the
    // fullest CU at the headline shape).
    auto slabs_of = [&](const int4& t) {
      const IgemmProb& p = probs[t.x];
      return (std::min(p.K - t.y * p.k_chunk, p.k_chunk) + 31) / 32;
    };
    for (int x = 0; x < NX; ++x) {
      std::vector<int4>& qx = q[x];
      const int n = (int)qx.size();
      if (n <= 32) continue;
      std::stable_sort(qx.begin(), qx.end(), [&](const int4& a, const int4& b) { return slabs_of(a) > slabs_of(b); });
      std::vector<std::vector<int4>> bin(32);
      int load[32] = {0};
      for (const int4& t : qx) {
        int best = -1;
        for (int c = 0; c < 32; ++c) {
          const int cap = (n - c + 31) / 32;
          if ((int)bin[c].size() >= cap) continue;
          if (best < 0 || load[c] < load[best]) best = c;
        }
        bin[best].push_back(t);
        load[best] += slabs_of(t);
      }
      for (int j = 0; j < n; ++j) qx[j] = bin[j % 32][j / 32];
    }
    std::vector<int4> out;
    out.reserve(tiles.size());
    size_t head[NX] = {0};
    for (size_t i = 0; i < tiles.size(); ++i) {
      int x = (int)(i % NX);
      if (head[x] >= q[x].size()) {                                    // this queue ran dry: the tail loses its affinity
        size_t most = 0;
        for (int y = 0; y < NX; ++y)
          if (q[y].size() - head[y] > most) { most = q[y].size() - head[y]; x = y; }
      }
      out.push_back(q[x][head[x]++]);
    }
    {   // reduction slabs each CU ends up holding if workgroup b lands on XCD b % 8, CU (b / 8) % 32
      double cu[256] = {0}, tot = 0, mx = 0;
      for (size_t i = 0; i < out.size(); ++i) {
        const double w = slabs_of(out[i]);
        cu[(i % 8) * 32 + (i / 8) % 32] += w;
        tot += w;
      }
      for (double c : cu) mx = std::max(mx, c);
      if (model_max) *model_max = mx + (out.size() > 1024 ? 1e6 : 0);   // more than 4 per CU: not all resident, the model does not hold
      else if (getenv("GRL_PLAN_DUMP"))
        fprintf(stderr, "grl plan: weight-gradient list of %zu tiles: %.0f slabs, per CU avg %.1f max %.0f (round-robin placement model)\n",
                out.size(), tot, tot / 256, mx);
    }
    return out;
  }

  // Placement-aware order of a backward-data work list whose tiles differ in length (exact taps: 1 .. 9 tap positions).
  // Same placement model as xcd_order: list position b runs on CU (b % 8) * 32 + (b / 8) % 32, i.e. positions congruent
  // mod 256 share a CU, all of a launch's workgroups are resident at once and a CU works through the SUM of what it
  // holds.  Longest-processing-time-first over 256 bins (bin c owns positions c, c + 256, ...); `extra_tiles` filler
  // tiles of weight `extra_w` will follow at positions n.. (riders of igemm2_pair_kernel) and are counted into their
  // bins beforehand.  Same tiles, same arithmetic.
  int lpt_extra_tiles = 0;
  double lpt_extra_w = 0;
  static std::vector<int4> lpt_order(const std::vector<int4>& tiles, const std::vector<IgemmProb>& probs, int BMt, int BNt,
                                     int extra_tiles, double extra_w, const std::string& tag) {
    const int n = (int)tiles.size();
    const double area = (double)BMt * BNt / 4096.0;
    auto w_of = [&](const int4& t) {
      const IgemmProb& p = probs[t.x];
      return area * ((std::min(p.K - t.y * p.k_chunk, p.k_chunk) + 31) / 32);
    };
    bool uniform = true;
    for (const int4& t : tiles) uniform = uniform && w_of(t) == w_of(tiles[0]);
    if (uniform || n <= 256) return tiles;
    constexpr int NB = 256;
    std::vector<double> load(NB, 0.0);
    for (int k = 0; k < extra_tiles; ++k) load[(n + k) % NB] += extra_w;
    std::vector<int> order(n);
    for (int i = 0; i < n; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return w_of(tiles[a]) > w_of(tiles[b]); });
    std::vector<std::vector<int4>> bin(NB);
    for (int i : order) {
      int best = -1;
      for (int c = 0; c < NB; ++c) {
        const int cap = (n - c + NB - 1) / NB;
        if ((int)bin[c].size() >= cap) continue;
        if (best < 0 || load[c] < load[best]) best = c;
      }
      bin[best].push_back(tiles[i]);
      load[best] += w_of(tiles[i]);
    }
    std::vector<int4> out(n);
    for (int j = 0; j < n; ++j) out[j] = bin[j % NB][j / NB];
    if (getenv("GRL_PLAN_DUMP")) {
      double tot = 0, mx = 0;
      for (double c : load) { tot += c; mx = std::max(mx, c); }
      fprintf(stderr, "grl plan: %-14s placement: %d tiles + %d fillers, 64x64-slab units per CU avg %.1f max %.1f\n", tag.c_str(), n,
              extra_tiles, tot / NB, mx);
    }
    return out;
  }

  // weight-gradient problem + the reductions that land its slab in the flat gradient bucket
  // work list of reduce_slabs_kernel over the descriptors `pick` selects: {descriptor, first output}; descriptors
  // whose geometry allows 16-byte accesses are marked (vec) and cut into tiles of 1024 * RS_QUADS outputs, the others of 256
  std::vector<int2> reduce_tiles(std::function<bool(const ReduceDesc&)> pick = nullptr) {
    std::vector<int2> rt;
    for (size_t k = 0; k < reduces.size(); ++k) {
      ReduceDesc& r = reduces[k];
      r.vec = (elem_vec4_built() && r.n % 4 == 0 && r.slab_stride % 4 == 0 && r.row_len % 4 == 0 && r.src_ld % 4 == 0 &&
               (((uintptr_t)r.dst | (uintptr_t)r.src) & 15) == 0) ? 1 : 0;
      if (pick && !pick(r)) continue;
      const int step = r.vec ? 1024 * RS_QUADS : 256;
      for (int st0 = 0; st0 < r.n; st0 += step) rt.push_back(make_int2((int)k, st0));
    }
    return rt;
  }

  void add_wgrad(std::vector<IgemmProb>& probs, IgemmProb p, int64_t w_off, int64_t w_rows_off, int rows,
                 int64_t b_off) {
    probs.push_back(p);
    ReduceDesc r;
    memset(&r, 0, sizeof(r));
    r.src = p.c; r.splits = p.split; r.slab_stride = p.slab_stride;
    r.dst = grads + w_off + w_rows_off * p.N; r.n = rows * p.N;
    reduces.push_back(r);
    if (b_off >= 0) {   // the bias gradient is the ones row of the slab
      ReduceDesc rb = r;
      rb.src = p.c + (int64_t)p.p_ones_i * p.N; rb.dst = grads + b_off; rb.n = p.N;
      reduces.push_back(rb);
    }
  }

  void alloc_head(HeadAct& h, int rows, int n_out, int out_dim) {
    for (int l = 0; l < L; ++l) h.z[l] = wk.f32((int64_t)rows * hid[l]);
    for (int k = 0; k < n_out; ++k) h.out[k] = wk.f32((int64_t)rows * out_dim);
  }
  void alloc_hgrad(HeadGrad& g, int rows, float* g0 = nullptr, int ld0 = 0) {
    for (int l = 0; l < L; ++l) g.g[l] = (l == 0 && g0) ? g0 : wk.f32((int64_t)rows * hid[l]);
    g.ld0 = g0 ? ld0 : hid[0];
  }

  // forward problems of one MLP head (layer l or the output layer)
  IgemmProb head_layer(const MlpP& m, const float* P, const HeadAct& h, int l, const float* x0, int ld0,
                       int K0, const float* x1, int ld1, int K1, int rows) {
    if (l == 0)
      return dense_fwd(x0, ld0, K0, x1, ld1, K1, rows, P + m.w[0], hid[0], P + m.b[0], h.z[0], hid[0], ACT_RELU);
    return dense_fwd(h.z[l - 1], hid[l - 1], hid[l - 1], nullptr, 0, 0, rows, P + m.w[l], hid[l], P + m.b[l],
                     h.z[l], hid[l], ACT_RELU);
  }
  IgemmProb head_out(const MlpP& m, const float* P, const HeadAct& h, int k, int rows) {
    return dense_fwd(h.z[L - 1], hid[L - 1], hid[L - 1], nullptr, 0, 0, rows, P + m.ow[k], m.out_dim,
                     P + m.ob[k], h.out[k], m.out_dim, ACT_NONE);
  }

  int plan();          // lays everything out in the arenas and builds the launch plans
  int plan_sac();
  int plan_q();        // DQN / BDQ (MLP towers, dueling, double-Q)
  int plan_ae();       // depth auto-encoder training (encoders.py:40-50,70-136)
  float* ae_x = nullptr;            // [B, 4096] minibatch of depth images (staged per step)
  float* ae_out = nullptr;          // [B, 4096] reconstruction of that minibatch
  std::vector<Op> ops_ae_fwd;       // forward half of ops_ae (grl_ae_reconstruct)
  std::vector<Op> ops_ae;
  // Q-learning state
  PerArgs per;                      // prioritised replay (cfg.q_per): device arrays + kernel arguments
  bool per_on = false;
  int per_blocks = 0;
  double* per_u = nullptr;
  std::vector<Op> ops_per_rng, ops_per_u, ops_per_update;
  std::vector<Op> ops_per_rng_g, ops_per_u_g;   // ... the sampler launches that also gather their rows (fully fused PER update)
  std::vector<Op> ops_per_rng_g_inc;            // ... without the block-sum pass over the ring, and the update whose apply launch keeps the
  std::vector<Op> ops_grads_apply_per_r;        //     block sums current instead (updates 2 .. n of one call)
  int qD = 0, qN = 0;
  float *q_td = nullptr, *q_prio = nullptr, *q_aout = nullptr;
  // pinned host staging of the per-env-step calls (grl_act / grl_encode): pageable copies cost more than the kernels
  float *pin_in = nullptr, *pin_out = nullptr;
  char* pin_stats[2] = {nullptr, nullptr};          // page-locked mirrors of the VecNormalize statistics span
  hipEvent_t pin_stats_ev[2] = {nullptr, nullptr};
  bool pin_stats_used[2] = {false, false};
  int pin_stats_next = 0;
  size_t pin_in_n = 0, pin_out_n = 0;
  int act_rows = 0;   // rows the act-path output kernel covers (the launches are sized for act_batch: one static graph)
  int64_t q_online_off = 0, q_online_n = 0;
  int run_ops(std::vector<Op>& ops);
  int capture(std::vector<std::vector<Op>*> seq, hipGraphExec_t* out);
};

static ConvGeom cnn_geom(int l, int C_img) {
  ConvGeom g;
  if (l == 0) g = {64, 64, C_img, 8, 8, 4, 0, 15, 15, 32};
  else if (l == 1) g = {15, 15, 32, 4, 4, 2, 0, 6, 6, 64};
  else g = {6, 6, 64, 3, 3, 1, 0, 4, 4, 64};
  return g;
}

// --------------------------------------------------------------------------------------------------
int grl_ctx::plan() {
  if (cfg.algo == GRL_ALGO_SAC) return plan_sac();
  if (cfg.algo == GRL_ALGO_AE) return plan_ae();
  return plan_q();
}

#include "plan_sac.inl"
#include "plan_q.inl"
#include "plan_ae.inl"

int grl_ctx::run_ops(std::vector<Op>& ops) {
  if (!prof) {
    for (auto& op : ops) op.run(stream);
    return GRL_OK;
  }
  while (ev.size() < 2 * ops.size()) {
    hipEvent_t e;
    HIPCHK(hipEventCreate(&e));
    ev.push_back(e);
  }
  for (size_t i = 0; i < ops.size(); ++i) {
    HIPCHK(hipEventRecord(ev[2 * i], stream));
    ops[i].run(stream);
    HIPCHK(hipEventRecord(ev[2 * i + 1], stream));
  }
  HIPCHK(hipStreamSynchronize(stream));
  for (size_t i = 0; i < ops.size(); ++i) {
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]));
    ProfAcc& a = prof_acc[ops[i].tag];
    a.ms += ms; a.n += 1; a.flops += ops[i].flops; a.bytes += ops[i].bytes;
    a.flops_exec += ops[i].flops_exec > 0 ? ops[i].flops_exec : ops[i].flops;
  }
  return GRL_OK;
}

int grl_ctx::capture(std::vector<std::vector<Op>*> seq, hipGraphExec_t* out) {
  hipGraph_t g;
  if (!side) HIPCHK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
  size_t nev = 0;
  auto next_event = [&](hipEvent_t* e) -> hipError_t {
    if (nev == lane_ev.size()) {
      hipEvent_t x;
      hipError_t r = hipEventCreateWithFlags(&x, hipEventDisableTiming);
      if (r != hipSuccess) return r;
      lane_ev.push_back(x);
    }
    *e = lane_ev[nev++];
    return hipSuccess;
  };
  HIPCHK(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
  bool side_open = false;   // the side lane carries work that has not been joined yet
  for (auto* ops : seq)
    for (auto& op : *ops) {
      if (op.lane == 1) {
        if (op.fork) {
          hipEvent_t e;
          HIPCHK(next_event(&e));
          HIPCHK(hipEventRecord(e, stream));
          HIPCHK(hipStreamWaitEvent(side, e, 0));
        }
        op.run(side);
        side_open = true;
      } else {
        if (op.join && side_open) {
          hipEvent_t e;
          HIPCHK(next_event(&e));
          HIPCHK(hipEventRecord(e, side));
          HIPCHK(hipStreamWaitEvent(stream, e, 0));
          side_open = false;
        }
        op.run(stream);
      }
    }
  if (side_open) {   // every forked lane must be joined before the capture ends
    hipEvent_t e;
    HIPCHK(next_event(&e));
    HIPCHK(hipEventRecord(e, side));
    HIPCHK(hipStreamWaitEvent(stream, e, 0));
  }
  HIPCHK(hipStreamEndCapture(stream, &g));
  HIPCHK(hipGraphInstantiate(out, g, nullptr, nullptr, 0));
  HIPCHK(hipGraphDestroy(g));
  return GRL_OK;
}

#include "capi.inl"
