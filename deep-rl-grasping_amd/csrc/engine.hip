// engine.hip -- host side of libgrl.so: parameter layout, addressing tables, the launch plan of one
// SAC update and the C ABI of include/grl.h.
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
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <map>
#include <string>
#include <vector>

#include "../../include/grl.h"
#include "elem_kernels.h"
#include "igemm.h"
#include "igemm2.h"
#include "heads_kernels.h"
#include "heads_mfma.h"
#include "per_kernels.h"
#include "ae_kernels.h"
#include "q_kernels.h"
#include "q_apply_kernels.h"
#include "igemm_sk.h"
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
  int32_t* d_pre = nullptr;     // per-tile preambles (igemm2.h, I2F_PRE): the table entries a tile needs before its first load
  int n_tiles = 0;
  Launch* filler = nullptr;     // tiles of a second instantiation carried by the same launch (igemm2_pair_kernel)
  unsigned dyn_lds = 0;         // extra LDS bytes requested per workgroup: caps the workgroups a CU holds at once
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
  std::vector<Op> ops_act_norm;   // the act path with VecNormalize applied to raw observations by the ingest launch
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
  int ld1 = 32;   // pixel stride of a1 / g1 (64: the two trained networks side by side)
  float* act_p = nullptr;            // [B, Ap] row-padded copy of the minibatch actions (Ap = rup(A, 4))
  int Ap = 0, ld_d = 1, ld_dm = 0;   // strides of the packed output-gradient buffers (fused heads)
  bool fused_heads = false;          // heads_kernels.h path (row-local chains) instead of per-layer GEMMs
  bool exact_tap = true;             // conv backward-data over exactly the taps that exist (conv_bwd_tabs_exact)
  bool heads_mfma = false;           // heads_mfma.h: forward + backward of all heads as ONE launch on 16x16x4 MFMAs
  float *u_l0[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};   // layer-0 feature partials: pi, vf, qf1, qf2, target
  int l0_split = 1;
  float* g0cat = nullptr;            // [B, 3*H0]: layer-0 gradients of vf | qf1 | qf2
  // act path
  float *ax, *aa1, *aa2, *aa3, *afeat, *a_eps, *a_out;
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
  bool use_lanes = false;
  LossArgs loss_args;              // SAC: batch reductions appended to the reduce_slabs launch (fused heads)
  std::vector<Op> ops_grads_apply; // SAC: ops_grads with Adam + Polyak fused into the slab-reduction launch (full updates)
  // SAC, calls of several updates on the device RNG: the next minibatch is gathered inside the last launch of an update
  std::vector<Op> ops_pf_first, ops_pf_mid, ops_pf_last;
  Op pf_heads[2];
  GatherArgs pf_ga;
  int pf_gx = 0;
  bool prefetch_ok = false;
  std::vector<Op> ops_grads_apply_per;   // DQN / BDQ with prioritised replay: ... and the priority write-back
  // data parallel, staged (grl_compute_grads_staged): stage 0 ends with the dense (fc + head) gradients final in the
  // bucket, stage 1 is the convolution backward + its weight gradients + the loss reductions
  std::vector<Op> ops_stage0, ops_stage1;
  bool staged_ok = false;
  bool loss_in_reduce = false;
  float grad_scale = 1.f;   // read by the apply op

  // data parallel without a host round trip per update (grl_allreduce_*, csrc/dp_kernels.h)
  DpArgs dp;
  bool dp_on = false;
  void* dp_buf = nullptr;                // this rank's exchange buffer (the one device allocation the library makes)
  void* dp_peer[DP_MAX_WORLD] = {nullptr};
  std::vector<Op> ops_dp;                // publish | reduce + push | Adam + Polyak on the exchanged bucket

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
    for (int k = 0; k < 2; ++k) {
      if (pin_stats[k]) hipHostFree(pin_stats[k]);
      if (pin_stats_ev[k]) hipEventDestroy(pin_stats_ev[k]);
    }
    for (int p = 0; p < DP_MAX_WORLD; ++p)
      if (dp_peer[p]) (void)hipIpcCloseMemHandle(dp_peer[p]);
    if (dp_buf) (void)hipFree(dp_buf);
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

  // ---------------------------------------------------------------- helpers
  std::map<const void*, std::vector<int32_t>> htab;   // host copies of the int32 addressing tables (plan time only: preambles)
  template <class T>
  T* upload_vec(Arena& a, const std::vector<T>& v) {
    T* d = (T*)a.take(std::max<size_t>(v.size(), 1) * sizeof(T));
    if constexpr (std::is_same<T, int32_t>::value) htab[(const void*)d] = v;
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
  // split over wave pairs; otherwise 64x64 (shape 2, a 4-way split, stays selectable by GRL_I2CFG_<tag>)
  static int v2_pick_cfg(const std::vector<IgemmProb>& probs, int variant, const std::string& tag) {
    const std::string key = "GRL_I2CFG_" + tag;
    if (const char* e = getenv(key.c_str())) return atoi(e);
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
    return ka == 10030 || ka == 10000 || ka == 12130 || ka == 12110 || ka == 12100 || ka == 11130 || ka == 11110 || ka == 11100 ||
           ka == 11134 || ka == 11114 || ka == 11104;
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
      suppress_pre = true;                                          // (the pair kernel's filler side is one fixed instantiation)
      add_launch(tmp_b, ftag, fvariant, fprobs, "", 0, {}, 0);      // fillers keep the 64x64 shape of the merged launch
      suppress_pre = false;
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
          const dim3 grid(la->n_tiles + lb->n_tiles), block(256);
          const int ka = la->variant * 10000 + la->pm * 1000 + la->qm * 100 + la->cfg * 10 + la->flags;
#define GRL_I2PF(PLv, QLv, PMv, QMv, CF, FLv)                                                                            \
  hipLaunchKernelGGL((igemm2_pair_kernel<PLv, QLv, PMv, QMv, CF, FLv, I2_P_ALONG_I, I2_Q_ALONG_J, PM_TABLE, QM_AFFINE, 0, I2F_ONES>), \
                     grid, block, 0, s, la->d_probs, la->d_tiles, la->n_tiles, lb->d_probs, lb->d_tiles, (const int*)la->d_pre)
#define GRL_I2P(PLv, QLv, PMv, QMv, CF) GRL_I2PF(PLv, QLv, PMv, QMv, CF, 0)
          switch (ka) {
            case 10030: GRL_I2P(I2_P_ALONG_R, I2_Q_ALONG_R, PM_AFFINE, QM_AFFINE, 3); break;
            case 10000: GRL_I2P(I2_P_ALONG_R, I2_Q_ALONG_R, PM_AFFINE, QM_AFFINE, 0); break;
            case 12130: GRL_I2P(I2_P_ALONG_R, I2_Q_ALONG_R, PM_TABLE_MASK, QM_TABLE, 3); break;
            case 12110: GRL_I2P(I2_P_ALONG_R, I2_Q_ALONG_R, PM_TABLE_MASK, QM_TABLE, 1); break;
            case 12100: GRL_I2P(I2_P_ALONG_R, I2_Q_ALONG_R, PM_TABLE_MASK, QM_TABLE, 0); break;
            case 11130: GRL_I2P(I2_P_ALONG_R, I2_Q_ALONG_R, PM_TABLE, QM_TABLE, 3); break;
            case 11110: GRL_I2P(I2_P_ALONG_R, I2_Q_ALONG_R, PM_TABLE, QM_TABLE, 1); break;
            case 11100: GRL_I2P(I2_P_ALONG_R, I2_Q_ALONG_R, PM_TABLE, QM_TABLE, 0); break;
            case 11134: GRL_I2PF(I2_P_ALONG_R, I2_Q_ALONG_R, PM_TABLE, QM_TABLE, 3, I2F_PRE); break;
            case 11114: GRL_I2PF(I2_P_ALONG_R, I2_Q_ALONG_R, PM_TABLE, QM_TABLE, 1, I2F_PRE); break;
            case 11104: GRL_I2PF(I2_P_ALONG_R, I2_Q_ALONG_R, PM_TABLE, QM_TABLE, 0, I2F_PRE); break;
            default:
              fprintf(stderr, "grl: no igemm2 pair instantiation for launch '%s' (key %d)\n", t2.c_str(), ka);
              abort();
          }
#undef GRL_I2P
#undef GRL_I2PF
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
            (!p.c_tab_i || (p.vflags & VF_CT4)) && (!p.relu_mask || al16(p.relu_mask)) && (!p.bias || al16(p.bias)))
          p.vflags |= VF_C_VEC;
    // short reductions with a narrow output (first convolution of the extractors): streaming kernel, igemm_sk.h
    {
      const char* ns = getenv("GRL_NO_SK");
      bool ok = l->v2 && variant == 0 && l->pm == PM_TABLE && l->qm == QM_AFFINE && l->flags == 0 && !(ns && ns[0] == '1');
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
      if (const char* e = getenv("GRL_SK_WGS")) slots = std::max(1, atoi(e));   // tuning aid
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
      op.run = [l](hipStream_t s) {
        if (l->sk == 64) hipLaunchKernelGGL((igemm_sk_kernel<64>), dim3(l->n_tiles), dim3(256), 0, s, l->d_probs, l->d_tiles);
        else hipLaunchKernelGGL((igemm_sk_kernel<32>), dim3(l->n_tiles), dim3(256), 0, s, l->d_probs, l->d_tiles);
      };
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
    if (variant == 2 && l->v2) {
      if (const char* e = getenv("GRL_WG_DYNLDS")) l->dyn_lds = (unsigned)atoi(e);
    }
    l->n_tiles = (int)tiles.size();
    {   // per-tile preambles (I2F_PRE): unmasked table addressing of the three convolution directions.  Opt-in
        // (GRL_PREAMBLE=1): measured neutral on MI355X at B = 256 (5 102 against 5 092 updates/s on one box) -- the 2 - 4 us
        // between a workgroup's start and its first barrier are the operand loads of 400 - 800 workgroups arriving at
        // once, not the table hop in front of them.  Kept as a tested switch (bit-identical results).
      const char* np = getenv("GRL_PREAMBLE");
      bool ok = l->v2 && !suppress_pre && (np && atoi(np)) && l->pm == PM_TABLE &&
                ((variant == 0 && l->qm == QM_AFFINE && l->flags == 0) || (variant == 1 && l->qm == QM_TABLE && l->flags == 0) ||
                 (variant == 2 && l->qm == QM_AFFINE && l->flags == I2F_ONES && l->cfg <= 1));
      for (auto& p : l->probs)
        ok = ok && htab.count(p.p_tab_i) && htab.count(p.p_tab_r) && (!p.q_tab_r || htab.count(p.q_tab_r)) &&
             (!p.c_tab_i || htab.count(p.c_tab_i));
      if (ok) {
        const int BKTt = 32 * (l->cfg == 2 ? 4 : (l->cfg == 3 ? 2 : 1));
        const int ROWo = 0, PKo = BMt, QKo = BMt + 3 * BKTt, CTo = BMt + 6 * BKTt, STR = 2 * BMt + 6 * BKTt;
        std::vector<int32_t> pre((size_t)tiles.size() * STR, 0);
        for (size_t ti = 0; ti < tiles.size(); ++ti) {
          const int4& t = tiles[ti];
          const IgemmProb& p = l->probs[t.x];
          const std::vector<int32_t>&hi = htab[p.p_tab_i], &hr = htab[p.p_tab_r];
          const std::vector<int32_t>* hq = p.q_tab_r ? &htab[p.q_tab_r] : nullptr;
          const std::vector<int32_t>* hc = p.c_tab_i ? &htab[p.c_tab_i] : nullptr;
          const int Meff = p.p_ones_i >= 0 ? p.M - 1 : p.M;
          const int i0 = t.z * BMt, r_begin = t.y * p.k_chunk, r_end = std::min(p.K, r_begin + p.k_chunk);
          int32_t* o = pre.data() + ti * STR;
          for (int x = 0; x < BMt; ++x) {
            const int i = i0 + x;
            o[ROWo + x] = hi[(size_t)(i < Meff ? i : i0)];
            if (hc) o[CTo + x] = (*hc)[(size_t)(i < Meff ? i : 0)];
          }
          for (int k = 0; k < 3 * BKTt; ++k) {
            const int r = r_begin + k, rc = r < r_end ? r : r_begin;
            o[PKo + k] = hr[(size_t)rc];
            if (hq) o[QKo + k] = (*hq)[(size_t)rc];
          }
        }
        l->d_pre = upload_vec(wk, pre);
        htab.erase((const void*)l->d_pre);          // (not an addressing table)
        l->flags |= I2F_PRE;
      }
    }
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
      dim3 grid(l->n_tiles), block(256);
      if (l->v2) {
        const int key = l->variant * 10000 + l->pm * 1000 + l->qm * 100 + l->cfg * 10 + l->flags;
#define GRL_I2(PLv, QLv, PMv, QMv, CF, FL) \
  hipLaunchKernelGGL((igemm2_kernel<PLv, QLv, PMv, QMv, CF, FL>), grid, block, l->dyn_lds, s, l->d_probs, l->d_tiles, (const int*)l->d_pre)
#define GRL_I2_CFGS(base, PLv, QLv, PMv, QMv, FL)                           \
  case base + 0 + FL: GRL_I2(PLv, QLv, PMv, QMv, 0, FL); break;              \
  case base + 10 + FL: GRL_I2(PLv, QLv, PMv, QMv, 1, FL); break;             \
  case base + 20 + FL: GRL_I2(PLv, QLv, PMv, QMv, 2, FL); break;             \
  case base + 30 + FL: GRL_I2(PLv, QLv, PMv, QMv, 3, FL); break;
        switch (key) {
          GRL_I2_CFGS(0, I2_P_ALONG_R, I2_Q_ALONG_J, PM_AFFINE, QM_AFFINE, 0)            // dense forward
          GRL_I2_CFGS(0, I2_P_ALONG_R, I2_Q_ALONG_J, PM_AFFINE, QM_AFFINE, I2F_KTAIL)    //   ... K % 4 != 0
          GRL_I2_CFGS(1000, I2_P_ALONG_R, I2_Q_ALONG_J, PM_TABLE, QM_AFFINE, 0)          // VALID conv forward
          GRL_I2_CFGS(2000, I2_P_ALONG_R, I2_Q_ALONG_J, PM_TABLE_MASK, QM_AFFINE, 0)     // padded conv forward
          GRL_I2_CFGS(10000, I2_P_ALONG_R, I2_Q_ALONG_R, PM_AFFINE, QM_AFFINE, 0)        // dense backward-data
          GRL_I2_CFGS(12100, I2_P_ALONG_R, I2_Q_ALONG_R, PM_TABLE_MASK, QM_TABLE, 0)     // conv backward-data, masked taps
          GRL_I2_CFGS(11100, I2_P_ALONG_R, I2_Q_ALONG_R, PM_TABLE, QM_TABLE, 0)          // conv backward-data, exact taps
          GRL_I2_CFGS(1000, I2_P_ALONG_R, I2_Q_ALONG_J, PM_TABLE, QM_AFFINE, I2F_PRE)    // ... the same with per-tile preambles
          GRL_I2_CFGS(11100, I2_P_ALONG_R, I2_Q_ALONG_R, PM_TABLE, QM_TABLE, I2F_PRE)
          case 21005: GRL_I2(I2_P_ALONG_I, I2_Q_ALONG_J, PM_TABLE, QM_AFFINE, 0, I2F_ONES | I2F_PRE); break;
          case 21015: GRL_I2(I2_P_ALONG_I, I2_Q_ALONG_J, PM_TABLE, QM_AFFINE, 1, I2F_ONES | I2F_PRE); break;
          GRL_I2_CFGS(10100, I2_P_ALONG_R, I2_Q_ALONG_R, PM_AFFINE, QM_TABLE, 0)         // dense backward-data over several kernels
          case 20000: GRL_I2(I2_P_ALONG_I, I2_Q_ALONG_J, PM_AFFINE, QM_AFFINE, 0, 0); break;         // dense weight gradient
          case 20001: GRL_I2(I2_P_ALONG_I, I2_Q_ALONG_J, PM_AFFINE, QM_AFFINE, 0, I2F_ONES); break;  //   ... with bias row
          case 20010: GRL_I2(I2_P_ALONG_I, I2_Q_ALONG_J, PM_AFFINE, QM_AFFINE, 1, 0); break;
          case 20011: GRL_I2(I2_P_ALONG_I, I2_Q_ALONG_J, PM_AFFINE, QM_AFFINE, 1, I2F_ONES); break;
          case 21000: GRL_I2(I2_P_ALONG_I, I2_Q_ALONG_J, PM_TABLE, QM_AFFINE, 0, 0); break;          // conv weight gradient
          case 21001: GRL_I2(I2_P_ALONG_I, I2_Q_ALONG_J, PM_TABLE, QM_AFFINE, 0, I2F_ONES); break;
          case 21010: GRL_I2(I2_P_ALONG_I, I2_Q_ALONG_J, PM_TABLE, QM_AFFINE, 1, 0); break;
          case 21011: GRL_I2(I2_P_ALONG_I, I2_Q_ALONG_J, PM_TABLE, QM_AFFINE, 1, I2F_ONES); break;
          default:
            fprintf(stderr, "grl: no igemm2 instantiation for launch '%s' (key %d)\n", tag.c_str(), key);
            abort();
        }
#undef GRL_I2_CFGS
#undef GRL_I2
        return;
      }
      const int key = l->np * 1000 + l->pm * 100 + l->qm * 10 + l->variant;
#define GRL_IGEMM(PMv, QMv, PR, QJ, NPv) \
  hipLaunchKernelGGL((igemm_kernel<PMv, QMv, PR, QJ, NPv>), grid, block, 0, s, l->d_probs, l->d_tiles)
      switch (key) {
        case 1000: GRL_IGEMM(PM_AFFINE, QM_AFFINE, true, true, 1); break;        // dense forward
        case 3000: GRL_IGEMM(PM_AFFINE, QM_AFFINE, true, true, 3); break;        //   ... concatenated input
        case 1100: GRL_IGEMM(PM_TABLE, QM_AFFINE, true, true, 1); break;         // VALID conv forward
        case 1200: GRL_IGEMM(PM_TABLE_MASK, QM_AFFINE, true, true, 1); break;    // padded conv forward
        case 1001: GRL_IGEMM(PM_AFFINE, QM_AFFINE, true, false, 1); break;       // dense backward-data
        case 3001: GRL_IGEMM(PM_AFFINE, QM_AFFINE, true, false, 3); break;       //   ... summed over heads
        case 1211: GRL_IGEMM(PM_TABLE_MASK, QM_TABLE, true, false, 1); break;    // conv backward-data
        case 1111: GRL_IGEMM(PM_TABLE, QM_TABLE, true, false, 1); break;         //   ... over exact taps
        case 1011: GRL_IGEMM(PM_AFFINE, QM_TABLE, true, false, 1); break;        // dense backward-data over several kernels
        case 1002: GRL_IGEMM(PM_AFFINE, QM_AFFINE, false, true, 1); break;       // dense weight gradient
        case 1102: GRL_IGEMM(PM_TABLE, QM_AFFINE, false, true, 1); break;        // conv weight gradient
        case 1202: GRL_IGEMM(PM_TABLE_MASK, QM_AFFINE, false, true, 1); break;   //   ... of a padded conv
        default:
          fprintf(stderr, "grl: no igemm instantiation for launch '%s' (key %d)\n", tag.c_str(), key);
          abort();
      }
#undef GRL_IGEMM
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
  // re-emitted so that position i comes from queue i % 8.  Same tiles, same arithmetic; GRL_NO_XCD_ORDER=1 keeps the
  // list order (test / measurement switch).
  static std::vector<int4> xcd_order(const std::vector<int4>& tiles, const std::vector<IgemmProb>& probs, int BMt, int BNt,
                                     double* model_max = nullptr) {
    if (const char* e = getenv("GRL_NO_XCD_ORDER")) if (atoi(e) && !model_max) return tiles;   // (the split chooser still models the ordered list)
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
  // bins beforehand.  Same tiles, same arithmetic; GRL_NO_LPT_ORDER=1 keeps the list order (measurement switch).
  int lpt_extra_tiles = 0;
  double lpt_extra_w = 0;
  bool suppress_pre = false;
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
    if (const char* e = getenv("GRL_NO_LPT_ORDER")) if (atoi(e)) uniform = true;
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
  // whose geometry allows 16-byte accesses are marked (vec) and cut into tiles of 1024 outputs, the others of 256
  std::vector<int2> reduce_tiles(std::function<bool(const ReduceDesc&)> pick = nullptr) {
    std::vector<int2> rt;
    for (size_t k = 0; k < reduces.size(); ++k) {
      ReduceDesc& r = reduces[k];
#ifdef GRL_HOSTEMU
      r.vec = 0;
#else
      const char* nv = getenv("GRL_NO_VEC_REDUCE");   // test switch: the one-output-per-thread form everywhere
      r.vec = (!(nv && atoi(nv)) && r.n % 4 == 0 && r.slab_stride % 4 == 0 && r.row_len % 4 == 0 && r.src_ld % 4 == 0 &&
               (((uintptr_t)r.dst | (uintptr_t)r.src) & 15) == 0) ? 1 : 0;
#endif
      if (pick && !pick(r)) continue;
      const int step = r.vec ? 1024 : 256;
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

int grl_ctx::plan_sac() {
  const grl_config& c = cfg;
  cnn = c.extractor != GRL_EXTRACTOR_MLP;
  A = c.act_dim; L = c.n_layers; B = c.batch_size; NA = std::max(1, c.act_batch);
  for (int l = 0; l < L; ++l) hid[l] = c.layers[l];
  hw = c.img_hw;
  if (cnn) {
    const int nd = c.extractor == GRL_EXTRACTOR_AUGMENTED ? c.n_direct : 0;
    C_img = c.obs_channels - (nd > 0 ? 1 : 0);
    img_elems = hw * hw * C_img;
    F = 512 + nd; Fc = 512;
  } else {
    C_img = 0; img_elems = c.obs_dim; F = c.obs_dim; Fc = 0;
  }
  const int nd = cnn ? F - 512 : 0;
  ldf = (int)rup(F, 4);
  build_layout();

  // ---------------- state arena
  params = st.f32(n_params);
  adam_m = st.f32(n_train);
  adam_v = st.f32(n_train);
  sc = (DevScalars*)st.take(sizeof(DevScalars));
  n_count = (double*)st.take(16);
  s_mean = (double*)st.take((size_t)img_elems * 8);
  s_std = (double*)st.take((size_t)img_elems * 8);
  s_dmean = (double*)st.take((size_t)std::max(nd, 1) * 8);
  s_dstd = (double*)st.take((size_t)std::max(nd, 1) * 8);
  s_ret = (double*)st.take(8);
  n_elems = cnn ? (int64_t)hw * hw * c.obs_channels : c.obs_dim;
  n_mean = (double*)st.take((size_t)n_elems * 8);     // (directly behind s_ret: grl_set_obs_stats uploads the span in one copy)
  n_var = (double*)st.take((size_t)n_elems * 8);
  grads = gr.f32(n_train);

  // ---------------- replay arena
  const int64_t cap = c.replay_capacity;
  // stored observation: img_elems floats, or (replay_rgb_u8) one packed colour dword + one depth float per pixel
  const int64_t obs_store = c.replay_rgb_u8 ? 2 * (int64_t)hw * hw : img_elems;
  rp_obs = rp.f32(cap * obs_store);
  rp_next = rp.f32(cap * obs_store);
  rp_dobs = rp.f32(cap * std::max(nd, 1));
  rp_dnext = rp.f32(cap * std::max(nd, 1));
  rp_act = rp.f32(cap * A);
  rp_rew = rp.f32(cap);
  rp_done = rp.f32(cap);

  // ---------------- staging (host-facing calls)
  const int64_t obs_elems = cnn ? (int64_t)hw * hw * c.obs_channels : c.obs_dim;
  stg_n = std::max(NA, 64);
  stg_obs = wk.f32(stg_n * obs_elems);
  stg_next = wk.f32(stg_n * obs_elems);
  stg_act = wk.f32((int64_t)stg_n * A);
  stg_rew = wk.f32(stg_n);
  stg_done = wk.f32(stg_n);
  n_stage = wk.f32(stg_n * obs_elems);

  // ---------------- training workspace
  idx_buf = (int64_t*)wk.take((size_t)B * 8);
  eps_buf = wk.f32((int64_t)B * A);
  for (int n = 0; n < 3; ++n) {
    feat[n] = wk.f32((int64_t)B * ldf);
    zero_once.push_back({feat[n], (size_t)B * ldf * 4});   // row padding [F, ldf) is read by 16-byte loads
  }
  if (cnn) {
    x_obs = wk.f32((int64_t)B * img_elems);
    x_next = wk.f32((int64_t)B * img_elems);
    // Layer-1 activations of the two TRAINED networks (and their gradients below) sit side by side, pixel stride 64:
    // pi in columns 0..31, values_fn in 32..63.  Both networks read the same observations, so conv1's weight
    // gradient becomes ONE product obs-patches^T x [dY_pi | dY_vf] (N = 64: full 64x64 tiles, the gathered patches
    // read once) instead of two half-empty ones.  The target network's buffer keeps the stride (columns 32..63 idle)
    // so that one set of conv2 tables serves all three.  GRL_NO_CONV1_SIDE=1: separate dense buffers (test switch).
    {
      const char* ns = getenv("GRL_NO_CONV1_SIDE");
      ld1 = (ns && atoi(ns)) ? 32 : 64;
    }
    float* a1_pair = ld1 == 64 ? wk.f32((int64_t)B * 225 * 64) : nullptr;
    for (int n = 0; n < 3; ++n) {
      a1[n] = ld1 == 32 ? wk.f32((int64_t)B * 225 * 32) : (n < 2 ? a1_pair + 32 * n : wk.f32((int64_t)B * 225 * 64));
      a2[n] = wk.f32((int64_t)B * 36 * 64);
      a3[n] = wk.f32((int64_t)B * 16 * 64);
    }
  }
  act = wk.f32((int64_t)B * A); rew = wk.f32(B); done = wk.f32(B);
  alloc_head(hPI, B, 2, A); alloc_head(hVF, B, 1, 1); alloc_head(hQF1, B, 1, 1); alloc_head(hQF2, B, 1, 1);
  alloc_head(hTGT, B, 1, 1); alloc_head(hQF1PI, B, 1, 1); alloc_head(hQF2PI, B, 1, 1);
  pi_a = wk.f32((int64_t)B * A); logp = wk.f32(B); ent = wk.f32(B);
  {
    const char* nf = getenv("GRL_NO_FUSED_HEADS");
    fused_heads = !(nf && nf[0] == '1') && A <= HT_MAXA && (hid[0] % 4) == 0;
    for (int l = 0; l < L; ++l) fused_heads = fused_heads && hid[l] <= HT_MAXW;
  }
  Ap = (int)rup(A, 4);
  if (fused_heads) {
    // output gradients packed / row-padded so that the weight-gradient GEMM can fetch them 16 bytes at a time
    float** dd[4] = {&d_v, &d_qf1, &d_qf2, &d_qf1pi};
    for (auto* q : dd) {
      *q = wk.f32((int64_t)B * 4);
      zero_once.push_back({*q, (size_t)B * 16});
    }
    ld_d = 4;
    act_p = wk.f32((int64_t)B * Ap);
    zero_once.push_back({act_p, (size_t)B * Ap * 4});
  } else {
    d_qf1 = wk.f32(B); d_qf2 = wk.f32(B); d_v = wk.f32(B); d_qf1pi = wk.f32(B); ld_d = 1;
  }
  if (fused_heads) {
    {
      const char* e = getenv("GRL_L0_SPLIT");
      l0_split = e ? std::max(1, atoi(e)) : 3;       // reduction of the layer-0 GEMM (K = 513) cut into partial sums
      IgemmProb probe = blank();
      probe.M = B; probe.N = hid[0]; probe.K = F;
      set_split(probe, l0_split);
      l0_split = probe.split;
    }
    for (int k = 0; k < 5; ++k) u_l0[k] = wk.f32((int64_t)B * hid[0] * l0_split);
    g0cat = wk.f32((int64_t)B * 3 * hid[0]);
    alloc_hgrad(gPI, B);
    alloc_hgrad(gVF, B, g0cat, 3 * hid[0]);
    alloc_hgrad(gQF1, B, g0cat + hid[0], 3 * hid[0]);
    alloc_hgrad(gQF2, B, g0cat + 2 * hid[0], 3 * hid[0]);
    alloc_hgrad(gQF1PI, B);
  } else {
    alloc_hgrad(gPI, B); alloc_hgrad(gVF, B); alloc_hgrad(gQF1, B); alloc_hgrad(gQF2, B); alloc_hgrad(gQF1PI, B);
  }
  ld_dm = fused_heads ? Ap : A;
  da_pi = wk.f32((int64_t)B * A); dmu = wk.f32((int64_t)B * ld_dm); dls = wk.f32((int64_t)B * ld_dm);
  if (fused_heads) {
    zero_once.push_back({dmu, (size_t)B * ld_dm * 4});
    zero_once.push_back({dls, (size_t)B * ld_dm * 4});
  }
  if (cnn) {
    float* g1_pair = ld1 == 64 ? wk.f32((int64_t)B * 225 * 64) : nullptr;
    for (int n = 0; n < 2; ++n) {
      dfeat[n] = wk.f32((int64_t)B * ldf);
      g3[n] = wk.f32((int64_t)B * 16 * 64);
      g2[n] = wk.f32((int64_t)B * 36 * 64);
      g1[n] = ld1 == 64 ? g1_pair + 32 * n : wk.f32((int64_t)B * 225 * 32);
    }
  }

  const float* P = params;
  const float* T = params;   // target block uses absolute offsets too

  // =============================================================== minibatch: replay gather (+ device RNG)
  {
    GatherArgs ga;
    memset(&ga, 0, sizeof(ga));
    ga.idx = idx_buf; ga.B = B; ga.img_elems = img_elems; ga.n_direct = nd; ga.act_dim = A;
    ga.rp_obs = rp_obs; ga.rp_next = rp_next; ga.rp_dobs = rp_dobs; ga.rp_dnext = rp_dnext;
    ga.rp_act = rp_act; ga.rp_rew = rp_rew; ga.rp_done = rp_done;
    ga.mean = s_mean; ga.stdv = s_std; ga.dmean = s_dmean; ga.dstd = s_dstd; ga.ret_std = s_ret;
    ga.normalize = (c.normalize == 1 || c.normalize == 2); ga.normalize_rew = (c.normalize == 1 || c.normalize == 3);
    ga.clip_obs = c.clip_obs; ga.clip_rew = c.clip_reward;
    ga.scale_div = cnn ? 255.f : 1.f;
    if (cnn) {
      ga.x_obs = x_obs; ga.x_obs2 = nullptr; ga.x_next = x_next; ga.ldx = img_elems;
      ga.d_obs0 = feat[0] + 512; ga.d_obs1 = feat[1] + 512; ga.d_next = feat[2] + 512; ga.ldd = ldf;
    } else {
      ga.x_obs = feat[0]; ga.x_obs2 = feat[1]; ga.x_next = feat[2]; ga.ldx = ldf;
      ga.d_obs0 = ga.d_obs1 = ga.d_next = feat[0]; ga.ldd = ldf;   // n_direct == 0: never written
    }
    ga.act_out = act; ga.ld_act = A; ga.rew_out = rew; ga.done_out = done;
    ga.act_out2 = act_p; ga.ld_act2 = Ap;
    ga.rgb_u8 = c.replay_rgb_u8;
    ga.sc = sc; ga.seed = c.seed; ga.idx_w = idx_buf; ga.eps_w = eps_buf; ga.n_eps = A;
    ga.adam_tick = fused_heads ? 1 : 0;   // otherwise sac_loss_kernel fixes the step size
#ifndef GRL_HOSTEMU
    ga.vec4 = (img_elems % 4 == 0) && (ga.ldx % 4 == 0);
#endif
    const int per_block = ga.vec4 ? 1024 : 256;
    pf_ga = ga;
    pf_gx = (ga.img_elems + per_block - 1) / per_block;
    for (int mode = 0; mode < 2; ++mode) {
      ga.use_rng = mode;
      Op op; op.tag = "gather_norm";
      op.bytes = 2.0 * B * ((double)img_elems * 4 + (double)obs_store * 4 + 4.0 * nd) + B * (4.0 * A + 8) * 2;
      op.run = [ga, per_block](hipStream_t s) {
        hipLaunchKernelGGL(gather_norm_kernel, dim3((ga.img_elems + per_block - 1) / per_block, ga.B, 2), dim3(256), 0, s, ga);
      };
      (mode ? ops_rng : ops_gather).push_back(op);
    }
  }

  ConvGeom cg[3];
  ConvFwdTabs ft[3];
  if (cnn) {
    for (int l = 0; l < 3; ++l) {
      cg[l] = cnn_geom(l, C_img);
      if (l == 0) cg[l].ldy = ld1;
      if (l == 1) cg[l].ldx = ld1;
      ft[l] = conv_fwd_tabs(cg[l], B);
    }
    const float* xin[3] = {x_obs, x_obs, x_next};
    const char* tags[3] = {"conv1_fwd", "conv2_fwd", "conv3_fwd"};
    for (int l = 0; l < 3; ++l) {
      std::vector<IgemmProb> pr;
      for (int n = 0; n < 3; ++n) {
        const float* in = l == 0 ? xin[n] : (l == 1 ? a1[n] : a2[n]);
        float* out = l == 0 ? a1[n] : (l == 1 ? a2[n] : a3[n]);
        pr.push_back(conv_fwd(in, ft[l], cg[l], P + ex[n].w[l], P + ex[n].b[l], out, ACT_RELU, 0.f));
      }
      add_launch(ops_grads, tags[l], 0, pr);
    }
    std::vector<IgemmProb> pr;
    for (int n = 0; n < 3; ++n)
      pr.push_back(dense_fwd(a3[n], 1024, 1024, nullptr, 0, 0, B, P + ex[n].fw, 512, P + ex[n].fb, feat[n], ldf,
                             ACT_RELU));
    add_launch(ops_grads, "fc_fwd", 0, pr);
  }
  (void)T;

  // description of one head for the row-local kernels (heads_kernels.h)
  auto mk_head = [&](const MlpP& m, const HeadAct& h, const HeadGrad* g, const float* u, const float* xa, int ld_xa,
                     int n_xa) {
    HtHead H;
    memset(&H, 0, sizeof(H));
    H.u = u; H.ldu = hid[0]; H.u_split = l0_split; H.u_stride = (long)B * hid[0];
    H.xa = xa; H.ld_xa = ld_xa; H.n_xa = n_xa;
    H.w0a = P + m.w[0] + (int64_t)F * hid[0];
    H.b0 = P + m.b[0]; H.z0 = h.z[0]; H.H0 = hid[0]; H.L = L;
    for (int l = 0; l < L; ++l) H.hid[l] = hid[l];
    for (int l = 1; l < L; ++l) { H.w[l] = P + m.w[l]; H.b[l] = P + m.b[l]; H.z[l] = h.z[l]; }
    if (g) {
      H.g0 = g->g[0]; H.ldg0 = g->ld0;
      for (int l = 1; l < L; ++l) H.g[l] = g->g[l];
    }
    H.n_out = m.n_out; H.out_dim = m.out_dim;
    for (int k = 0; k < m.n_out; ++k) { H.ow[k] = P + m.ow[k]; H.ob[k] = P + m.ob[k]; H.out[k] = h.out[k]; }
    return H;
  };
  if (fused_heads) {
    // layer 0, feature part only (no bias, no activation): u = feat . W0[0:F]
    const MlpP* ms[5] = {&m_pi, &m_vf, &m_qf1, &m_qf2, &m_tgt};
    const float* fin[5] = {feat[0], feat[1], feat[1], feat[1], feat[2]};
    std::vector<IgemmProb> pr;
    for (int k = 0; k < 5; ++k) {
      IgemmProb p = dense_fwd(fin[k], ldf, F, nullptr, 0, 0, B, P + ms[k]->w[0], hid[0], nullptr, u_l0[k], hid[0], ACT_NONE);
      set_split(p, l0_split);     // partial sums [split][B, H0]: the head chains add them (40 -> 120 workgroups)
      pr.push_back(p);
    }
    add_launch(ops_grads, "heads_l0", 0, pr);
    HeadsFwdArgs fa;
    memset(&fa, 0, sizeof(fa));
    fa.h[0] = mk_head(m_pi, hPI, nullptr, u_l0[0], nullptr, 0, 0);
    fa.h[1] = mk_head(m_vf, hVF, nullptr, u_l0[1], nullptr, 0, 0);
    fa.h[2] = mk_head(m_qf1, hQF1, nullptr, u_l0[2], act, A, A);
    fa.h[3] = mk_head(m_qf2, hQF2, nullptr, u_l0[3], act, A, A);
    fa.h[4] = mk_head(m_tgt, hTGT, nullptr, u_l0[4], nullptr, 0, 0);
    fa.h[5] = mk_head(m_qf1, hQF1PI, nullptr, u_l0[2], pi_a, A, A);
    fa.h[6] = mk_head(m_qf2, hQF2PI, nullptr, u_l0[3], pi_a, A, A);
    fa.B = B; fa.A = A; fa.eps = eps_buf; fa.pi_a = pi_a; fa.logp = logp; fa.ent = ent;
    {
      const char* nm = getenv("GRL_NO_HEADS_MFMA");
      heads_mfma = !(nm && nm[0] == '1') && 2 * A <= 64;      // heads_mfma.h: layer widths (and 2A) up to 64
      for (int l = 0; l < L; ++l) heads_mfma = heads_mfma && hid[l] <= 64;
    }
    if (heads_mfma) {
      // forward and backward of every head in one launch (heads_mfma.h); d_out / gradients as in the backward args below
      HeadsFusedArgs ha;
      memset(&ha, 0, sizeof(ha));
      ha.h[0] = mk_head(m_pi, hPI, &gPI, u_l0[0], nullptr, 0, 0);
      ha.h[1] = mk_head(m_vf, hVF, &gVF, u_l0[1], nullptr, 0, 0);
      ha.h[2] = mk_head(m_qf1, hQF1, &gQF1, u_l0[2], act, A, A);
      ha.h[3] = mk_head(m_qf2, hQF2, &gQF2, u_l0[3], act, A, A);
      ha.h[4] = mk_head(m_tgt, hTGT, nullptr, u_l0[4], nullptr, 0, 0);
      ha.h[5] = mk_head(m_qf1, hQF1PI, nullptr, u_l0[2], pi_a, A, A);
      ha.h[6] = mk_head(m_qf2, hQF2PI, nullptr, u_l0[3], pi_a, A, A);
      ha.B = B; ha.A = A; ha.eps = eps_buf; ha.pi_a = pi_a; ha.logp = logp; ha.ent = ent;
      ha.log_ent_coef = params + ent_off; ha.da_pi = da_pi; ha.dmu = dmu; ha.dls = dls; ha.ld_dm = ld_dm;
      ha.rew = rew; ha.done = done; ha.gamma = c.gamma;
      ha.d_out[1] = d_v; ha.d_out[2] = d_qf1; ha.d_out[3] = d_qf2; ha.d_out[4] = d_qf1pi; ha.ld_d = ld_d;
      if (const char* e = getenv("GRL_HEADS_STAMPS")) {
        if (e[0] == '1') {
          ha.stamps = (unsigned long long*)wk.take(4 * 32 * 8);
          zero_once.push_back({ha.stamps, 4 * 32 * 8});
          dbg["heads_stamps"] = {(const float*)ha.stamps, 4 * 32 * 2};
        }
      }
      // argument blocks: [0] the plain update, [1] / [2] updates of a prefetching multi-update call whose head launch
      // opens the update (Adam step size) and, from the second update on, advances the RNG counter
      HeadsFusedArgs hb = ha, hc = ha;
      hb.sc = hc.sc = sc; hb.tick = hc.tick = 1; hb.rng_advance = 0; hc.rng_advance = 1;
      const HeadsFusedArgs* d_ha = upload_vec(wk, std::vector<HeadsFusedArgs>{ha, hb, hc});
      const int nblk = (B + HT_RB - 1) / HT_RB;
      const bool fast = L == 2 && hid[0] == 64 && hid[1] == 64 && B % HT_RB == 0;   // the reference's layers [64, 64]
      for (int v = 0; v < 3; ++v) {
        Op op; op.tag = "heads";
        const HeadsFusedArgs* dv = d_ha + v;
        op.run = [dv, nblk, fast](hipStream_t s) {
          if (fast) hipLaunchKernelGGL((heads_fused_kernel<64, true>), dim3(nblk, 4), dim3(256), 0, s, dv);
          else hipLaunchKernelGGL((heads_fused_kernel<64, false>), dim3(nblk, 4), dim3(256), 0, s, dv);
        };
        if (v == 0) ops_grads.push_back(op);
        else pf_heads[v - 1] = op;
      }
    } else {
    Op op; op.tag = "heads_fwd";
    op.run = [fa](hipStream_t s) {
      hipLaunchKernelGGL(heads_fwd_kernel, dim3((fa.B + HT_RB - 1) / HT_RB, 6), dim3(256), 0, s, fa);
    };
    ops_grads.push_back(op);
    }
  } else {
    // heads forward: pi, vf, qf1, qf2 (data action), target vf
    for (int l = 0; l < L; ++l) {
      std::vector<IgemmProb> pr;
      pr.push_back(head_layer(m_pi, P, hPI, l, feat[0], ldf, F, nullptr, 0, 0, B));
      pr.push_back(head_layer(m_vf, P, hVF, l, feat[1], ldf, F, nullptr, 0, 0, B));
      pr.push_back(head_layer(m_qf1, P, hQF1, l, feat[1], ldf, F, act, A, A, B));
      pr.push_back(head_layer(m_qf2, P, hQF2, l, feat[1], ldf, F, act, A, A, B));
      pr.push_back(head_layer(m_tgt, P, hTGT, l, feat[2], ldf, F, nullptr, 0, 0, B));
      add_launch(ops_grads, "heads_fwd", 0, pr);
    }
    {
      std::vector<IgemmProb> pr;
      pr.push_back(head_out(m_pi, P, hPI, 0, B));
      pr.push_back(head_out(m_pi, P, hPI, 1, B));
      pr.push_back(head_out(m_vf, P, hVF, 0, B));
      pr.push_back(head_out(m_qf1, P, hQF1, 0, B));
      pr.push_back(head_out(m_qf2, P, hQF2, 0, B));
      pr.push_back(head_out(m_tgt, P, hTGT, 0, B));
      add_launch(ops_grads, "heads_fwd", 0, pr);
    }
    {
      SampleArgs sa{hPI.out[0], hPI.out[1], eps_buf, B, A, pi_a, nullptr, logp, ent};
      Op op; op.tag = "sample";
      op.run = [sa](hipStream_t s) {
        hipLaunchKernelGGL(sample_kernel, dim3((sa.B + 255) / 256), dim3(256), 0, s, sa);
      };
      ops_grads.push_back(op);
    }
    for (int l = 0; l < L; ++l) {
      std::vector<IgemmProb> pr;
      pr.push_back(head_layer(m_qf1, P, hQF1PI, l, feat[1], ldf, F, pi_a, A, A, B));
      pr.push_back(head_layer(m_qf2, P, hQF2PI, l, feat[1], ldf, F, pi_a, A, A, B));
      add_launch(ops_grads, "heads_fwd", 0, pr);
    }
    {
      std::vector<IgemmProb> pr;
      pr.push_back(head_out(m_qf1, P, hQF1PI, 0, B));
      pr.push_back(head_out(m_qf2, P, hQF2PI, 0, B));
      add_launch(ops_grads, "heads_fwd", 0, pr);
    }
  }
  {
    LossArgs la;
    la.B = B; la.gamma = c.gamma; la.target_entropy = c.target_entropy; la.lr = c.lr;
    la.rew = rew; la.done = done; la.v_tgt = hTGT.out[0]; la.qf1 = hQF1.out[0]; la.qf2 = hQF2.out[0];
    la.v = hVF.out[0]; la.qf1_pi = hQF1PI.out[0]; la.qf2_pi = hQF2PI.out[0]; la.logp = logp; la.entropy = ent;
    la.log_ent_coef = params + ent_off;
    la.d_qf1 = d_qf1; la.d_qf2 = d_qf2; la.d_v = d_v; la.d_qf1_pi = d_qf1pi; la.ld_d = ld_d;
    la.g_log_ent_coef = grads + ent_off; la.sc = sc;
    la.write_d = fused_heads ? 0 : 1;
    la.adam_ticked = fused_heads ? 1 : 0;
    la.ent_param = params + ent_off; la.ent_m = adam_m + ent_off; la.ent_v = adam_v + ent_off;
    loss_args = la;
    if (!fused_heads) {   // fused heads: output gradients are formed in heads_bwd_kernel, reductions ride on reduce_slabs
      Op op; op.tag = "sac_loss";
      op.run = [la](hipStream_t s) { hipLaunchKernelGGL(sac_loss_kernel, dim3(1), dim3(256), 0, s, la); };
      ops_grads.push_back(op);
    }
  }

  // =============================================================== backward through the heads
  // g[l] = gradient w.r.t. the pre-activation of layer l (ReLU mask already applied)
  if (fused_heads) {
    HeadsBwdArgs ba;
    memset(&ba, 0, sizeof(ba));
    ba.h[0] = mk_head(m_pi, hPI, &gPI, u_l0[0], nullptr, 0, 0);
    ba.h[1] = mk_head(m_vf, hVF, &gVF, u_l0[1], nullptr, 0, 0);
    ba.h[2] = mk_head(m_qf1, hQF1, &gQF1, u_l0[2], act, A, A);
    ba.h[3] = mk_head(m_qf2, hQF2, &gQF2, u_l0[3], act, A, A);
    ba.h[4] = mk_head(m_qf1, hQF1PI, &gQF1PI, u_l0[2], pi_a, A, A);
    ba.h[1].dout[0] = d_v; ba.h[2].dout[0] = d_qf1; ba.h[3].dout[0] = d_qf2; ba.h[4].dout[0] = d_qf1pi;
    for (int k = 1; k < 5; ++k) ba.h[k].ld_dout = ld_d;
    ba.ld_dm = ld_dm;
    ba.rew = rew; ba.done = done; ba.v_tgt = hTGT.out[0]; ba.qf1 = hQF1.out[0]; ba.qf2 = hQF2.out[0]; ba.v = hVF.out[0];
    ba.qf1_pi = hQF1PI.out[0]; ba.qf2_pi = hQF2PI.out[0]; ba.logp = logp; ba.gamma = c.gamma;
    ba.d_out[1] = d_v; ba.d_out[2] = d_qf1; ba.d_out[3] = d_qf2; ba.d_out[4] = d_qf1pi; ba.ld_d = ld_d;
    ba.B = B; ba.A = A; ba.mu = hPI.out[0]; ba.ls_raw = hPI.out[1]; ba.eps = eps_buf; ba.pi_a = pi_a;
    ba.log_ent_coef = params + ent_off; ba.da_pi = da_pi; ba.dmu = dmu; ba.dls = dls;
    if (!heads_mfma) {
    Op op; op.tag = "heads_bwd";
    op.run = [ba](hipStream_t s) {
      hipLaunchKernelGGL(heads_bwd_kernel, dim3((ba.B + HT_RB - 1) / HT_RB, 4), dim3(256), 0, s, ba);
    };
    ops_grads.push_back(op);
    }
    if (cnn) {
      // d feat = g0 . W0[0:Fc]^T, masked by feat > 0.  Critic net: the three layer-0 gradients sit side
      // by side in g0cat and the three kernels are reached through a table (K = 3*H0 in one pass).
      const int H0 = hid[0];
      std::vector<int32_t> qt3(3 * H0), qt1(H0);
      const int64_t offs[3] = {m_vf.w[0], m_qf1.w[0], m_qf2.w[0]};
      for (int k = 0; k < 3; ++k)
        for (int n = 0; n < H0; ++n) qt3[k * H0 + n] = (int32_t)(offs[k] - offs[0]) + n;
      for (int n = 0; n < H0; ++n) qt1[n] = n;
      auto tab_bwd = [&](const float* g, int ldg, int K, const float* wbase, const int32_t* dtab, float* dx,
                         const float* mask) {
        IgemmProb p = blank();
        p.M = B; p.N = Fc; p.K = K;
        p.p_base[0] = g; p.p_ld_i[0] = ldg; p.p_ld_r[0] = 1; single_part(p);
        p.q_base[0] = wbase; p.q_tab_r = dtab; p.q_ld_j[0] = H0;
        p.c = dx; p.ldc = ldf; p.relu_mask = mask;
        p.vflags = VF_Q_TAB;
        set_split(p, 1);
        return p;
      };
      std::vector<IgemmProb> pr;
      pr.push_back(tab_bwd(g0cat, 3 * H0, 3 * H0, P + m_vf.w[0], upload_vec(wk, qt3), dfeat[1], feat[1]));
      pr.push_back(tab_bwd(gPI.g[0], H0, H0, P + m_pi.w[0], upload_vec(wk, qt1), dfeat[0], feat[0]));
      add_launch(ops_grads, "heads_dfeat", 1, pr);
    }
  } else {
    {
      std::vector<IgemmProb> pr;   // output layer -> g[L-1]
      pr.push_back(dense_bwd({{d_v, 1, 1, P + m_vf.ow[0]}}, B, 0, hid[L - 1], gVF.g[L - 1], hid[L - 1], hVF.z[L - 1]));
      pr.push_back(dense_bwd({{d_qf1, 1, 1, P + m_qf1.ow[0]}}, B, 0, hid[L - 1], gQF1.g[L - 1], hid[L - 1], hQF1.z[L - 1]));
      pr.push_back(dense_bwd({{d_qf2, 1, 1, P + m_qf2.ow[0]}}, B, 0, hid[L - 1], gQF2.g[L - 1], hid[L - 1], hQF2.z[L - 1]));
      pr.push_back(dense_bwd({{d_qf1pi, 1, 1, P + m_qf1.ow[0]}}, B, 0, hid[L - 1], gQF1PI.g[L - 1], hid[L - 1], hQF1PI.z[L - 1]));
      add_launch(ops_grads, "heads_bwd", 1, pr);
    }
    for (int l = L - 1; l >= 1; --l) {
      std::vector<IgemmProb> pr;
      pr.push_back(dense_bwd({{gVF.g[l], hid[l], hid[l], P + m_vf.w[l]}}, B, 0, hid[l - 1], gVF.g[l - 1], hid[l - 1], hVF.z[l - 1]));
      pr.push_back(dense_bwd({{gQF1.g[l], hid[l], hid[l], P + m_qf1.w[l]}}, B, 0, hid[l - 1], gQF1.g[l - 1], hid[l - 1], hQF1.z[l - 1]));
      pr.push_back(dense_bwd({{gQF2.g[l], hid[l], hid[l], P + m_qf2.w[l]}}, B, 0, hid[l - 1], gQF2.g[l - 1], hid[l - 1], hQF2.z[l - 1]));
      pr.push_back(dense_bwd({{gQF1PI.g[l], hid[l], hid[l], P + m_qf1.w[l]}}, B, 0, hid[l - 1], gQF1PI.g[l - 1], hid[l - 1], hQF1PI.z[l - 1]));
      add_launch(ops_grads, "heads_bwd", 1, pr);
    }
    {
      std::vector<IgemmProb> pr;   // first layer: d a_pi (policy path) and d feat (critic CNN path)
      pr.push_back(dense_bwd({{gQF1PI.g[0], hid[0], hid[0], P + m_qf1.w[0]}}, B, F, A, da_pi, A, nullptr));
      if (cnn)
        pr.push_back(dense_bwd({{gVF.g[0], hid[0], hid[0], P + m_vf.w[0]},
                                {gQF1.g[0], hid[0], hid[0], P + m_qf1.w[0]},
                                {gQF2.g[0], hid[0], hid[0], P + m_qf2.w[0]}},
                               B, 0, Fc, dfeat[1], ldf, feat[1]));
      add_launch(ops_grads, "heads_bwd", 1, pr);
    }
    {
      SampleBwdArgs sb{hPI.out[0], hPI.out[1], eps_buf, pi_a, da_pi, A, params + ent_off, B, A, dmu, dls};
      Op op; op.tag = "sample_bwd";
      op.run = [sb](hipStream_t s) {
        hipLaunchKernelGGL(sample_bwd_kernel, dim3((sb.B + 255) / 256), dim3(256), 0, s, sb);
      };
      ops_grads.push_back(op);
    }
    {
      std::vector<IgemmProb> pr;
      pr.push_back(dense_bwd({{dmu, A, A, P + m_pi.ow[0]}, {dls, A, A, P + m_pi.ow[1]}}, B, 0, hid[L - 1],
                             gPI.g[L - 1], hid[L - 1], hPI.z[L - 1]));
      add_launch(ops_grads, "heads_bwd", 1, pr);
    }
    for (int l = L - 1; l >= 1; --l) {
      std::vector<IgemmProb> pr;
      pr.push_back(dense_bwd({{gPI.g[l], hid[l], hid[l], P + m_pi.w[l]}}, B, 0, hid[l - 1], gPI.g[l - 1], hid[l - 1], hPI.z[l - 1]));
      add_launch(ops_grads, "heads_bwd", 1, pr);
    }
    if (cnn) {
      std::vector<IgemmProb> pr;
      pr.push_back(dense_bwd({{gPI.g[0], hid[0], hid[0], P + m_pi.w[0]}}, B, 0, Fc, dfeat[0], ldf, feat[0]));
      add_launch(ops_grads, "heads_bwd", 1, pr);
    }

  }
  // =============================================================== backward through the two CNNs
  bool fillers_on = false, only_vec_dense = false;
  int rider_budget = 0;                            // empty slots of conv3_bwd's last dispatch round (see below)
  std::vector<Op> conv3_bwd_plain;                 // conv3_bwd without riders, for the staged data-parallel plan
  std::vector<IgemmProb> dense_affine, conv_all;   // dense / conv weight-gradient problems as first built (staged plan)
  std::vector<IgemmProb> wg, wgc[3];   // weight gradients: dense layers / conv layers 1..3
  std::vector<IgemmProb> bwd_pr[3];    // backward-data stages fc, conv3, conv2 (launched below, once their fillers are known)
  // Measured and rejected on top of the merged weight-gradient launch (MI355X, B = 256, updates/s; conv2_bwd needs 46 KB of
  // LDS, so a CU holds three of its 1024 tiles and the launch runs as a wave of 768 plus a third-full wave of 256):
  //  * every weight gradient except conv1's riding on conv2_bwd's launch, longest tiles first, conv1 (the only consumer
  //    of conv2_bwd's result) as a launch of its own: the pair launch takes 57.7 us against 31.8 + 36.5, but conv1 alone
  //    costs 13.2 us (one short tile per CU) and its 225 slabs another 8 us of reduction: 4 395 against 4 600;
  //  * only the dense layers' weight gradients (303 tiles of 8 slabs, the length of a conv2_bwd tile) riding behind
  //    conv2_bwd's tiles: they start when the first wave drains (18 us) and end at 36 us instead of 28: conv2_bwd
  //    31.8 -> 41.2 us, the weight-gradient launch 36.6 -> 28.2 us, reduction +2.5 us: 4 519 against 4 580.
  if (cnn) {
    // backward-data with exact taps (conv_bwd_tabs_exact); GRL_NO_EXACT_TAP=1 keeps the masked parity-class form
    // (bit-identical results, 1.8-2.25x the MACs: test / measurement switch)
    const char* net = getenv("GRL_NO_EXACT_TAP");
    exact_tap = !(net && atoi(net));
    std::vector<int32_t> untouched;
    std::vector<ConvBwdClass> bc3 = exact_tap ? conv_bwd_tabs_exact(cg[2], B, nullptr) : conv_bwd_tabs(cg[2], B);
    std::vector<ConvBwdClass> bc2 = exact_tap ? conv_bwd_tabs_exact(cg[1], B, &untouched) : conv_bwd_tabs(cg[1], B);
    if (!untouched.empty())   // input pixels of conv2 that no output window covers (row / column 14): their gradient is zero
      for (int n = 0; n < 2; ++n) zero_once.push_back({g1[n], (size_t)(((int64_t)B * 225 - 1) * ld1 + 32) * 4});
    for (int n = 0; n < 2; ++n)
      bwd_pr[0].push_back(dense_bwd({{dfeat[n], ldf, 512, P + ex[n].fw}}, B, 0, 1024, g3[n], 1024, a3[n]));
    for (int n = 0; n < 2; ++n)
      for (auto& cl : bc3) bwd_pr[1].push_back(conv_bwd(g3[n], cl, cg[2], P + ex[n].w[2], g2[n], a2[n]));
    for (int n = 0; n < 2; ++n)
      for (auto& cl : bc2) bwd_pr[2].push_back(conv_bwd(g2[n], cl, cg[1], P + ex[n].w[1], g1[n], a1[n]));
    // conv / fc weight gradients (split reductions land in slabs, summed by reduce_slabs)
    // conv3_bwd (576 tiles of the 32x64 shape at B = 256, three per CU) fills two dispatch rounds and a quarter of the third:
    // 192 CUs hold two tiles, 64 hold three, and the launch lasts as long as those 64.  Dense-layer weight gradients
    // whose operands are complete by then (d feat, head gradients) and whose tiles are as long as conv3_bwd's (reduction
    // = the batch, 8 slabs) ride in the empty slots of that round -- list positions 576.. land exactly on the CUs that
    // hold two -- and leave the merged weight-gradient launch.  GRL_NO_CONV3_RIDERS=1 switches it off.
    {
      int cfg3 = -1;
      const int t3 = planned_tiles(bwd_pr[1], 1, "conv3_bwd", &cfg3);
      const char* nr = getenv("GRL_NO_CONV3_RIDERS");
      const char* nfl0 = getenv("GRL_FILLERS");
      const char* nm0 = getenv("GRL_NO_WGRAD_MERGE");
      const char* le0 = getenv("GRL_LANES");
      const bool off = (nr && atoi(nr)) || (nfl0 && nfl0[0] == '1') || (nm0 && nm0[0] == '1') || (le0 && le0[0] == '1');
      rider_budget = (!off && cfg3 == 3) ? free_slots(t3, 3) : 0;
    }
    int wsplit[3] = {72, 12, 6};   // reduction splits of conv1..3 (GRL_WG_SPLIT=a,b,c overrides the model's choice)
    pick_wgrad_splits(cg, ft, ld1 == 64 ? 2 : 1, wsplit, rider_budget);
    if (const char* e = getenv("GRL_WG_SPLIT")) sscanf(e, "%d,%d,%d", &wsplit[0], &wsplit[1], &wsplit[2]);
    if (ld1 == 64) {   // conv1 of both networks: one problem over the side-by-side gradient buffer, columns 32n.. -> net n
      IgemmProb p = conv_wgrad(x_obs, ft[0], cg[0], g1[0], nullptr, wsplit[0], 2);
      p.c = wk.f32(p.slab_stride * p.split);
      wgc[0].push_back(p);
      for (int n = 0; n < 2; ++n) {
        ReduceDesc r;
        memset(&r, 0, sizeof(r));
        r.src = p.c + 32 * n; r.splits = p.split; r.slab_stride = p.slab_stride; r.row_len = 32; r.src_ld = 64;
        r.dst = grads + ex[n].w[0]; r.n = cg[0].K() * 32;
        reduces.push_back(r);
        ReduceDesc rb = r;
        rb.src = p.c + (int64_t)p.p_ones_i * 64 + 32 * n; rb.dst = grads + ex[n].b[0]; rb.n = 32;
        reduces.push_back(rb);
      }
    }
    for (int n = 0; n < 2; ++n) {
      const float* xin = x_obs;
      if (ld1 != 64) {
        IgemmProb p = conv_wgrad(xin, ft[0], cg[0], g1[n], nullptr, wsplit[0]);
        p.c = wk.f32(p.slab_stride * p.split);
        add_wgrad(wgc[0], p, ex[n].w[0], 0, cg[0].K(), ex[n].b[0]);
      }
      {
        IgemmProb p = conv_wgrad(a1[n], ft[1], cg[1], g2[n], nullptr, wsplit[1]);
        p.c = wk.f32(p.slab_stride * p.split);
        add_wgrad(wgc[1], p, ex[n].w[1], 0, cg[1].K(), ex[n].b[1]);
      }
      {
        IgemmProb p = conv_wgrad(a2[n], ft[2], cg[2], g3[n], nullptr, wsplit[2]);
        p.c = wk.f32(p.slab_stride * p.split);
        add_wgrad(wgc[2], p, ex[n].w[2], 0, cg[2].K(), ex[n].b[2]);
      }
      {
        IgemmProb p = dense_wgrad(a3[n], 1024, 1024, true, dfeat[n], ldf, 512, B, nullptr, 1);
        p.c = wk.f32(p.slab_stride * p.split);
        add_wgrad(wg, p, ex[n].fw, 0, 1024, ex[n].fb);
      }
    }
  }
  // head weight gradients.  Fused-heads layout: every operand is row-padded to a multiple of 4 floats, and
  // problems without a bias still carry the ones row (its slab row is simply not reduced), so that all
  // dense weight gradients form ONE uniform launch of the vectorised kernel.
  auto head_wgrads = [&](const MlpP& m, const HeadAct& h, const HeadGrad& g, const float* x0, int ld0, int K0,
                         const float* x1, int ld1, int K1, std::vector<const float*> douts, int ld_dout) {
    const int K0p = ld0 >= (int)rup(K0, 4) ? (int)rup(K0, 4) : K0;   // feat rows are padded to ldf (zeros)
    const int K1p = (K1 > 0 && ld1 >= (int)rup(K1, 4)) ? (int)rup(K1, 4) : K1;
    for (int l = 0; l < L; ++l) {
      if (l == 0) {
        if (K1 > 0) {
          IgemmProb p0 = dense_wgrad(x0, ld0, K0, fused_heads, g.g[0], g.ld0, hid[0], B, nullptr, 1, K0p);
          p0.c = wk.f32(p0.slab_stride * p0.split);
          add_wgrad(wg, p0, m.w[0], 0, K0, -1);
          IgemmProb p1 = dense_wgrad(x1, ld1, K1, true, g.g[0], g.ld0, hid[0], B, nullptr, 1, K1p);
          p1.c = wk.f32(p1.slab_stride * p1.split);
          add_wgrad(wg, p1, m.w[0], K0, K1, m.b[0]);
        } else {
          IgemmProb p0 = dense_wgrad(x0, ld0, K0, true, g.g[0], g.ld0, hid[0], B, nullptr, 1, K0p);
          p0.c = wk.f32(p0.slab_stride * p0.split);
          add_wgrad(wg, p0, m.w[0], 0, K0, m.b[0]);
        }
      } else {
        IgemmProb p = dense_wgrad(h.z[l - 1], hid[l - 1], hid[l - 1], true, g.g[l], hid[l], hid[l], B, nullptr, 1);
        p.c = wk.f32(p.slab_stride * p.split);
        add_wgrad(wg, p, m.w[l], 0, hid[l - 1], m.b[l]);
      }
    }
    for (int k = 0; k < m.n_out; ++k) {
      IgemmProb p = dense_wgrad(h.z[L - 1], hid[L - 1], hid[L - 1], true, douts[k], ld_dout, m.out_dim, B, nullptr, 1);
      p.c = wk.f32(p.slab_stride * p.split);
      add_wgrad(wg, p, m.ow[k], 0, hid[L - 1], m.ob[k]);
    }
  };
  const float* act_w = fused_heads ? act_p : act;
  const int ld_act_w = fused_heads ? Ap : A;
  head_wgrads(m_pi, hPI, gPI, feat[0], ldf, F, nullptr, 0, 0, {dmu, dls}, ld_dm);
  head_wgrads(m_vf, hVF, gVF, feat[1], ldf, F, nullptr, 0, 0, {d_v}, ld_d);
  head_wgrads(m_qf1, hQF1, gQF1, feat[1], ldf, F, act_w, ld_act_w, A, {d_qf1}, ld_d);
  head_wgrads(m_qf2, hQF2, gQF2, feat[1], ldf, F, act_w, ld_act_w, A, {d_qf2}, ld_d);
  {
    // the vectorised kernel needs uniform launches: with / without bias row; whatever it cannot take
    // goes to igemm_kernel.  (Fused heads: everything lands in the first group.)
    std::vector<IgemmProb> wg_ones, wg_plain, wg_rest;
    for (auto& p : wg) {
      if (!v2_prob_ok(p, 2) || (p.K % 4)) wg_rest.push_back(p);
      else if (p.p_ones_i >= 0) wg_ones.push_back(p);
      else wg_plain.push_back(p);
    }
    const char* le = getenv("GRL_LANES");
    use_lanes = le && le[0] == '1';   // ROCm's graph scheduler serialises most forked kernels: off by default
    wgrad_ops.clear();
    // One launch for all weight gradients: the dense problems (x^T read with affine addresses) are re-expressed
    // with the table addressing of the convolution ones and appended to that launch -- one kernel boundary
    // less, and their short reductions (K = B) fill the tail of the long convolution tiles.
    dense_affine = wg_ones;                                   // (kept for the staged data-parallel plan below)
    for (int l = 2; l >= 0; --l) conv_all.insert(conv_all.end(), wgc[l].begin(), wgc[l].end());
    only_vec_dense = wg_plain.empty() && wg_rest.empty();
    std::vector<IgemmProb> wg_merged;
    const char* nm = getenv("GRL_NO_WGRAD_MERGE");
    if (cnn && !use_lanes && !(nm && nm[0] == '1') && wg_plain.empty() && !wgc[0].empty() && v2_prob_ok(wgc[0][0], 2)) {
      for (auto p : wg_ones) {
        std::vector<int32_t> ti(p.M), tr(p.K);
        for (int i = 0; i < p.M; ++i) ti[i] = i * p.p_ld_i[0];
        for (int r = 0; r < p.K; ++r) tr[r] = r * p.p_ld_r[0];
        p.p_tab_i = upload_vec(wk, ti);
        p.p_tab_r = upload_vec(wk, tr);
        p.vflags |= VF_P_TABS;                    // v2_prob_ok held for the affine form: unit stride along i, ld_r % 4 == 0
        wg_merged.push_back(p);
      }
      wg_ones.clear();
    }
    // Weight gradients as FILLERS of the backward-data launches (opt-in, GRL_FILLERS=1): each group only needs
    // tensors that are complete when the stage it rides on starts (fc + head layers: d feat and the head gradients;
    // conv3: g3 from fc_bwd; conv2: g2 from conv3_bwd), so its tiles share that stage's launch
    // (igemm2_pair_kernel) and only conv1's (needs g1, the last backward-data result) remain a launch of their own.
    // Same tiles, same workgroup shape, bit-identical results -- but MEASURED SLOWER on MI355X at B = 256
    // (fc_bwd 10.9 -> 21.1, conv3_bwd 23.3 -> 30.3, conv2_bwd 29.8 -> 43.5 us, conv1's weight gradient alone 21.0
    // against 38.5 us for the single merged launch: 116 vs 102.5 us, 3998 vs 4287 updates/s): the 10-us reduction
    // chunks of the weight gradients lengthen every stage's tail by more than the launch they save, and the
    // merged launch (747 tiles, heaviest first) was already the better packing.  Kept as a tested switch.
    const char* nfl = getenv("GRL_FILLERS");
    const bool fillers = cnn && !wg_merged.empty() && nfl && nfl[0] == '1';
    fillers_on = fillers;
    if (cnn) {
      if (fillers) {
        add_launch(ops_grads, "fc_bwd", 1, bwd_pr[0], "wgrad_dense", 2, wg_merged);
        add_launch(ops_grads, "conv3_bwd", 1, bwd_pr[1], "wgrad_conv3", 2, wgc[2]);
        add_launch(ops_grads, "conv2_bwd", 1, bwd_pr[2], "wgrad_conv2", 2, wgc[1]);
        wg_merged.clear(); wgc[2].clear(); wgc[1].clear();
      } else {
        add_launch(ops_grads, "fc_bwd", 1, bwd_pr[0]);
        std::vector<IgemmProb> riders = take_riders(wg_merged, rider_budget);
        if (!riders.empty()) add_launch(conv3_bwd_plain, "conv3_bwd", 1, bwd_pr[1]);
        add_launch(ops_grads, "conv3_bwd", 1, bwd_pr[1], "wgrad_dense", 2, riders);
        add_launch(ops_grads, "conv2_bwd", 1, bwd_pr[2]);
      }
    }
    add_launch(wgrad_ops, "wgrad_dense", 2, wg_ones);
    add_launch(wgrad_ops, "wgrad_dense", 2, wg_plain);
    add_launch(wgrad_ops, "wgrad_small", 2, wg_rest);
    if (use_lanes) {
      add_launch(wgrad_ops, "wgrad_conv3", 2, wgc[2]);
      add_launch(wgrad_ops, "wgrad_conv2", 2, wgc[1]);
      add_launch(wgrad_ops, "wgrad_conv1", 2, wgc[0]);
    } else {
      std::vector<IgemmProb> all;
      for (int l = 2; l >= 0; --l) all.insert(all.end(), wgc[l].begin(), wgc[l].end());
      all.insert(all.end(), wg_merged.begin(), wg_merged.end());
      add_launch(wgrad_ops, "wgrad_conv", 2, all, "", 0, {}, fillers ? 0 : -1);   // (fillers: conv1 alone keeps the 64x64 shape)
    }
  }
  // ---- schedule: weight gradients ride on the side lane next to the backward-data chain.  In list
  // (= eager) order every op still follows its producers.
  {
    std::vector<Op> sched;
    auto take = [&](const char* tag, bool side_lane) {
      for (auto& o : wgrad_ops)
        if (o.tag == tag) {
          Op c = o;
          if (side_lane && use_lanes) { c.lane = 1; c.fork = true; }
          sched.push_back(c);
        }
    };
    int dense_after = -1;   // the dense weight gradients need every head gradient and (CNN) d feat
    for (size_t k = 0; k < ops_grads.size(); ++k)
      if (ops_grads[k].tag == "heads_dfeat" || ops_grads[k].tag == "heads_bwd" || ops_grads[k].tag == "heads") dense_after = (int)k;
    for (size_t k = 0; k < ops_grads.size(); ++k) {
      const Op& o = ops_grads[k];
      sched.push_back(o);
      if ((int)k == dense_after) { take("wgrad_dense", cnn); take("wgrad_small", cnn); }
      if (o.tag == "fc_bwd") take("wgrad_conv3", true);
      else if (o.tag == "conv3_bwd") take("wgrad_conv2", true);
      else if (o.tag == "conv2_bwd") { take("wgrad_conv1", false); take("wgrad_conv", false); }
    }
    ops_grads.swap(sched);
  }
  std::vector<Op> st0_ops, st1_ops;     // staged plan without its two reductions (added below)
  {
    // ---- staged gradient computation (data parallel): the dense weight gradients -- 90 % of the bucket's bytes --
    // get a launch of their own right after the feature gradients, so that their all-reduce can travel while the
    // convolution backward and the convolution weight gradients run (grasp_rl/parallel.py).  Costs two launches
    // more than the single-exchange plan; same tiles, same arithmetic.
    staged_ok = cnn && fused_heads && only_vec_dense && !dense_affine.empty() && !conv_all.empty() && !fillers_on && !use_lanes;
    if (staged_ok) {
      int cut = -1;
      for (size_t k = 0; k < ops_grads.size(); ++k)
        if (ops_grads[k].tag == "heads_dfeat") cut = (int)k;
      staged_ok = cut >= 0;
      if (staged_ok) {
        for (int k = 0; k <= cut; ++k) st0_ops.push_back(ops_grads[k]);
        add_launch(st0_ops, "wgrad_dense", 2, dense_affine, "", 0, {}, 0);
        for (size_t k = cut + 1; k < ops_grads.size(); ++k) {
          if (ops_grads[k].tag.compare(0, 5, "wgrad") == 0) continue;
          if (ops_grads[k].tag == "conv3_bwd" && !conv3_bwd_plain.empty()) st1_ops.push_back(conv3_bwd_plain[0]);   // (riders belong to stage 0 here)
          else st1_ops.push_back(ops_grads[k]);
        }
        add_launch(st1_ops, "wgrad_conv", 2, conv_all, "", 0, {}, 0);
      }
    }
  }
  {
    std::vector<int2> rt = reduce_tiles();          // (marks the 16-byte-eligible descriptors: before the upload)
    d_reduces = upload_vec(wk, reduces);
    int2* d_rt = upload_vec(wk, rt);
    const int ntiles = (int)rt.size();
    ReduceDesc* dr = d_reduces;
    Op op; op.tag = "reduce_slabs";
    op.join = true;
    const LossArgs la = loss_args;
    const int has_loss = fused_heads ? 1 : 0;
    AdamArgs aa;
    memset(&aa, 0, sizeof(aa));
    aa.params = params; aa.grads = grads; aa.m = adam_m; aa.v = adam_v; aa.n_train = n_train; aa.sc = sc;
    aa.grad_scale = 1.f; aa.tau = c.tau; aa.eps = 1e-8f;
    aa.src_ofs = vf_off; aa.n_polyak = n_polyak; aa.target = params + tgt_off;
    op.run = [dr, d_rt, ntiles, la, has_loss, aa](hipStream_t s) {
      hipLaunchKernelGGL(reduce_slabs_kernel, dim3(ntiles + has_loss), dim3(256), 0, s, dr, d_rt, ntiles, la, has_loss, aa, 0);
    };
    ops_grads.push_back(op);
    if (staged_ok) {
      // reductions of the two stages: convolution descriptors are those that land in the conv variables of a net
      auto is_conv = [&](const ReduceDesc& r) {
        for (int n = 0; n < 2; ++n)
          if (r.dst >= grads + ex[n].w[0] && r.dst < grads + ex[n].fw) return true;
        return false;
      };
      std::vector<int2> rt0 = reduce_tiles([&](const ReduceDesc& r) { return !is_conv(r); });
      std::vector<int2> rt1 = reduce_tiles(is_conv);
      int2* d_rt0 = upload_vec(wk, rt0);
      int2* d_rt1 = upload_vec(wk, rt1);
      const int n0 = (int)rt0.size(), n1 = (int)rt1.size();
      Op r0; r0.tag = "reduce_dense";
      r0.run = [dr, d_rt0, n0, la, aa](hipStream_t s) {
        hipLaunchKernelGGL(reduce_slabs_kernel, dim3(n0), dim3(256), 0, s, dr, d_rt0, n0, la, 0, aa, 0);
      };
      Op r1; r1.tag = "reduce_conv";
      r1.run = [dr, d_rt1, n1, la, has_loss, aa](hipStream_t s) {
        hipLaunchKernelGGL(reduce_slabs_kernel, dim3(n1 + has_loss), dim3(256), 0, s, dr, d_rt1, n1, la, has_loss, aa, 0);
      };
      ops_stage0 = st0_ops; ops_stage0.push_back(r0);
      ops_stage1 = st1_ops; ops_stage1.push_back(r1);
    }
    // Full updates (no gradient exchange in between): every trainable element is the sum of one slab
    // column, so Adam + Polyak are applied where the sum is formed -- one launch and one pass over the
    // gradient bucket less.  log_ent_coef, whose gradient comes from the loss workgroup, is applied there.
    const char* nf = getenv("GRL_NO_FUSED_ADAM");
    if (has_loss && !(nf && nf[0] == '1')) {
      ops_grads_apply.assign(ops_grads.begin(), ops_grads.end() - 1);
      Op fo; fo.tag = "reduce_adam";
      fo.join = true;
      fo.bytes = (double)n_train * 4 * 7 + (double)n_polyak * 4 * 2;
      fo.run = [dr, d_rt, ntiles, la, has_loss, aa](hipStream_t s) {
        hipLaunchKernelGGL(reduce_slabs_kernel, dim3(ntiles + has_loss), dim3(256), 0, s, dr, d_rt, ntiles, la, has_loss, aa, 1);
      };
      ops_grads_apply.push_back(fo);
      // ---- "prefetch": a call of n >= 2 updates on the device RNG gathers the minibatch of update t+1 inside the LAST
      // launch of update t (reduce_slabs_gather_kernel): nothing enters the replay between the updates of one call, the
      // Philox counter makes the draw independent of when it happens, and every reader of update t's minibatch tensors
      // has finished when that launch starts.  Same kernels, same arithmetic, one launch (and one dependent latency
      // chain) less per update.  The head launch opens the update instead of the gather (see HeadsFusedArgs).
      //   first : gather (does not touch the Adam step size) | body, heads[tick] | reduce + Adam + gather(t+1, counter + 1)
      //   middle:                                              body, heads[tick, counter += 1] | reduce + Adam + gather(t+1)
      //   last  :                                              body, heads[tick, counter += 1] | reduce + Adam (counter += 1)
      const char* npf = getenv("GRL_NO_GATHER_PREFETCH");
      if (heads_mfma && !(npf && atoi(npf)) && !use_lanes) {
        GatherArgs g1 = pf_ga;
        g1.use_rng = 1; g1.adam_tick = 0; g1.quiet = 0; g1.rng_ahead = 0;
        const int gx = pf_gx;
        {
          Op op; op.tag = "gather_norm";
          op.bytes = ops_rng[0].bytes;
          op.run = [g1, gx](hipStream_t s) { hipLaunchKernelGGL(gather_norm_kernel, dim3(gx, g1.B, 2), dim3(256), 0, s, g1); };
          ops_pf_first.push_back(op);
        }
        GatherArgs g2 = g1;
        g2.quiet = 1; g2.rng_ahead = 1;
        LossArgs lk = la;
        lk.keep_rng = 1;
        Op ro; ro.tag = "reduce_adam";
        ro.join = true;
        ro.bytes = fo.bytes + ops_rng[0].bytes;
        ro.run = [dr, d_rt, ntiles, lk, has_loss, aa, g2, gx](hipStream_t s) {
          hipLaunchKernelGGL(reduce_slabs_gather_kernel, dim3(ntiles + has_loss + gx * g2.B * 2), dim3(256), 0, s, dr, d_rt, ntiles, lk,
                             has_loss, aa, 1, g2, gx);
        };
        for (int v = 0; v < 3; ++v) {     // 0 first, 1 middle, 2 last
          std::vector<Op>& dst = v == 0 ? ops_pf_first : (v == 1 ? ops_pf_mid : ops_pf_last);
          for (size_t k = 0; k + 1 < ops_grads_apply.size(); ++k) {
            if (ops_grads_apply[k].tag == "heads") dst.push_back(pf_heads[v == 0 ? 0 : 1]);
            else dst.push_back(ops_grads_apply[k]);
          }
          dst.push_back(v == 2 ? fo : ro);
        }
        prefetch_ok = true;
      }
    }
  }

  // =============================================================== apply
  {
    Op op; op.tag = "adam_polyak";
    op.bytes = (double)n_train * 4 * 7 + (double)n_polyak * 4 * 2;
    grl_ctx* self = this;
    op.run = [self](hipStream_t s) {
      AdamArgs aa;
      aa.params = self->params; aa.grads = self->grads; aa.m = self->adam_m; aa.v = self->adam_v;
      aa.n_train = self->n_train; aa.sc = self->sc; aa.grad_scale = self->grad_scale; aa.tau = self->cfg.tau; aa.eps = 1e-8f;
      aa.src_ofs = self->vf_off; aa.n_polyak = self->n_polyak; aa.target = self->params + self->tgt_off;
      const int blocks = (int)std::min<int64_t>(2048, (self->n_train + 255) / 256);
      hipLaunchKernelGGL(adam_polyak_kernel, dim3(blocks), dim3(256), 0, s, aa);
    };
    ops_apply.push_back(op);
  }

  // =============================================================== act path (batch NA, pi net only)
  {
    ax = cnn ? wk.f32((int64_t)NA * img_elems) : nullptr;
    afeat = wk.f32((int64_t)NA * ldf);
    a_eps = wk.f32((int64_t)NA * A);
    a_out = wk.f32((int64_t)NA * A);
    alloc_head(ahPI, NA, 2, A);
    ActIngestArgs ia;
    memset(&ia, 0, sizeof(ia));
    ia.obs = stg_obs; ia.n = NA; ia.hw = hw * hw; ia.c_obs = c.obs_channels; ia.c_img = C_img; ia.n_direct = nd;
    ia.vec_dim = cnn ? 0 : c.obs_dim; ia.scale_div = cnn ? 255.f : 1.f;
    ia.x = cnn ? ax : afeat; ia.ldx = cnn ? img_elems : ldf; ia.d = afeat + 512; ia.ldd = ldf;
    ActIngestArgs ian = ia;
    ian.normalize = 1; ian.clip_obs = c.clip_obs;
    ian.mean = s_mean; ian.stdv = s_std; ian.dmean = s_dmean; ian.dstd = s_dstd;
    {
      const int elems = cnn ? img_elems : c.obs_dim;
      for (int v = 0; v < 2; ++v) {
        const ActIngestArgs iv = v ? ian : ia;
        Op op; op.tag = "act_ingest";
        op.run = [iv, elems](hipStream_t s) {
          hipLaunchKernelGGL(act_ingest_kernel, dim3((elems + 255) / 256, iv.n), dim3(256), 0, s, iv);
        };
        (v ? ops_act_norm : ops_act).push_back(op);
      }
    }
    if (cnn) {
      aa1 = wk.f32((int64_t)NA * 225 * 32); aa2 = wk.f32((int64_t)NA * 36 * 64); aa3 = wk.f32((int64_t)NA * 16 * 64);
      float* io[4] = {ax, aa1, aa2, aa3};
      for (int l = 0; l < 3; ++l) {
        ConvGeom ag = cg[l];
        ag.ldx = ag.ldy = 0;                      // the acting pass keeps dense buffers of its own
        ConvFwdTabs t = conv_fwd_tabs(ag, NA);
        add_launch(ops_act, "act_conv", 0,
                   {conv_fwd(io[l], t, ag, P + ex[0].w[l], P + ex[0].b[l], io[l + 1], ACT_RELU, 0.f)});
      }
      add_launch(ops_act, "act_fc", 0,
                 {dense_fwd(aa3, 1024, 1024, nullptr, 0, 0, NA, P + ex[0].fw, 512, P + ex[0].fb, afeat, ldf, ACT_RELU)});
    }
    for (int l = 0; l < L; ++l)
      add_launch(ops_act, "act_head", 0, {head_layer(m_pi, P, ahPI, l, afeat, ldf, F, nullptr, 0, 0, NA)});
    add_launch(ops_act, "act_head", 0, {head_out(m_pi, P, ahPI, 0, NA), head_out(m_pi, P, ahPI, 1, NA)});
    for (size_t k = 1; k < ops_act.size(); ++k) ops_act_norm.push_back(ops_act[k]);
    // final tanh (+ sampling): two variants so that each is a static graph
    for (int det = 0; det < 2; ++det) {
      const float* mu = ahPI.out[0]; const float* ls = ahPI.out[1]; const float* ep = a_eps; float* ao = a_out;
      const int rows = NA, Ad = A;
      Op op; op.tag = "act_out";
      op.run = [mu, ls, ep, ao, rows, Ad, det](hipStream_t s) {
        hipLaunchKernelGGL(act_out_kernel, dim3((rows * Ad + 255) / 256), dim3(256), 0, s, mu, ls, ep, rows, Ad, det, ao);
      };
      (det ? ops_act_det : ops_act_sto).push_back(op);
    }
  }

  // =============================================================== Keras auto-encoder (A.9), batch NA
  {
    const ConvGeom eg[3] = {{64, 64, 1, 7, 7, 2, 2, 32, 32, 32}, {32, 32, 32, 5, 5, 2, 1, 16, 16, 32},
                            {16, 16, 32, 3, 3, 2, 0, 8, 8, 32}};
    const int64_t wn[8] = {7 * 7 * 32, 32, 5 * 5 * 32 * 32, 32, 3 * 3 * 32 * 32, 32, 2048 * 100, 100};
    for (int k = 0; k < 8; ++k) enc_w[k] = wk.f32(wn[k]);
    ex_in = wk.f32((int64_t)NA * 4096);
    ec1 = wk.f32((int64_t)NA * 32 * 32 * 32); ec2 = wk.f32((int64_t)NA * 16 * 16 * 32);
    ec3 = wk.f32((int64_t)NA * 8 * 8 * 32); eout = wk.f32((int64_t)NA * 100);
    float* io[4] = {ex_in, ec1, ec2, ec3};
    for (int l = 0; l < 3; ++l) {
      ConvFwdTabs t = conv_fwd_tabs(eg[l], NA);
      add_launch(ops_enc, "enc_conv", 0,
                 {conv_fwd(io[l], t, eg[l], enc_w[2 * l], enc_w[2 * l + 1], io[l + 1], ACT_LEAKY, 0.1f)});
    }
    IgemmProb p = dense_fwd(ec3, 2048, 2048, nullptr, 0, 0, NA, enc_w[6], 100, enc_w[7], eout, 100, ACT_LEAKY);
    p.act_alpha = 0.1f;
    add_launch(ops_enc, "enc_dense", 0, {p});
  }

  // debug taps
  dbg["feat_pi"] = {feat[0], (int64_t)B * ldf};
  dbg["feat_vf"] = {feat[1], (int64_t)B * ldf};
  dbg["feat_tgt"] = {feat[2], (int64_t)B * ldf};
  if (cnn) {
    dbg["x_obs"] = {x_obs, (int64_t)B * img_elems};
    dbg["x_next"] = {x_next, (int64_t)B * img_elems};
    dbg["a1_pi"] = {a1[0], (int64_t)B * 225 * 32};   // (side-by-side layout: the first half of the pair buffer, stride 64)
    dbg["a2_pi"] = {a2[0], (int64_t)B * 36 * 64};
    dbg["a3_pi"] = {a3[0], (int64_t)B * 1024};
    dbg["g1_vf"] = {g1[1], (int64_t)B * 225 * 32};
    dbg["g2_vf"] = {g2[1], (int64_t)B * 36 * 64};
    dbg["g3_vf"] = {g3[1], (int64_t)B * 1024};
    dbg["dfeat_pi"] = {dfeat[0], (int64_t)B * ldf};
    dbg["dfeat_vf"] = {dfeat[1], (int64_t)B * ldf};
  }
  dbg["mu"] = {hPI.out[0], (int64_t)B * A};
  dbg["log_std"] = {hPI.out[1], (int64_t)B * A};
  dbg["pi"] = {pi_a, (int64_t)B * A};
  dbg["logp"] = {logp, B};
  dbg["qf1"] = {hQF1.out[0], B}; dbg["qf2"] = {hQF2.out[0], B};
  dbg["v"] = {hVF.out[0], B}; dbg["v_tgt"] = {hTGT.out[0], B};
  dbg["qf1_pi"] = {hQF1PI.out[0], B}; dbg["qf2_pi"] = {hQF2PI.out[0], B};
  dbg["rew"] = {rew, B}; dbg["done"] = {done, B}; dbg["act"] = {act, (int64_t)B * A};
  dbg["dmu"] = {dmu, (int64_t)B * A}; dbg["dls"] = {dls, (int64_t)B * A}; dbg["da_pi"] = {da_pi, (int64_t)B * A};
  dbg["grads"] = {grads, n_train};
  dbg["adam_m"] = {adam_m, n_train};
  dbg["adam_v"] = {adam_v, n_train};
  dbg["idx_raw"] = {(const float*)idx_buf, (int64_t)2 * B};     // replay indices of the last minibatch: int64 viewed as float pairs
  return GRL_OK;
}

// --------------------------------------------------------------------------------------------------
// DQN (sb_helper.py:159-165) and BDQ (sb_helper.py:210-224): MLP towers on vector observations.
// One structure covers both: optional shared trunk -> D advantage branches + a state
// value tower, dueling aggregation per branch, double-Q target averaged over branches.
namespace {
struct QNetP {
  std::vector<int64_t> cw, cb;                    // trunk
  std::vector<std::vector<int64_t>> bw, bb;       // [branch][hidden..., out]
  std::vector<int64_t> vw, vb;                    // value tower [hidden..., out]
};
struct QNetAct {
  std::vector<float*> zc;
  std::vector<std::vector<float*>> zb;
  float* adv = nullptr;
  std::vector<float*> zv;
  float* v = nullptr;
};
}  // namespace

int grl_ctx::plan_q() {
  // 1 once the plan ends with the fused reduction + clip + Adam launch: the loss launch then leaves its batch sums to it
  // (no device-scope fence / last-workgroup pass) and the gather launch fixes the Adam step size.  Read at launch time.
  auto q_defer = std::make_shared<int>(0);
  auto q_ga = std::make_shared<GatherArgs>();   // the minibatch gather of this plan (filled below; the prioritised sampler can run it)
  memset(q_ga.get(), 0, sizeof(GatherArgs));
  float* q_row_part = nullptr;
  int q_finish = 0;
  const grl_config& c = cfg;
  cnn = false;
  const int D = c.q_branches, nb = c.q_bins, Lc = c.q_n_common, Lb = c.q_n_branch, Lv = c.q_n_value;
  qD = D; qN = nb;
  A = D; L = 0; B = c.batch_size; NA = std::max(1, c.act_batch);
  img_elems = c.obs_dim; F = c.obs_dim; Fc = 0; ldf = (int)rup(F, 4); C_img = 0; hw = c.img_hw;
  const std::string scope = c.algo == GRL_ALGO_DQN ? "deepq" : "bdq";
  auto fcname = [](int k) { return k == 0 ? std::string("fully_connected") : "fully_connected_" + std::to_string(k); };

  // ---------------- layout (TF creation order of the shipped zips)
  add_var(scope + "/eps:0", {}, false);   // exploration epsilon: stored with the model, never trained
  auto add_net = [&](const std::string& pre, QNetP& P, bool tr) {
    int d = c.obs_dim;
    for (int k = 0; k < Lc; ++k) {
      P.cw.push_back(add_var(pre + "/common_net/" + fcname(k) + "/weights:0", {d, c.q_common[k]}, tr));
      P.cb.push_back(add_var(pre + "/common_net/" + fcname(k) + "/biases:0", {c.q_common[k]}, tr));
      d = c.q_common[k];
    }
    int k = 0;
    P.bw.resize(D); P.bb.resize(D);
    for (int br = 0; br < D; ++br) {
      int dd = d;
      for (int l = 0; l < Lb; ++l, ++k) {
        P.bw[br].push_back(add_var(pre + "/action_value/" + fcname(k) + "/weights:0", {dd, c.q_branch[l]}, tr));
        P.bb[br].push_back(add_var(pre + "/action_value/" + fcname(k) + "/biases:0", {c.q_branch[l]}, tr));
        dd = c.q_branch[l];
      }
      P.bw[br].push_back(add_var(pre + "/action_value/" + fcname(k) + "/weights:0", {dd, nb}, tr));
      P.bb[br].push_back(add_var(pre + "/action_value/" + fcname(k) + "/biases:0", {nb}, tr));
      ++k;
    }
    int dd = d;
    for (int l = 0; l < Lv; ++l) {
      P.vw.push_back(add_var(pre + "/state_value/" + fcname(l) + "/weights:0", {dd, c.q_value[l]}, tr));
      P.vb.push_back(add_var(pre + "/state_value/" + fcname(l) + "/biases:0", {c.q_value[l]}, tr));
      dd = c.q_value[l];
    }
    P.vw.push_back(add_var(pre + "/state_value/" + fcname(Lv) + "/weights:0", {dd, 1}, tr));
    P.vb.push_back(add_var(pre + "/state_value/" + fcname(Lv) + "/biases:0", {1}, tr));
  };
  QNetP Pon, Ptg;
  q_online_off = n_params;
  add_net(scope + "/model", Pon, true);
  q_online_n = n_params - q_online_off;
  n_train = n_params;       // eps sits inside the bucket with a permanently zero gradient
  tgt_off = n_params;
  add_net(scope + "/target_q_func/model", Ptg, false);
  vf_off = 0; n_polyak = 0; ent_off = 0;

  // ---------------- arenas
  params = st.f32(n_params);
  adam_m = st.f32(n_train);
  adam_v = st.f32(n_train);
  sc = (DevScalars*)st.take(sizeof(DevScalars));
  s_mean = (double*)st.take((size_t)img_elems * 8);
  s_std = (double*)st.take((size_t)img_elems * 8);
  s_dmean = (double*)st.take(8); s_dstd = (double*)st.take(8);
  s_ret = (double*)st.take(8);
  grads = gr.f32(n_train);
  const int64_t cap = c.replay_capacity;
  rp_obs = rp.f32(cap * img_elems); rp_next = rp.f32(cap * img_elems);
  rp_dobs = rp.f32(cap); rp_dnext = rp.f32(cap);
  rp_act = rp.f32(cap * A); rp_rew = rp.f32(cap); rp_done = rp.f32(cap);
  stg_n = std::max(NA, 64);
  stg_obs = wk.f32((int64_t)stg_n * c.obs_dim); stg_next = wk.f32((int64_t)stg_n * c.obs_dim);
  stg_act = wk.f32((int64_t)stg_n * A); stg_rew = wk.f32(stg_n); stg_done = wk.f32(stg_n);
  idx_buf = (int64_t*)wk.take((size_t)B * 8);
  eps_buf = wk.f32(std::max(B, B * A));      // importance weights [B]
  for (int n = 0; n < 3; ++n) feat[n] = wk.f32((int64_t)B * ldf);
  act = wk.f32((int64_t)B * A); rew = wk.f32(B); done = wk.f32(B);

  auto alloc_net = [&](QNetAct& a, int rows) {
    for (int k = 0; k < Lc; ++k) a.zc.push_back(wk.f32((int64_t)rows * c.q_common[k]));
    a.zb.resize(D);
    for (int br = 0; br < D; ++br)
      for (int l = 0; l < Lb; ++l) a.zb[br].push_back(wk.f32((int64_t)rows * c.q_branch[l]));
    a.adv = wk.f32((int64_t)rows * D * nb);
    for (int l = 0; l < Lv; ++l) a.zv.push_back(wk.f32((int64_t)rows * c.q_value[l]));
    a.v = wk.f32(rows);
  };
  QNetAct net[3], gact, aact;           // online(s), online(s'), target(s'); gradients; act path
  for (int n = 0; n < 3; ++n) alloc_net(net[n], B);
  alloc_net(gact, B);                   // same shapes: gradient w.r.t. each pre-activation
  q_td = wk.f32((int64_t)B * D); q_prio = wk.f32(B);
  const float* P = params;
  const int hdim = Lc > 0 ? c.q_common[Lc - 1] : c.obs_dim;

  // forward stages of one net
  auto fwd_stages = [&](const QNetP& W, const QNetAct& a, const float* x, int ldx, int rows,
                        std::vector<std::vector<IgemmProb>>& st_common, std::vector<std::vector<IgemmProb>>& st_hidden,
                        std::vector<IgemmProb>& st_out) {
    const float* in = x; int ldin = ldx, kin = c.obs_dim;
    st_common.resize(Lc);
    for (int k = 0; k < Lc; ++k) {
      st_common[k].push_back(dense_fwd(in, ldin, kin, nullptr, 0, 0, rows, P + W.cw[k], c.q_common[k], P + W.cb[k],
                                       a.zc[k], c.q_common[k], ACT_RELU));
      in = a.zc[k]; ldin = kin = c.q_common[k];
    }
    const float* h = in; const int ldh = ldin;
    st_hidden.resize(std::max(Lb, Lv));
    for (int br = 0; br < D; ++br) {
      const float* z = h; int ldz = ldh, kz = hdim;
      for (int l = 0; l < Lb; ++l) {
        st_hidden[l].push_back(dense_fwd(z, ldz, kz, nullptr, 0, 0, rows, P + W.bw[br][l], c.q_branch[l],
                                         P + W.bb[br][l], a.zb[br][l], c.q_branch[l], ACT_RELU));
        z = a.zb[br][l]; ldz = kz = c.q_branch[l];
      }
      st_out.push_back(dense_fwd(z, ldz, kz, nullptr, 0, 0, rows, P + W.bw[br][Lb], nb, P + W.bb[br][Lb],
                                 a.adv + br * nb, D * nb, ACT_NONE));
    }
    const float* z = h; int ldz = ldh, kz = hdim;
    for (int l = 0; l < Lv; ++l) {
      st_hidden[l].push_back(dense_fwd(z, ldz, kz, nullptr, 0, 0, rows, P + W.vw[l], c.q_value[l], P + W.vb[l],
                                       a.zv[l], c.q_value[l], ACT_RELU));
      z = a.zv[l]; ldz = kz = c.q_value[l];
    }
    st_out.push_back(dense_fwd(z, ldz, kz, nullptr, 0, 0, rows, P + W.vw[Lv], 1, P + W.vb[Lv], a.v, 1, ACT_NONE));
  };

  // =============================================================== prioritised replay (per_kernels.h)
  per_on = c.q_per != 0;
  if (per_on) {
    memset(&per, 0, sizeof(per));
    per_blocks = (int)((cap + PER_BLK - 1) / PER_BLK);
    per.p = (double*)rp.take((size_t)cap * 8);
    per.bsum = (double*)rp.take((size_t)per_blocks * 8);
    per.bmin = (double*)rp.take((size_t)per_blocks * 8);
    per.st = (PerState*)rp.take(sizeof(PerState));
    per_u = (double*)wk.take((size_t)B * 8);
    per.sc = sc; per.seed = c.seed; per.B = B; per.alpha = c.q_per_alpha; per.eps = c.q_per_eps;
    per.alpha64 = c.q_per_alpha64 != 0.0 ? c.q_per_alpha64 : (double)c.q_per_alpha;
    per.stratified = c.q_per_stratified != 0;
    per.idx_out = idx_buf; per.w_out = eps_buf; per.prio_in = nullptr;   // set below (q_prio)
    for (int mode = 0; mode < 2; ++mode) {
      PerArgs pa = per;
      pa.u = mode ? per_u : nullptr;
      const int nb = per_blocks;
      grl_ctx* self = this;
      for (int with_gather = 0; with_gather < 2; ++with_gather) {
        Op op; op.tag = "per_sample";
        op.run = [self, pa, nb, mode, with_gather, q_ga, q_defer](hipStream_t s) {
          PerArgs q = pa;
          q.prio_in = self->q_prio;
          GatherArgs g = *q_ga;
          g.adam_tick = *q_defer;
          hipLaunchKernelGGL(per_blocksum_kernel, dim3(nb), dim3(256), 0, s, q);
          hipLaunchKernelGGL(per_sample_kernel, dim3(q.B), dim3(256), 0, s, q, nb, g, with_gather);   // RNG mode: marks rng_used, q_loss ticks
        };
        (with_gather ? (mode ? ops_per_u_g : ops_per_rng_g) : (mode ? ops_per_u : ops_per_rng)).push_back(op);
      }
    }
    {
      grl_ctx* self = this;
      Op op; op.tag = "per_update";
      op.run = [self](hipStream_t s) {
        PerArgs q = self->per;
        q.prio_in = self->q_prio;
        hipLaunchKernelGGL(per_update_kernel, dim3(1), dim3(256), 0, s, q, (const int64_t*)self->idx_buf);
      };
      ops_per_update.push_back(op);
    }
  }

  // =============================================================== RNG (uniform indices; weights = 1)
  {
    Op op; op.tag = "rng";
    RngArgs ra{sc, c.seed, B, 1, idx_buf, eps_buf, eps_buf, 1};   // weights = 1; the loss kernel advances rng_step
    op.run = [ra](hipStream_t s) {
      hipLaunchKernelGGL(rng_kernel, dim3((ra.B + 255) / 256), dim3(256), 0, s, ra);
    };
    ops_rng.push_back(op);
  }
  // =============================================================== forward
  {
    GatherArgs ga;
    memset(&ga, 0, sizeof(ga));
    ga.idx = idx_buf; ga.B = B; ga.img_elems = img_elems; ga.n_direct = 0; ga.act_dim = A;
    ga.rp_obs = rp_obs; ga.rp_next = rp_next; ga.rp_dobs = rp_dobs; ga.rp_dnext = rp_dnext;
    ga.rp_act = rp_act; ga.rp_rew = rp_rew; ga.rp_done = rp_done;
    ga.mean = s_mean; ga.stdv = s_std; ga.dmean = s_dmean; ga.dstd = s_dstd; ga.ret_std = s_ret;
    ga.normalize = (c.normalize == 1 || c.normalize == 2); ga.normalize_rew = (c.normalize == 1 || c.normalize == 3);
    ga.clip_obs = c.clip_obs; ga.clip_rew = c.clip_reward; ga.scale_div = 1.f;
    ga.x_obs = feat[0]; ga.x_obs2 = nullptr; ga.x_next = feat[2]; ga.ldx = ldf;
    ga.d_obs0 = ga.d_obs1 = ga.d_next = feat[0]; ga.ldd = ldf;
    ga.act_out = act; ga.ld_act = A; ga.rew_out = rew; ga.done_out = done;
    ga.sc = sc;
    *q_ga = ga;
    Op op; op.tag = "gather_norm";
    op.run = [ga, q_defer](hipStream_t s) {
      GatherArgs g2 = ga;
      g2.adam_tick = *q_defer;        // deferred loss sums: the Adam step size of the update is fixed here, as in the SAC plan
      hipLaunchKernelGGL(gather_norm_kernel, dim3((g2.img_elems + 255) / 256, g2.B, 2), dim3(256), 0, s, g2);
    };
    ops_grads.push_back(op);
  }
  // ---- row-local chains (q_kernels.h) when every width fits the head primitives; else one GEMM launch per layer
  bool fused_q = false;
  {
    const char* nf = getenv("GRL_NO_FUSED_Q");
    bool ok = !(nf && nf[0] == '1') && nb <= 64 && Lc + std::max(Lb, Lv) <= GRL_MAX_LAYERS;
    for (int k = 0; k < Lc; ++k) ok = ok && c.q_common[k] <= HT_MAXW;
    for (int l = 0; l < Lb; ++l) ok = ok && c.q_branch[l] <= HT_MAXW;
    for (int l = 0; l < Lv; ++l) ok = ok && c.q_value[l] <= HT_MAXW;
    if (Lc > 0) ok = ok && c.q_common[Lc - 1] <= HT_MAXA;
    fused_q = ok;
  }
  QFusedArgs qf;
  memset(&qf, 0, sizeof(qf));
  if (fused_q) {
    const QNetP* Wn[3] = {&Pon, &Pon, &Ptg};
    const float* xin[3] = {feat[0], feat[2], feat[2]};
    auto tower_w = [&](const QNetP& W, int tw, int l) { return P + (tw < D ? W.bw[tw][l] : W.vw[l]); };
    auto tower_b = [&](const QNetP& W, int tw, int l) { return P + (tw < D ? W.bb[tw][l] : W.vb[l]); };
    auto tower_hid = [&](int tw, int l) { return tw < D ? c.q_branch[l] : c.q_value[l]; };
    auto tower_z = [&](const QNetAct& a, int tw, int l) { return tw < D ? a.zb[tw][l] : a.zv[l]; };
    std::vector<HtHead> hf, hb;
    std::vector<IgemmProb> l0;
    for (int n = 0; n < 3; ++n) {
      const QNetP& W = *Wn[n];
      const QNetAct& a = net[n];
      float* u_trunk = nullptr;
      if (Lc > 0) {   // layer 0 of the trunk: one GEMM (K = obs_dim), no bias / activation (applied by the chain)
        u_trunk = wk.f32((int64_t)B * c.q_common[0]);
        l0.push_back(dense_fwd(xin[n], ldf, c.obs_dim, nullptr, 0, 0, B, P + W.cw[0], c.q_common[0], nullptr, u_trunk,
                               c.q_common[0], ACT_NONE));
      }
      for (int tw = 0; tw <= D; ++tw) {
        const int Lt = tw < D ? Lb : Lv;
        HtHead h;
        memset(&h, 0, sizeof(h));
        int li = 0;
        if (Lc > 0) {
          h.u = u_trunk; h.ldu = c.q_common[0]; h.b0 = P + W.cb[0]; h.H0 = c.q_common[0];
          h.z0 = tw == 0 ? a.zc[0] : nullptr;             // the trunk is recomputed per tower, stored once
          h.hid[0] = c.q_common[0];
          for (li = 1; li < Lc; ++li) {
            h.w[li] = P + W.cw[li]; h.b[li] = P + W.cb[li]; h.hid[li] = c.q_common[li];
            h.z[li] = tw == 0 ? a.zc[li] : nullptr;
          }
          for (int l = 0; l < Lt; ++l, ++li) {
            h.w[li] = tower_w(W, tw, l); h.b[li] = tower_b(W, tw, l); h.hid[li] = tower_hid(tw, l); h.z[li] = tower_z(a, tw, l);
          }
        } else {      // no trunk: layer 0 of every tower from the GEMM launch
          float* u = wk.f32((int64_t)B * tower_hid(tw, 0));
          l0.push_back(dense_fwd(xin[n], ldf, c.obs_dim, nullptr, 0, 0, B, tower_w(W, tw, 0), tower_hid(tw, 0), nullptr, u,
                                 tower_hid(tw, 0), ACT_NONE));
          h.u = u; h.ldu = tower_hid(tw, 0); h.b0 = tower_b(W, tw, 0); h.H0 = tower_hid(tw, 0);
          h.z0 = tower_z(a, tw, 0); h.hid[0] = h.H0;
          for (li = 1; li < Lt; ++li) {
            h.w[li] = tower_w(W, tw, li); h.b[li] = tower_b(W, tw, li); h.hid[li] = tower_hid(tw, li); h.z[li] = tower_z(a, tw, li);
          }
        }
        h.L = li;
        h.n_out = 1; h.out_dim = tw < D ? nb : 1;
        h.ow[0] = tower_w(W, tw, Lt); h.ob[0] = tower_b(W, tw, Lt);
        h.out[0] = tw < D ? a.adv + tw * nb : a.v;
        h.ld_out = tw < D ? D * nb : 1;
        hf.push_back(h);
      }
    }
    // backward views of the online net on s
    const QNetAct& a = net[0];
    for (int tw = 0; tw <= D; ++tw) {
      const int Lt = tw < D ? Lb : Lv;
      HtHead h;
      memset(&h, 0, sizeof(h));
      h.H0 = tower_hid(tw, 0); h.L = Lt; h.hid[0] = h.H0;
      h.z0 = tower_z(a, tw, 0); h.g0 = tower_z(gact, tw, 0); h.ldg0 = h.H0;
      if (Lc > 0) { h.n_xa = c.q_common[Lc - 1]; h.w0a = tower_w(Pon, tw, 0); }
      for (int l = 1; l < Lt; ++l) {
        h.w[l] = tower_w(Pon, tw, l); h.hid[l] = tower_hid(tw, l); h.z[l] = tower_z(a, tw, l); h.g[l] = tower_z(gact, tw, l);
      }
      h.n_out = 1; h.out_dim = tw < D ? nb : 1; h.ow[0] = tower_w(Pon, tw, Lt);
      hb.push_back(h);
    }
    qf.fwd = upload_vec(wk, hf);
    qf.bwd_tw = upload_vec(wk, hb);
    qf.B = B; qf.D = D; qf.nb = nb; qf.Ht = Lc > 0 ? c.q_common[Lc - 1] : 0;
    qf.d_adv = gact.adv; qf.d_v = gact.v; qf.trunk_scale = c.q_trunk_scale;
    if (Lc > 0) {
      HtHead h;
      memset(&h, 0, sizeof(h));
      h.H0 = c.q_common[0]; h.L = Lc; h.z0 = a.zc[0]; h.g0 = gact.zc[0]; h.ldg0 = h.H0; h.hid[0] = h.H0;
      for (int l = 1; l < Lc; ++l) { h.w[l] = P + Pon.cw[l]; h.hid[l] = c.q_common[l]; h.z[l] = a.zc[l]; h.g[l] = gact.zc[l]; }
      qf.bwd_tr = upload_vec(wk, std::vector<HtHead>{h});
      qf.dh_part = wk.f32((int64_t)(D + 1) * B * qf.Ht);
    }
    add_launch(ops_grads, "q_l0", 0, l0);
    Op op; op.tag = "q_fwd";
    const QFusedArgs fa = qf;
    op.run = [fa](hipStream_t s) {
      hipLaunchKernelGGL(q_fwd_fused_kernel, dim3((fa.B + HT_RB - 1) / HT_RB, 3, fa.D + 1), dim3(256), 0, s, fa);
    };
    ops_grads.push_back(op);
  } else {
    std::vector<std::vector<IgemmProb>> sc_[3], sh_[3];
    std::vector<IgemmProb> so_[3];
    fwd_stages(Pon, net[0], feat[0], ldf, B, sc_[0], sh_[0], so_[0]);
    fwd_stages(Pon, net[1], feat[2], ldf, B, sc_[1], sh_[1], so_[1]);
    fwd_stages(Ptg, net[2], feat[2], ldf, B, sc_[2], sh_[2], so_[2]);
    auto merged = [&](std::vector<IgemmProb> a, const std::vector<IgemmProb>& b, const std::vector<IgemmProb>& d) {
      a.insert(a.end(), b.begin(), b.end()); a.insert(a.end(), d.begin(), d.end()); return a;
    };
    for (int k = 0; k < Lc; ++k) add_launch(ops_grads, "q_fwd", 0, merged(sc_[0][k], sc_[1][k], sc_[2][k]));
    for (size_t l = 0; l < sh_[0].size(); ++l) add_launch(ops_grads, "q_fwd", 0, merged(sh_[0][l], sh_[1][l], sh_[2][l]));
    add_launch(ops_grads, "q_fwd", 0, merged(so_[0], so_[1], so_[2]));
  }
  {
    QLossArgs qa;
    qa.B = B; qa.D = D; qa.n = nb; qa.gamma = c.gamma; qa.lr = c.lr; qa.huber = c.q_huber; qa.double_q = c.q_double;
    qa.adv0 = net[0].adv; qa.v0 = net[0].v; qa.adv1 = net[1].adv; qa.adv2 = net[2].adv; qa.v2 = net[2].v;
    qa.act = act; qa.rew = rew; qa.done = done; qa.weights = eps_buf;
    qa.d_adv0 = gact.adv; qa.d_v0 = gact.v; qa.td = q_td; qa.priority = q_prio; qa.sc = sc;
    qa.row_part = wk.f32(3 * (int64_t)B);
    qa.counter = (unsigned*)wk.take(16);
    qa.defer_finish = 0;
    qa.loss_sum = (c.algo == GRL_ALGO_BDQ && c.q_loss_sum_branches) ? 1 : 0;
    zero_once.push_back({qa.counter, 16});
    q_row_part = qa.row_part;
#ifdef GRL_HOSTEMU
    q_finish = 1;
#else
    q_finish = qa.n <= 64 ? 1 : 0;     // (the one-workgroup fallback for > 64 bins forms its sums itself)
#endif
    Op op; op.tag = "q_loss";
    op.run = [qa, q_defer](hipStream_t s) {
      QLossArgs q2 = qa;
      q2.defer_finish = *q_defer;
#ifdef GRL_HOSTEMU
      hipLaunchKernelGGL(q_loss_kernel, dim3(1), dim3(256), 0, s, q2);
#else
      if (q2.n <= 64) hipLaunchKernelGGL(q_loss_kernel, dim3((q2.B + 3) / 4), dim3(256), 0, s, q2);
      else hipLaunchKernelGGL(q_loss_rows_kernel, dim3(1), dim3(256), 0, s, q2);
#endif
    };
    ops_grads.push_back(op);
  }
  // =============================================================== backward (online net on s)
  {
    const QNetAct& a = net[0];
    if (fused_q) {
      const QFusedArgs fa = qf;
      Op op; op.tag = "q_bwd";
      op.run = [fa](hipStream_t s) {
        hipLaunchKernelGGL(q_bwd_towers_kernel, dim3((fa.B + HT_RB - 1) / HT_RB, fa.D + 1), dim3(256), 0, s, fa);
        if (fa.bwd_tr) hipLaunchKernelGGL(q_bwd_trunk_kernel, dim3((fa.B + HT_RB - 1) / HT_RB), dim3(256), 0, s, fa);
      };
      ops_grads.push_back(op);
    } else {
      std::vector<IgemmProb> pr;      // output layers -> last hidden
      for (int br = 0; br < D; ++br)
        pr.push_back(dense_bwd({{gact.adv + br * nb, D * nb, nb, P + Pon.bw[br][Lb]}}, B, 0, c.q_branch[Lb - 1],
                               gact.zb[br][Lb - 1], c.q_branch[Lb - 1], a.zb[br][Lb - 1]));
      pr.push_back(dense_bwd({{gact.v, 1, 1, P + Pon.vw[Lv]}}, B, 0, c.q_value[Lv - 1], gact.zv[Lv - 1], c.q_value[Lv - 1],
                             a.zv[Lv - 1]));
      add_launch(ops_grads, "q_bwd", 1, pr);
      for (int l = std::max(Lb, Lv) - 1; l >= 1; --l) {
        std::vector<IgemmProb> p2;
        if (l < Lb)
          for (int br = 0; br < D; ++br)
            p2.push_back(dense_bwd({{gact.zb[br][l], c.q_branch[l], c.q_branch[l], P + Pon.bw[br][l]}}, B, 0,
                                   c.q_branch[l - 1], gact.zb[br][l - 1], c.q_branch[l - 1], a.zb[br][l - 1]));
        if (l < Lv)
          p2.push_back(dense_bwd({{gact.zv[l], c.q_value[l], c.q_value[l], P + Pon.vw[l]}}, B, 0, c.q_value[l - 1],
                                 gact.zv[l - 1], c.q_value[l - 1], a.zv[l - 1]));
        add_launch(ops_grads, "q_bwd", 1, p2);
      }
      if (Lc > 0) {   // into the shared trunk: sum over the D+1 towers in chunks of three reduction parts
        std::vector<BwdPart> towers;
        for (int br = 0; br < D; ++br) towers.push_back({gact.zb[br][0], c.q_branch[0], c.q_branch[0], P + Pon.bw[br][0]});
        towers.push_back({gact.zv[0], c.q_value[0], c.q_value[0], P + Pon.vw[0]});
        for (size_t t0 = 0; t0 < towers.size(); t0 += 3) {
          std::vector<BwdPart> chunk(towers.begin() + t0, towers.begin() + std::min(towers.size(), t0 + 3));
          const bool last = t0 + 3 >= towers.size();
          IgemmProb p = dense_bwd(chunk, B, 0, hdim, gact.zc[Lc - 1], hdim, last ? a.zc[Lc - 1] : nullptr);
          p.accumulate = t0 > 0 ? 1 : 0;
          p.out_scale = c.q_trunk_scale;
          add_launch(ops_grads, "q_bwd", 1, {p});
        }
        for (int k = Lc - 1; k >= 1; --k)
          add_launch(ops_grads, "q_bwd", 1,
                     {dense_bwd({{gact.zc[k], c.q_common[k], c.q_common[k], P + Pon.cw[k]}}, B, 0, c.q_common[k - 1],
                                gact.zc[k - 1], c.q_common[k - 1], a.zc[k - 1])});
      }
    }
    // weight gradients
    std::vector<IgemmProb> wg;
    auto wgrad = [&](const float* x, int ldx, int kin, const float* g, int ldg, int n, int64_t woff, int64_t boff) {
      IgemmProb p = dense_wgrad(x, ldx, kin, true, g, ldg, n, B, nullptr, 1);
      p.c = wk.f32(p.slab_stride * p.split);
      add_wgrad(wg, p, woff, 0, kin, boff);
    };
    const float* in = feat[0]; int ldin = ldf, kin = c.obs_dim;
    for (int k = 0; k < Lc; ++k) {
      wgrad(in, ldin, kin, gact.zc[k], c.q_common[k], c.q_common[k], Pon.cw[k], Pon.cb[k]);
      in = a.zc[k]; ldin = kin = c.q_common[k];
    }
    for (int br = 0; br < D; ++br) {
      const float* z = in; int ldz = ldin, kz = kin;
      for (int l = 0; l < Lb; ++l) {
        wgrad(z, ldz, kz, gact.zb[br][l], c.q_branch[l], c.q_branch[l], Pon.bw[br][l], Pon.bb[br][l]);
        z = a.zb[br][l]; ldz = kz = c.q_branch[l];
      }
      wgrad(z, ldz, kz, gact.adv + br * nb, D * nb, nb, Pon.bw[br][Lb], Pon.bb[br][Lb]);
    }
    const float* z = in; int ldz = ldin, kz = kin;
    for (int l = 0; l < Lv; ++l) {
      wgrad(z, ldz, kz, gact.zv[l], c.q_value[l], c.q_value[l], Pon.vw[l], Pon.vb[l]);
      z = a.zv[l]; ldz = kz = c.q_value[l];
    }
    wgrad(z, ldz, kz, gact.v, 1, 1, Pon.vw[Lv], Pon.vb[Lv]);
    add_launch(ops_grads, "q_wgrad", 2, wg);
  }
  {
    std::vector<int2> rt = reduce_tiles();
    d_reduces = upload_vec(wk, reduces);
    int2* d_rt = upload_vec(wk, rt);
    const int ntiles = (int)rt.size();
    ReduceDesc* dr = d_reduces;
    Op op; op.tag = "reduce_slabs";
    LossArgs none;
    memset(&none, 0, sizeof(none));
    op.run = [dr, d_rt, ntiles, none](hipStream_t s) {
      hipLaunchKernelGGL(reduce_slabs_kernel, dim3(ntiles), dim3(256), 0, s, dr, d_rt, ntiles, none, 0, AdamArgs{}, 0);
    };
    ops_grads.push_back(op);
  }
  if (c.q_grad_clip > 0.f) {   // per-variable tf.clip_by_norm, after the data-parallel all-reduce point
    std::vector<VarSeg> segs;
    for (auto& v : vars)
      if (v.trainable) segs.push_back({v.off, v.numel});
    VarSeg* d_segs = upload_vec(wk, segs);
    const int nseg = (int)segs.size();
    float* g = grads; const float clip = c.q_grad_clip;
    Op op; op.tag = "clip_by_norm";
    op.run = [g, d_segs, nseg, clip](hipStream_t s) {
      hipLaunchKernelGGL(clip_by_norm_kernel, dim3(nseg), dim3(256), 0, s, g, d_segs, clip);
    };
    ops_apply.push_back(op);
  }
  {
    Op op; op.tag = "adam";
    grl_ctx* self = this;
    op.run = [self](hipStream_t s) {
      AdamArgs aa;
      aa.params = self->params; aa.grads = self->grads; aa.m = self->adam_m; aa.v = self->adam_v;
      aa.n_train = self->n_train; aa.sc = self->sc; aa.grad_scale = self->grad_scale; aa.tau = 0.f; aa.eps = 1e-8f;
      aa.src_ofs = 0; aa.n_polyak = 0; aa.target = self->params + self->tgt_off;
      const int blocks = (int)std::min<int64_t>(2048, (self->n_train + 255) / 256);
      hipLaunchKernelGGL(adam_polyak_kernel, dim3(blocks), dim3(256), 0, s, aa);
    };
    ops_apply.push_back(op);
  }
  {
    // Full updates: reduction + clip + Adam as one launch (q_reduce_clip_adam_kernel) when every trainable variable is
    // exactly one reduction descriptor and fits the kernel's LDS buffer.  GRL_NO_FUSED_QAPPLY=1 keeps the three launches.
    const char* nf = getenv("GRL_NO_FUSED_QAPPLY");
    bool ok = !(nf && atoi(nf)) && !ops_grads.empty() && ops_grads.back().tag == "reduce_slabs";
    size_t n_tr = 0;
    for (auto& v : vars) {
      if (!v.trainable) continue;
      ++n_tr;
      int hits = 0;
      for (auto& r : reduces) hits += (r.dst == grads + v.off && r.n == v.numel && r.n <= GRL_QAPPLY_MAX) ? 1 : 0;
      ok = ok && hits == 1;
    }
    ok = ok && n_tr == reduces.size();
    for (auto& r : reduces) ok = ok && r.row_len == 0;     // the kernel reads plain (non-strided) slabs
#ifndef GRL_HOSTEMU
    if (ok) {   // ~71 KB of static LDS per workgroup: fits gfx950's 160 KB; any device that offers less keeps the three launches
      int dev = 0, lds = 0;
      if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess ||
          lds < (int)(GRL_QAPPLY_MAX * 4 + 1024 * 4 + 3 * 256 * 4))
        ok = false;
    }
#endif
    if (getenv("GRL_PLAN_DUMP")) fprintf(stderr, "grl plan: q_apply       reduction + clip + Adam in one launch: %s (%zu variables)\n", ok ? "yes" : "no", n_tr);
    if (ok) {
      ops_grads_apply.assign(ops_grads.begin(), ops_grads.end() - 1);
      Op op; op.tag = "q_apply";
      grl_ctx* self = this;
      const ReduceDesc* dr = d_reduces;
      const int nd = (int)reduces.size();
      const float clip = c.q_grad_clip;
      const float* rp = q_row_part; const int rows = B, fin = q_finish;
      auto apply_op = [self, dr, nd, clip, rp, rows, fin](bool with_per) {
        return [self, dr, nd, clip, rp, rows, fin, with_per](hipStream_t s) {
          AdamArgs aa;
          aa.params = self->params; aa.grads = self->grads; aa.m = self->adam_m; aa.v = self->adam_v;
          aa.n_train = self->n_train; aa.sc = self->sc; aa.grad_scale = self->grad_scale; aa.tau = 0.f; aa.eps = 1e-8f;
          aa.src_ofs = 0; aa.n_polyak = 0; aa.target = self->params + self->tgt_off;
          PerArgs q = self->per;
          q.prio_in = self->q_prio;
          // prioritised replay: one more workgroup writes the new priorities back (per_update_kernel's work)
          hipLaunchKernelGGL(q_reduce_clip_adam_kernel, dim3(nd + (with_per ? 2 : 1)), dim3(1024), 0, s, dr, nd, clip, aa, rp, rows, fin,
                             q, (const int64_t*)self->idx_buf);
        };
      };
      op.run = apply_op(false);
      ops_grads_apply.push_back(op);
      if (per_on && B <= 1024) {
        for (size_t k = 0; k + 1 < ops_grads_apply.size(); ++k)
          if (ops_grads_apply[k].tag != "gather_norm") ops_grads_apply_per.push_back(ops_grads_apply[k]);   // (the sampler gathers its rows)
        Op po; po.tag = "q_apply";
        po.run = apply_op(true);
        ops_grads_apply_per.push_back(po);
      }
      *q_defer = 1;
      if (fin) {     // split compute / apply path of this plan: the batch means as a launch of their own, behind the reduction
        Op fo; fo.tag = "q_finish";
        DevScalars* scp = sc;
        fo.run = [scp, rp, rows](hipStream_t s) { hipLaunchKernelGGL(q_finish_kernel, dim3(1), dim3(256), 0, s, scp, rp, rows); };
        ops_grads.push_back(fo);
      }
    }
  }
  // =============================================================== act path: Q-values of n observations
  {
    afeat = wk.f32((int64_t)NA * ldf);
    q_aout = wk.f32((int64_t)NA * D * nb);
    a_eps = wk.f32(NA); a_out = q_aout;
    alloc_net(aact, NA);
    ActIngestArgs ia;
    memset(&ia, 0, sizeof(ia));
    ia.obs = stg_obs; ia.n = NA; ia.vec_dim = c.obs_dim; ia.scale_div = 1.f; ia.x = afeat; ia.ldx = ldf; ia.d = afeat; ia.ldd = ldf;
    Op op; op.tag = "act_ingest";
    const int elems = c.obs_dim;
    op.run = [ia, elems](hipStream_t s) {
      hipLaunchKernelGGL(act_ingest_kernel, dim3((elems + 255) / 256, ia.n), dim3(256), 0, s, ia);
    };
    ops_act.push_back(op);
    std::vector<std::vector<IgemmProb>> sc1, sh1;
    std::vector<IgemmProb> so1;
    fwd_stages(Pon, aact, afeat, ldf, NA, sc1, sh1, so1);
    for (auto& v : sc1) add_launch(ops_act, "act_q", 0, v);
    for (auto& v : sh1) add_launch(ops_act, "act_q", 0, v);
    add_launch(ops_act, "act_q", 0, so1);
    const float* adv = aact.adv; const float* vv = aact.v; float* qo = q_aout; const int rows = NA, Dq = D, nq = nb;
    Op op2; op2.tag = "dueling";
    op2.run = [adv, vv, qo, rows, Dq, nq](hipStream_t s) {
      hipLaunchKernelGGL(dueling_kernel, dim3((rows * Dq + 255) / 256), dim3(256), 0, s, adv, vv, rows, Dq, nq, qo);
    };
    ops_act.push_back(op2);
  }
  for (int k = 0; k < 8; ++k) enc_w[k] = nullptr;
  dbg["feat_pi"] = {feat[0], (int64_t)B * ldf};
  dbg["feat_tgt"] = {feat[2], (int64_t)B * ldf};
  dbg["adv"] = {net[0].adv, (int64_t)B * D * nb};
  dbg["v"] = {net[0].v, B};
  dbg["adv_next"] = {net[1].adv, (int64_t)B * D * nb};
  dbg["adv_tgt"] = {net[2].adv, (int64_t)B * D * nb};
  dbg["v_tgt"] = {net[2].v, B};
  dbg["td"] = {q_td, (int64_t)B * D};
  dbg["idx_raw"] = {(const float*)idx_buf, (int64_t)2 * B};     // int64 viewed as float pairs
  dbg["weights"] = {eps_buf, B};
  if (per_on) dbg["per_p"] = {(const float*)per.p, 2 * cap};   // float64 leaves, handed out as raw 4-byte words
  dbg["priority"] = {q_prio, B};
  dbg["rew"] = {rew, B}; dbg["done"] = {done, B}; dbg["act"] = {act, (int64_t)B * A};
  dbg["grads"] = {grads, n_train};
  dbg["adam_m"] = {adam_m, n_train};
  dbg["adam_v"] = {adam_v, n_train};
  return GRL_OK;
}


// --------------------------------------------------------------------------------------------------
// Depth auto-encoder training: encoders.py:90-124 network, :127 MSE, :130 Adam,
// config/encoder.yaml (7/5/3 kernels, 32 filters, stride 2, encoding 100, lr 2e-4, batch 128).
// Parameters are the 16 Keras tensors in model.h5 order; one step = forward, MSE, backward, Keras-Adam.
int grl_ctx::plan_ae() {
  const grl_config& c = cfg;
  cnn = false;
  B = c.batch_size; NA = std::max(1, c.act_batch); A = 1; L = 0;
  img_elems = 4096; F = 100; Fc = 0; ldf = 100; C_img = 1; hw = 64;
  const float LA = 0.1f;   // LeakyReLU alpha (encoders.py:87)
  // ---------------- parameter layout (Keras creation order == HDF5 order of the shipped model.h5)
  struct CL { const char* name; int kh, cin, cout; };
  const CL enc[3] = {{"encoder/conv2d_1", 7, 1, 32}, {"encoder/conv2d_2", 5, 32, 32}, {"encoder/conv2d_3", 3, 32, 32}};
  const CL dec[3] = {{"decoder/conv2d_4", 3, 32, 32}, {"decoder/conv2d_5", 5, 32, 32}, {"decoder/conv2d_6", 7, 32, 1}};
  int64_t ew[3], eb[3], dw[3], db[3];
  for (int l = 0; l < 3; ++l) {
    ew[l] = add_var(std::string(enc[l].name) + "/kernel", {enc[l].kh, enc[l].kh, enc[l].cin, enc[l].cout}, true);
    eb[l] = add_var(std::string(enc[l].name) + "/bias", {enc[l].cout}, true);
  }
  const int64_t edw = add_var("encoder/dense_1/kernel", {2048, 100}, true), edb = add_var("encoder/dense_1/bias", {100}, true);
  const int64_t ddw = add_var("decoder/dense_2/kernel", {100, 2048}, true), ddb = add_var("decoder/dense_2/bias", {2048}, true);
  for (int l = 0; l < 3; ++l) {
    dw[l] = add_var(std::string(dec[l].name) + "/kernel", {dec[l].kh, dec[l].kh, dec[l].cin, dec[l].cout}, true);
    db[l] = add_var(std::string(dec[l].name) + "/bias", {dec[l].cout}, true);
  }
  n_train = n_params; tgt_off = n_params; vf_off = 0; n_polyak = 0; ent_off = 0;
  // ---------------- arenas
  params = st.f32(n_params);
  adam_m = st.f32(n_train);
  adam_v = st.f32(n_train);
  sc = (DevScalars*)st.take(sizeof(DevScalars));
  s_mean = (double*)st.take(8); s_std = (double*)st.take(8); s_dmean = (double*)st.take(8); s_dstd = (double*)st.take(8);
  s_ret = (double*)st.take(8);
  grads = gr.f32(n_train);
  rp_obs = rp_next = rp_dobs = rp_dnext = rp_act = rp_rew = rp_done = rp.f32(4);   // no replay on this path
  stg_n = std::max(NA, 64);
  stg_obs = stg_next = stg_act = stg_rew = stg_done = wk.f32(4);
  idx_buf = (int64_t*)wk.take(8); eps_buf = wk.f32(4);
  for (int n = 0; n < 3; ++n) feat[n] = wk.f32(4);
  const float* P = params;
  // activations [B, H, W, C] (NHWC) and their gradients
  auto T = [&](int h, int ch) { return wk.f32((int64_t)B * h * h * ch); };
  ae_x = T(64, 1);
  float *e1 = T(32, 32), *e2 = T(16, 32), *e3 = T(8, 32), *z = wk.f32((int64_t)B * 100), *dh = T(8, 32);
  float *d4 = T(16, 32), *d5 = T(32, 32), *u6 = T(64, 32), *out = T(64, 1);
  // inputs of the padded convolutions live in zero-bordered buffers (border written once): the 'same'
  // convolution becomes a 'valid' one over the bordered image, so neither the forward GEMM nor the weight
  // gradient needs per-tap bounds masks and both run on the vectorised kernel
  auto TP = [&](int h, int lo, int hi, int ch) {
    const int64_t n = (int64_t)B * (h + lo + hi) * (h + lo + hi) * ch;
    float* b = wk.f32(n);
    zero_once.push_back({b, (size_t)n * 4});
    return b;
  };
  float *x_p = TP(64, 2, 3, 1), *e1_p = TP(32, 1, 2, 32), *e2_p = TP(16, 0, 1, 32);
  float *u4 = TP(16, 1, 1, 32), *u5 = TP(32, 2, 2, 32);
  float* g_pad = nullptr;
  float *g_out = T(64, 1), *g_u6 = T(64, 32), *g_d5 = T(32, 32), *g_u5 = T(32, 32), *g_d4 = T(16, 32), *g_u4 = T(16, 32);
  float *g_dh = T(8, 32), *g_z = wk.f32((int64_t)B * 100), *g_e3 = T(8, 32), *g_e2 = T(16, 32), *g_e1 = T(32, 32);
  const int NPART = 256;
  float* partial = wk.f32(NPART);
  // geometry: encoder convs 'SAME' stride 2 (TF asymmetric padding: low pad 2 / 1 / 0), decoder convs 'SAME' stride 1
  const ConvGeom ge[3] = {{64, 64, 1, 7, 7, 2, 2, 32, 32, 32}, {32, 32, 32, 5, 5, 2, 1, 16, 16, 32}, {16, 16, 32, 3, 3, 2, 0, 8, 8, 32}};
  const ConvGeom gd[3] = {{16, 16, 32, 3, 3, 1, 1, 16, 16, 32}, {32, 32, 32, 5, 5, 1, 2, 32, 32, 32}, {64, 64, 32, 7, 7, 1, 3, 64, 64, 1}};
  // the same convolutions as 'valid' ones over the bordered inputs (forward + weight gradient)
  const ConvGeom gev[3] = {{69, 69, 1, 7, 7, 2, 0, 32, 32, 32}, {35, 35, 32, 5, 5, 2, 0, 16, 16, 32}, {17, 17, 32, 3, 3, 2, 0, 8, 8, 32}};
  const ConvGeom gdv[2] = {{18, 18, 32, 3, 3, 1, 0, 16, 16, 32}, {36, 36, 32, 5, 5, 1, 0, 32, 32, 32}};
  ConvFwdTabs fte[3], ftd[2];
  for (int l = 0; l < 3; ++l) fte[l] = conv_fwd_tabs(gev[l], B);
  for (int l = 0; l < 2; ++l) ftd[l] = conv_fwd_tabs(gdv[l], B);
  auto elem = [&](const char* tag, std::function<void(hipStream_t)> f) {
    Op op; op.tag = tag; op.run = std::move(f);
    ops_ae.push_back(op);
  };
  auto up = [&](const float* h, float* u, int H, int border) {
    const int Bn = B;
    elem("ae_upsample", [=](hipStream_t s) {
      const long quads = (long)Bn * 2 * H * 2 * H * 8;
      hipLaunchKernelGGL(upsample2_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, s, h, u, Bn, H, H, 32,
                         border, border);
    });
  };
  auto padcp = [&](const float* x, float* xp, int H, int C, int lo, int hi) {
    const long total = (long)B * H * H * C;
    elem("ae_pad_copy", [=](hipStream_t s) {
      hipLaunchKernelGGL(pad_copy_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, xp, total, H, H, C, lo, hi);
    });
  };
  auto up_bwd = [&](const float* gu, const float* h, float* gh, int H) {
    const int Bn = B;
    elem("ae_upsample_bwd", [=](hipStream_t s) {
      const long n = (long)Bn * H * H * 32;
      hipLaunchKernelGGL(upsample2_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, gu, h, gh, Bn, H, H, 32, LA);
    });
  };
  // =============================================================== forward
  {
    const float* in[3] = {x_p, e1_p, e2_p};
    float* o[3] = {e1, e2, e3};
    padcp(ae_x, x_p, 64, 1, 2, 3);
    for (int l = 0; l < 3; ++l) {
      add_launch(ops_ae, "ae_enc_conv", 0, {conv_fwd(in[l], fte[l], gev[l], P + ew[l], P + eb[l], o[l], ACT_LEAKY, LA)});
      if (l == 0) padcp(e1, e1_p, 32, 32, 1, 2);
      if (l == 1) padcp(e2, e2_p, 16, 32, 0, 1);
    }
  }
  {
    IgemmProb p = dense_fwd(e3, 2048, 2048, nullptr, 0, 0, B, P + edw, 100, P + edb, z, 100, ACT_LEAKY);
    p.act_alpha = LA;
    add_launch(ops_ae, "ae_dense", 0, {p});
    IgemmProb q = dense_fwd(z, 100, 100, nullptr, 0, 0, B, P + ddw, 2048, P + ddb, dh, 2048, ACT_LEAKY);
    q.act_alpha = LA;
    add_launch(ops_ae, "ae_dense", 0, {q});
  }
  up(dh, u4, 8, 1);
  add_launch(ops_ae, "ae_dec_conv", 0, {conv_fwd(u4, ftd[0], gdv[0], P + dw[0], P + db[0], d4, ACT_LEAKY, LA)});
  up(d4, u5, 16, 2);
  add_launch(ops_ae, "ae_dec_conv", 0, {conv_fwd(u5, ftd[1], gdv[1], P + dw[1], P + db[1], d5, ACT_LEAKY, LA)});
  up(d5, u6, 32, 0);
  // output conv (7x7 'same', 32 -> 1): N = 1 wastes the matrix cores, so T[tap, p] = W[tap, :] . u6[p, :] as a
  // GEMM with M = 49, then a 49-tap gather-sum (ae_kernels.h: ae_tapsum_kernel)
  const long ldT = (long)B * 4096;
  float* Tt = wk.f32(49 * ldT);
  add_launch(ops_ae, "ae_out_conv", 1, {dense_bwd({{P + dw[2], 32, 32, u6}}, 49, 0, B * 4096, Tt, (int)ldT, nullptr)});
  {
    const long npix = (long)B * 4096;
    const float* b6 = P + db[2];
    elem("ae_out_tapsum", [=](hipStream_t s) {
      hipLaunchKernelGGL(ae_tapsum_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, s, (const float*)Tt, ldT, b6, out, npix);
    });
  }
  ae_out = out;
  ops_ae_fwd = ops_ae;            // everything so far: the forward pass (Model.predict / evaluate)
  // =============================================================== loss
  {
    g_pad = wk.f32((int64_t)B * 4900);
    zero_once.push_back({g_pad, (size_t)B * 4900 * 4});   // the 3-pixel border stays zero
    float* partial_g = wk.f32(NPART);
    MseArgs ma{out, ae_x, g_out, partial, (long)B * 4096, g_pad, partial_g};
    const float lr = c.lr;
    DevScalars* scp = sc;
    float* gb6 = grads + db[2];
    elem("ae_mse", [=](hipStream_t s) {
      hipLaunchKernelGGL(mse_kernel, dim3(NPART), dim3(256), 0, s, ma);
      hipLaunchKernelGGL(ae_finish_kernel, dim3(1), dim3(64), 0, s, (const float*)partial, (const float*)partial_g, NPART,
                         ma.n_total, lr, scp, gb6);
    });
  }
  // =============================================================== backward
  std::vector<IgemmProb> wgc, wgd;
  auto cw = [&](const float* x, const ConvFwdTabs& t, const ConvGeom& g, const float* gy, int64_t w_off, int64_t b_off, int split) {
    IgemmProb p = conv_wgrad(x, t, g, gy, nullptr, split);
    p.c = wk.f32(p.slab_stride * p.split);
    add_wgrad(wgc, p, w_off, 0, g.K(), b_off);
  };
  auto cb = [&](const char* tag, const float* gy, const ConvGeom& g, const float* w, float* dx, const float* mask) {
    std::vector<IgemmProb> pr;
    for (auto& cl : conv_bwd_tabs(g, B)) {
      IgemmProb p = conv_bwd(gy, cl, g, w, dx, mask);
      p.act_alpha = LA;                                  // LeakyReLU gradient where a mask is given
      pr.push_back(p);
    }
    add_launch(ops_ae, tag, 1, pr);
  };
  // output conv (7x7, 32 -> 1): dW[tap, c] = sum_p g[p - shift(tap)] u6[p, c] -- a GEMM with M = 49 taps, N = 32
  // channels, K = pixels; g is read from its zero-bordered copy so that no tap needs a mask.  (Its bias
  // gradient, sum g, comes from the MSE kernel.)
  {
    std::vector<int32_t> ti(49), tr((size_t)B * 4096);
    for (int kh = 0; kh < 7; ++kh)
      for (int kw = 0; kw < 7; ++kw) ti[kh * 7 + kw] = -((kh - 3) * 70 + (kw - 3));
    for (int n = 0; n < B; ++n)
      for (int oh = 0; oh < 64; ++oh)
        for (int ow = 0; ow < 64; ++ow) tr[((size_t)n * 64 + oh) * 64 + ow] = n * 4900 + (oh + 3) * 70 + (ow + 3);
    IgemmProb p = blank();
    p.M = 49; p.N = 32; p.K = B * 4096;
    p.p_base[0] = g_pad; p.p_tab_i = upload_vec(wk, ti); p.p_tab_r = upload_vec(wk, tr); single_part(p);
    p.q_base[0] = u6; p.q_ld_r[0] = 32; p.q_ld_j[0] = 1;
    p.ldc = 32;
    // 4-runs along the pixel index (rows are 64 pixels, quads never straddle one) at offsets that are only
    // 4-byte aligned: 16-byte buffer loads need no more than dword alignment
    // (measured: the gfx950 buffer_load_dwordx4 takes them, results match the oracle; 103 -> 50 us)
    p.vflags |= VF_P_TABS;
    set_split(p, 256);
    p.c = wk.f32(p.slab_stride * p.split);
    std::vector<IgemmProb> one;
    add_wgrad(one, p, dw[2], 0, 49, -1);
    add_launch(ops_ae, "ae_out_wgrad", 0, one);
  }
  {
    // backward-data of the output conv: g_u6[p, c] = sum_{kh,kw} g_pad[p - shift(kh,kw)] W[kh,kw,c], a GEMM with
    // M = pixels, N = 32, K = 7 x 8 taps (each kernel row flipped and padded to 8: ae_kernels.h) on the
    // vectorised kernel -- the taps of a quad are 4 neighbouring gradient pixels
    float* Wp = wk.f32(56 * 32);
    const float* W6 = P + dw[2];
    elem("ae_out_kernel_flip", [=](hipStream_t s) {
      hipLaunchKernelGGL(ae_out_kernel_flip, dim3((56 * 32 + 255) / 256), dim3(256), 0, s, W6, Wp, 32);
    });
    std::vector<int32_t> ti((size_t)B * 4096), tr(56);
    for (int n = 0; n < B; ++n)
      for (int ih = 0; ih < 64; ++ih)
        for (int iw = 0; iw < 64; ++iw) ti[((size_t)n * 64 + ih) * 64 + iw] = n * 4900 + (ih + 6) * 70 + (iw + 6);
    for (int kh = 0; kh < 7; ++kh)
      for (int j = 0; j < 8; ++j) tr[kh * 8 + j] = -kh * 70 - 7 + j;
    IgemmProb p = blank();
    p.M = B * 4096; p.N = 32; p.K = 56;
    p.p_base[0] = g_pad; p.p_tab_i = upload_vec(wk, ti); p.p_tab_r = upload_vec(wk, tr); single_part(p);
    p.vflags |= VF_P_TABS;                       // 4-runs along the taps, dword-aligned offsets
    p.q_base[0] = Wp; p.q_ld_r[0] = 32; p.q_ld_j[0] = 1;
    p.c = g_u6; p.ldc = 32;
    set_split(p, 1);
    add_launch(ops_ae, "ae_out_conv_bwd", 0, {p});
  }
  up_bwd(g_u6, d5, g_d5, 32);
  cw(u5, ftd[1], gdv[1], g_d5, dw[1], db[1], 32);
  cb("ae_dec_conv_bwd", g_d5, gd[1], P + dw[1], g_u5, nullptr);
  up_bwd(g_u5, d4, g_d4, 16);
  cw(u4, ftd[0], gdv[0], g_d4, dw[0], db[0], 8);
  cb("ae_dec_conv_bwd", g_d4, gd[0], P + dw[0], g_u4, nullptr);
  up_bwd(g_u4, dh, g_dh, 8);
  {
    IgemmProb p = dense_wgrad(z, 100, 100, true, g_dh, 2048, 2048, B, nullptr, 1);
    p.c = wk.f32(p.slab_stride * p.split);
    add_wgrad(wgd, p, ddw, 0, 100, ddb);
    IgemmProb b = dense_bwd({{g_dh, 2048, 2048, P + ddw}}, B, 0, 100, g_z, 100, z);
    b.act_alpha = LA;
    add_launch(ops_ae, "ae_dense_bwd", 1, {b});
    IgemmProb p2 = dense_wgrad(e3, 2048, 2048, true, g_z, 100, 100, B, nullptr, 1);
    p2.c = wk.f32(p2.slab_stride * p2.split);
    add_wgrad(wgd, p2, edw, 0, 2048, edb);
    IgemmProb b2 = dense_bwd({{g_z, 100, 100, P + edw}}, B, 0, 2048, g_e3, 2048, e3);
    b2.act_alpha = LA;
    add_launch(ops_ae, "ae_dense_bwd", 1, {b2});
  }
  cw(e2_p, fte[2], gev[2], g_e3, ew[2], eb[2], 4);
  cb("ae_enc_conv_bwd", g_e3, ge[2], P + ew[2], g_e2, e2);
  cw(e1_p, fte[1], gev[1], g_e2, ew[1], eb[1], 16);
  cb("ae_enc_conv_bwd", g_e2, ge[1], P + ew[1], g_e1, e1);
  {
    // weight gradient of the 1-channel first convolution (7x7, stride 2): rows = taps, each kernel row padded to 8
    // so that a quad of rows is 4 neighbouring pixels of the bordered image (16-byte loads at dword alignment);
    // the padding rows (kw = 7) are computed and never reduced
    std::vector<int32_t> ti(57), tr((size_t)B * 1024);
    for (int kh = 0; kh < 7; ++kh)
      for (int j = 0; j < 8; ++j) ti[kh * 8 + j] = kh * 69 + j;
    ti[56] = 0;
    for (int n = 0; n < B; ++n)
      for (int oh = 0; oh < 32; ++oh)
        for (int ow = 0; ow < 32; ++ow) tr[((size_t)n * 32 + oh) * 32 + ow] = n * 69 * 69 + 2 * oh * 69 + 2 * ow;
    IgemmProb p = blank();
    p.M = 57; p.N = 32; p.K = B * 1024;
    p.p_base[0] = x_p; p.p_tab_i = upload_vec(wk, ti); p.p_tab_r = upload_vec(wk, tr); single_part(p);
    p.p_ones_i = 56;
    p.vflags |= VF_P_TABS;
    p.q_base[0] = g_e1; p.q_ld_r[0] = 32; p.q_ld_j[0] = 1;
    p.ldc = 32;
    set_split(p, 64);
    p.c = wk.f32(p.slab_stride * p.split);
    wgc.push_back(p);
    for (int kh = 0; kh < 7; ++kh) {
      ReduceDesc r;
      memset(&r, 0, sizeof(r));
      r.src = p.c + (int64_t)kh * 8 * 32; r.splits = p.split; r.slab_stride = p.slab_stride;
      r.dst = grads + ew[0] + (int64_t)kh * 7 * 32; r.n = 7 * 32;
      reduces.push_back(r);
    }
    ReduceDesc rb;
    memset(&rb, 0, sizeof(rb));
    rb.src = p.c + (int64_t)56 * 32; rb.splits = p.split; rb.slab_stride = p.slab_stride; rb.dst = grads + eb[0]; rb.n = 32;
    reduces.push_back(rb);
  }
  {
    // uniform launches for the vectorised kernel; whatever it cannot take goes to igemm_kernel
    std::vector<IgemmProb> ok_c, rest;
    for (auto& p : wgc) (v2_prob_ok(p, 2) && (p.K % 4) == 0 ? ok_c : rest).push_back(p);
    add_launch(ops_ae, "ae_wgrad_conv", 2, ok_c);
    add_launch(ops_ae, "ae_wgrad_small", 2, rest);
    add_launch(ops_ae, "ae_wgrad_dense", 2, wgd);
  }
  {
    std::vector<int2> rt = reduce_tiles();
    d_reduces = upload_vec(wk, reduces);
    int2* d_rt = upload_vec(wk, rt);
    const int ntiles = (int)rt.size();
    ReduceDesc* dr = d_reduces;
    LossArgs none;
    memset(&none, 0, sizeof(none));
    elem("reduce_slabs", [=](hipStream_t s) {
      hipLaunchKernelGGL(reduce_slabs_kernel, dim3(ntiles), dim3(256), 0, s, dr, d_rt, ntiles, none, 0, AdamArgs{}, 0);
    });
  }
  {
    grl_ctx* self = this;
    elem("adam", [self](hipStream_t s) {
      AdamArgs aa;
      aa.params = self->params; aa.grads = self->grads; aa.m = self->adam_m; aa.v = self->adam_v;
      aa.n_train = self->n_train; aa.sc = self->sc; aa.grad_scale = 1.f; aa.tau = 0.f; aa.eps = 1e-7f;   // Keras epsilon
      aa.src_ofs = 0; aa.n_polyak = 0; aa.target = self->params;
      const int blocks = (int)std::min<int64_t>(2048, (self->n_train + 255) / 256);
      hipLaunchKernelGGL(adam_polyak_kernel, dim3(blocks), dim3(256), 0, s, aa);
    });
  }
  // =============================================================== encode path (batch NA) on the trained weights
  {
    ex_in = wk.f32((int64_t)NA * 4096);
    ec1 = wk.f32((int64_t)NA * 32 * 32 * 32); ec2 = wk.f32((int64_t)NA * 16 * 16 * 32);
    ec3 = wk.f32((int64_t)NA * 8 * 8 * 32); eout = wk.f32((int64_t)NA * 100);
    float* io[4] = {ex_in, ec1, ec2, ec3};
    for (int l = 0; l < 3; ++l) {
      ConvFwdTabs t = conv_fwd_tabs(ge[l], NA);
      add_launch(ops_enc, "enc_conv", 0, {conv_fwd(io[l], t, ge[l], P + ew[l], P + eb[l], io[l + 1], ACT_LEAKY, LA)});
    }
    IgemmProb p = dense_fwd(ec3, 2048, 2048, nullptr, 0, 0, NA, P + edw, 100, P + edb, eout, 100, ACT_LEAKY);
    p.act_alpha = LA;
    add_launch(ops_enc, "enc_dense", 0, {p});
    for (int k = 0; k < 8; ++k) enc_w[k] = nullptr;
    enc_loaded = true;
  }
  dbg["out"] = {out, (int64_t)B * 4096};
  dbg["z"] = {z, (int64_t)B * 100};
  dbg["e3"] = {e3, (int64_t)B * 2048};
  dbg["d5"] = {d5, (int64_t)B * 32 * 32 * 32};
  dbg["grads"] = {grads, n_train};
  dbg["adam_m"] = {adam_m, n_train};
  dbg["adam_v"] = {adam_v, n_train};
  return GRL_OK;
}

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

// ==================================================================================================
// C ABI
// ==================================================================================================
static int check_cfg(const grl_config* c) {
  if (!c) return fail(GRL_ERR_INVALID, "null config");
  if (c->algo == GRL_ALGO_AE) {
    if (c->batch_size < 1 || c->batch_size > 4096) return fail(GRL_ERR_INVALID, "batch_size out of range");
    if (c->replay_capacity < 1) return fail(GRL_ERR_INVALID, "replay_capacity must be >= 1");
    return GRL_OK;
  }
  if (c->extractor < 0 || c->extractor > 2) return fail(GRL_ERR_INVALID, "extractor must be 0..2");
  if (c->n_layers < 1 || c->n_layers > GRL_MAX_LAYERS) return fail(GRL_ERR_INVALID, "n_layers out of range");
  for (int l = 0; l < c->n_layers; ++l)
    if (c->layers[l] < 1 || c->layers[l] > 4096) return fail(GRL_ERR_INVALID, "layer width out of range");
  if (c->batch_size < 1 || c->batch_size > 65536) return fail(GRL_ERR_INVALID, "batch_size out of range");
  if (c->act_dim < 1 || c->act_dim > 64) return fail(GRL_ERR_INVALID, "act_dim out of range");
  if (c->replay_capacity < 1) return fail(GRL_ERR_INVALID, "replay_capacity must be >= 1");
  if (c->algo < 0 || c->algo > 3) return fail(GRL_ERR_INVALID, "algo must be 0..3");
  if (c->algo != GRL_ALGO_SAC) {
    if (c->extractor != GRL_EXTRACTOR_MLP) return fail(GRL_ERR_INVALID, "DQN/BDQ run on vector observations (MLP extractor)");
    if (c->q_branches < 1 || c->q_branches > 16 || c->q_branches != c->act_dim)
      return fail(GRL_ERR_INVALID, "q_branches must be 1..16 and equal act_dim");
    if (c->q_bins < 2 || c->q_bins > 1024) return fail(GRL_ERR_INVALID, "q_bins out of range");
    if (c->q_n_common < 0 || c->q_n_common > GRL_MAX_LAYERS || c->q_n_branch < 1 || c->q_n_branch > GRL_MAX_LAYERS ||
        c->q_n_value < 1 || c->q_n_value > GRL_MAX_LAYERS)
      return fail(GRL_ERR_INVALID, "tower depths out of range (branch and value towers need >= 1 hidden layer)");
    if (c->q_per && c->batch_size > 1024) return fail(GRL_ERR_INVALID, "prioritised replay supports batch_size <= 1024");
    if (c->q_per && c->replay_capacity > (int64_t)PER_BLK * PER_BLK)
      return fail(GRL_ERR_INVALID, "prioritised replay supports up to 1024 x 1024 transitions (two-level segment tree)");
  }
  if (c->replay_rgb_u8) {
    const int c_img = c->obs_channels - ((c->extractor == GRL_EXTRACTOR_AUGMENTED && c->n_direct > 0) ? 1 : 0);
    if (c->algo != GRL_ALGO_SAC || c->extractor == GRL_EXTRACTOR_MLP || c_img != 4)
      return fail(GRL_ERR_INVALID, "replay_rgb_u8 needs SAC on RGB-D images with 4 image channels (R, G, B, depth)");
  }
  if (c->extractor == GRL_EXTRACTOR_MLP) {
    if (c->obs_dim < 1) return fail(GRL_ERR_INVALID, "obs_dim must be >= 1 for the MLP extractor");
  } else {
    if (c->img_hw != 64) return fail(GRL_ERR_INVALID, "only 64x64 images (camera_info.yaml) are supported");
    if (c->obs_channels < 1 || c->obs_channels > 8) return fail(GRL_ERR_INVALID, "obs_channels out of range");
    if (c->extractor == GRL_EXTRACTOR_AUGMENTED && (c->n_direct < 0 || c->n_direct > 64))
      return fail(GRL_ERR_INVALID, "n_direct out of range");
    if (c->extractor == GRL_EXTRACTOR_AUGMENTED && c->n_direct > 0 && c->obs_channels < 2)
      return fail(GRL_ERR_INVALID, "augmented extractor needs >= 2 observation channels");
  }
  return GRL_OK;
}

extern "C" {

const char* grl_last_error(void) { return g_err.c_str(); }
int grl_version(void) { return 1; }

int grl_query_sizes(const grl_config* cfg, grl_sizes* out) {
  if (int e = check_cfg(cfg)) return e;
  if (!out) return fail(GRL_ERR_INVALID, "null out");
  grl_ctx ctx;
  ctx.cfg = *cfg;
  ctx.dry = true;
  if (int e = ctx.plan()) return e;
  out->state_bytes = ctx.st.off + 256;
  out->grads_bytes = ctx.gr.off + 256;
  out->work_bytes = ctx.wk.off + 256;
  out->replay_bytes = ctx.rp.off + 256;
  out->n_params = ctx.n_params;
  out->n_trainable = ctx.n_train;
  return GRL_OK;
}

int grl_create(const grl_config* cfg, const grl_buffers* bufs, grl_handle* out) {
  if (int e = check_cfg(cfg)) return e;
  if (!bufs || !out || !bufs->state || !bufs->grads || !bufs->work || !bufs->replay)
    return fail(GRL_ERR_INVALID, "null buffers");
  grl_ctx* h = new grl_ctx();
  h->cfg = *cfg;
  h->st.base = (char*)bufs->state; h->gr.base = (char*)bufs->grads;
  h->wk.base = (char*)bufs->work; h->rp.base = (char*)bufs->replay;
  if (int e = h->plan()) { delete h; return e; }
  const char* ng = getenv("GRL_NO_GRAPH");
  h->use_graph = !(ng && ng[0] == '1');
  for (auto& u : h->uploads) {
    hipError_t e = hipMemcpy(u.dst, u.bytes.data(), u.bytes.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) { delete h; return fail(GRL_ERR_HIP, std::string("table upload: ") + hipGetErrorString(e)); }
  }
  h->uploads.clear();
  h->htab.clear();
  for (auto& z : h->zero_once) hipMemset(z.first, 0, z.second);
  // state: zero Adam moments, scalars; stats = identity
  hipMemset(h->adam_m, 0, (size_t)h->n_train * 4);
  hipMemset(h->adam_v, 0, (size_t)h->n_train * 4);
  hipMemset(h->grads, 0, (size_t)h->n_train * 4);
  DevScalars s0;
  memset(&s0, 0, sizeof(s0));
  s0.beta1_power = 0.9f; s0.beta2_power = 0.999f;
  s0.lr = cfg->lr;
  hipError_t e = hipMemcpy(h->sc, &s0, sizeof(s0), hipMemcpyHostToDevice);
  if (e != hipSuccess) { delete h; return fail(GRL_ERR_HIP, std::string("scalar init: ") + hipGetErrorString(e)); }
  if (h->n_mean) {   // RunningMeanStd(): mean 0, var 1, count 1e-4
    std::vector<double> ones((size_t)h->n_elems, 1.0);
    const double c2[2] = {1e-4, 1e-4};
    hipMemset(h->n_mean, 0, (size_t)h->n_elems * 8);
    hipMemcpy(h->n_var, ones.data(), ones.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(h->n_count, c2, 16, hipMemcpyHostToDevice);
  }
  if (h->per_on) {
    hipMemset(h->per.p, 0, (size_t)cfg->replay_capacity * 8);
    PerState ps;
    memset(&ps, 0, sizeof(ps));
    ps.max_priority = 1.f; ps.p_min = 1.0; ps.beta = 1.0;
    e = hipMemcpy(h->per.st, &ps, sizeof(ps), hipMemcpyHostToDevice);
    if (e != hipSuccess) { delete h; return fail(GRL_ERR_HIP, std::string("per init: ") + hipGetErrorString(e)); }
  }
  *out = h;
  return GRL_OK;
}

int grl_destroy(grl_handle h) {
  if (!h) return GRL_OK;
  hipStreamSynchronize(h->stream);
  delete h;
  return GRL_OK;
}

int grl_set_stream(grl_handle h, void* s) {
  if (!h) return fail(GRL_ERR_INVALID, "null handle");
  if ((hipStream_t)s != h->stream) h->drop_graphs();
  h->stream = (hipStream_t)s;
  return GRL_OK;
}

int grl_param_count(grl_handle h) { return h ? (int)h->vars.size() : fail(GRL_ERR_INVALID, "null handle"); }

int grl_param_info(grl_handle h, int i, char* name, int cap, int64_t* off, int64_t* numel, int32_t* ndim,
                   int64_t shape[4], int32_t* trainable) {
  if (!h || i < 0 || i >= (int)h->vars.size()) return fail(GRL_ERR_INVALID, "bad parameter index");
  const Var& v = h->vars[i];
  if (name && cap > 0) { strncpy(name, v.name.c_str(), cap - 1); name[cap - 1] = 0; }
  if (off) *off = v.off;
  if (numel) *numel = v.numel;
  if (ndim) *ndim = v.ndim;
  if (shape) for (int k = 0; k < 4; ++k) shape[k] = v.shape[k];
  if (trainable) *trainable = v.trainable ? 1 : 0;
  return GRL_OK;
}

int grl_reset_optimizer(grl_handle h) {
  if (!h) return fail(GRL_ERR_INVALID, "null handle");
  HIPCHK(hipMemsetAsync(h->adam_m, 0, (size_t)h->n_train * 4, h->stream));
  HIPCHK(hipMemsetAsync(h->adam_v, 0, (size_t)h->n_train * 4, h->stream));
  DevScalars s0;
  HIPCHK(hipMemcpy(&s0, h->sc, sizeof(s0), hipMemcpyDeviceToHost));
  s0.beta1_power = 0.9f; s0.beta2_power = 0.999f;
  HIPCHK(hipMemcpy(h->sc, &s0, sizeof(s0), hipMemcpyHostToDevice));
  return GRL_OK;
}

int grl_set_learning_rate(grl_handle h, float lr) {
  if (!h) return fail(GRL_ERR_INVALID, "null handle");
  if (!(lr >= 0.f)) return fail(GRL_ERR_INVALID, "learning rate must be >= 0");
  // the step size lives in device memory (the captured graphs read it), set in stream order
  hipLaunchKernelGGL(set_f32_kernel, dim3(1), dim3(1), 0, h->stream, &h->sc->lr, lr);
  return GRL_OK;
}

int grl_set_obs_stats(grl_handle h, const double* mean, const double* var, double ret_var) {
  if (!h || !mean || !var) return fail(GRL_ERR_INVALID, "null argument");
  const grl_config& c = h->cfg;
  const double eps = c.norm_eps;
  const int nd = h->cnn ? h->F - 512 : 0;
  // The five statistic blocks sit one after the other in the state arena (alignment gaps in between are unused):
  // they are written into a page-locked mirror of that span and leave as ONE asynchronous copy in stream order.
  // Two mirrors alternate, each guarded by an event, so the host never waits for the GPU here (the learn loop calls
  // this before every update: a stream synchronisation plus five blocking copies serialised host and device).
  char* base = (char*)h->s_mean;
  const size_t span = h->n_mean ? (size_t)((char*)(h->n_var + h->n_elems) - base) : (size_t)((char*)h->s_ret + 8 - base);
  if (!h->pin_stats[0]) {
    for (int k = 0; k < 2; ++k) {
      HIPCHK(hipHostMalloc((void**)&h->pin_stats[k], span, 0));
      memset(h->pin_stats[k], 0, span);
      HIPCHK(hipEventCreateWithFlags(&h->pin_stats_ev[k], hipEventDisableTiming));
    }
  }
  const int k = h->pin_stats_next;
  h->pin_stats_next ^= 1;
  if (h->pin_stats_used[k]) HIPCHK(hipEventSynchronize(h->pin_stats_ev[k]));   // the copy issued two calls ago
  char* pm = h->pin_stats[k];
  double* m = (double*)pm;
  double* s = (double*)(pm + ((char*)h->s_std - base));
  double* dm = (double*)(pm + ((char*)h->s_dmean - base));
  double* ds = (double*)(pm + ((char*)h->s_dstd - base));
  double* rs = (double*)(pm + ((char*)h->s_ret - base));
  if (h->cnn) {
    const int co = c.obs_channels, ci = h->C_img;
    for (int px = 0; px < h->hw * h->hw; ++px)
      for (int ch = 0; ch < ci; ++ch) {
        m[px * ci + ch] = mean[px * co + ch];
        s[px * ci + ch] = std::sqrt(var[px * co + ch] + eps);
      }
    for (int q = 0; q < nd; ++q) {
      dm[q] = mean[q * co + (co - 1)];
      ds[q] = std::sqrt(var[q * co + (co - 1)] + eps);
    }
  } else {
    for (int q = 0; q < h->img_elems; ++q) { m[q] = mean[q]; s[q] = std::sqrt(var[q] + eps); }
  }
  *rs = std::sqrt(ret_var + eps);
  if (h->n_mean) {   // the running statistics grl_norm_update continues from (env layout)
    memcpy(pm + ((char*)h->n_mean - base), mean, (size_t)h->n_elems * 8);
    memcpy(pm + ((char*)h->n_var - base), var, (size_t)h->n_elems * 8);
  }
  HIPCHK(hipMemcpyAsync(base, pm, span, hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipEventRecord(h->pin_stats_ev[k], h->stream));
  h->pin_stats_used[k] = true;
  return GRL_OK;
}

int grl_set_ret_var(grl_handle h, double ret_var) {
  if (!h) return fail(GRL_ERR_INVALID, "null handle");
  const double sd = std::sqrt(ret_var + (double)h->cfg.norm_eps);
  HIPCHK(hipMemcpyAsync(h->s_ret, &sd, 8, hipMemcpyHostToDevice, h->stream));   // (pageable source: staged before returning)
  return GRL_OK;
}

int grl_set_obs_count(grl_handle h, double count) {
  if (!h || !h->n_count) return fail(GRL_ERR_STATE, "this handle keeps no running statistics");
  const double c2[2] = {count, count};
  HIPCHK(hipMemcpyAsync(h->n_count, c2, 16, hipMemcpyHostToDevice, h->stream));
  return GRL_OK;
}

int grl_norm_update(grl_handle h, const float* obs, int n) {
  if (!h || !obs || n < 1) return fail(GRL_ERR_INVALID, "bad argument");
  if (!h->n_mean) return fail(GRL_ERR_STATE, "this handle keeps no running statistics");
  if (n > h->stg_n) return fail(GRL_ERR_INVALID, "more observations than one env step of act_batch environments");
  const grl_config& c = h->cfg;
  HIPCHK(hipMemcpyAsync(h->n_stage, obs, (size_t)n * h->n_elems * 4, hipMemcpyHostToDevice, h->stream));
  NormUpdateArgs a;
  memset(&a, 0, sizeof(a));
  a.obs = h->n_stage; a.n = n; a.elems = (int)h->n_elems;
  a.mean = h->n_mean; a.var = h->n_var; a.count = h->n_count; a.parity = h->n_parity; a.eps = c.norm_eps;
  a.hw = h->hw * h->hw; a.c_obs = c.obs_channels; a.c_img = h->C_img; a.n_direct = h->cnn ? h->F - 512 : 0; a.vec = h->cnn ? 0 : 1;
  a.s_mean = h->s_mean; a.s_std = h->s_std; a.s_dmean = h->s_dmean; a.s_dstd = h->s_dstd;
  hipLaunchKernelGGL(norm_update_kernel, dim3((unsigned)((h->n_elems + 255) / 256)), dim3(256), 0, h->stream, a);
  h->n_parity ^= 1;
  HIPCHK(hipGetLastError());
  return GRL_OK;
}

int grl_get_obs_stats(grl_handle h, double* mean, double* var, double* count) {
  if (!h || !mean || !var || !count) return fail(GRL_ERR_INVALID, "null argument");
  if (!h->n_mean) return fail(GRL_ERR_STATE, "this handle keeps no running statistics");
  HIPCHK(hipStreamSynchronize(h->stream));
  HIPCHK(hipMemcpy(mean, h->n_mean, (size_t)h->n_elems * 8, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(var, h->n_var, (size_t)h->n_elems * 8, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(count, h->n_count + h->n_parity, 8, hipMemcpyDeviceToHost));
  return GRL_OK;
}

static int replay_add_dev(grl_handle h, const float* obs, const float* act, const float* rew, const float* nxt,
                          const float* done, int n) {
  const grl_config& c = h->cfg;
  IngestArgs ia;
  memset(&ia, 0, sizeof(ia));
  ia.obs = obs; ia.next_obs = nxt; ia.act = act; ia.rew = rew; ia.done = done;
  ia.n = n; ia.hw = h->hw * h->hw; ia.c_obs = c.obs_channels; ia.c_img = h->C_img;
  ia.n_direct = h->cnn ? h->F - 512 : 0; ia.act_dim = h->A; ia.vec_dim = h->cnn ? 0 : c.obs_dim;
  ia.pos = h->rp_pos; ia.cap = c.replay_capacity;
  ia.rp_obs = h->rp_obs; ia.rp_next = h->rp_next; ia.rp_dobs = h->rp_dobs; ia.rp_dnext = h->rp_dnext;
  ia.rp_act = h->rp_act; ia.rp_rew = h->rp_rew; ia.rp_done = h->rp_done;
  ia.rgb_u8 = (c.algo == GRL_ALGO_SAC) ? c.replay_rgb_u8 : 0;
  const int elems = h->cnn ? (ia.rgb_u8 ? h->hw * h->hw : h->img_elems) : c.obs_dim;
  hipLaunchKernelGGL(ingest_kernel, dim3((elems + 255) / 256, n, 2), dim3(256), 0, h->stream, ia);
  if (h->per_on)   // new transitions enter with max_priority ** alpha
    hipLaunchKernelGGL(per_add_kernel, dim3((n + 255) / 256), dim3(256), 0, h->stream, h->per, h->rp_pos, n,
                       (int64_t)c.replay_capacity);
  h->rp_pos = (h->rp_pos + n) % c.replay_capacity;
  h->rp_size = std::min<int64_t>(c.replay_capacity, h->rp_size + n);
  HIPCHK(hipMemcpyAsync(&h->sc->replay_size, &h->rp_size, 8, hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipGetLastError());
  return GRL_OK;
}

int grl_replay_add_device(grl_handle h, const float* obs, const float* act, const float* rew,
                          const float* next_obs, const float* done, int n) {
  if (!h || !obs || !act || !rew || !next_obs || !done || n < 1) return fail(GRL_ERR_INVALID, "bad argument");
  if (n > h->cfg.replay_capacity) return fail(GRL_ERR_INVALID, "n exceeds replay capacity");
  return replay_add_dev(h, obs, act, rew, next_obs, done, n);
}

// host buffer -> pinned staging -> device (and back): hipMemcpyAsync from pageable memory is a synchronous
// staged copy; through page-locked buffers the per-call overhead is a host memcpy plus a true async DMA
static int pin_reserve(grl_handle h, size_t in_floats, size_t out_floats) {
  if (in_floats > h->pin_in_n) {
    if (h->pin_in) hipHostFree(h->pin_in);
    h->pin_in = nullptr; h->pin_in_n = 0;
    HIPCHK(hipHostMalloc((void**)&h->pin_in, in_floats * 4, 0));
    h->pin_in_n = in_floats;
  }
  if (out_floats > h->pin_out_n) {
    if (h->pin_out) hipHostFree(h->pin_out);
    h->pin_out = nullptr; h->pin_out_n = 0;
    HIPCHK(hipHostMalloc((void**)&h->pin_out, out_floats * 4, 0));
    h->pin_out_n = out_floats;
  }
  return GRL_OK;
}

int grl_replay_add(grl_handle h, const float* obs, const float* act, const float* rew, const float* next_obs,
                   const float* done, int n) {
  if (!h || !obs || !act || !rew || !next_obs || !done || n < 1) return fail(GRL_ERR_INVALID, "bad argument");
  const int64_t oe = h->cnn ? (int64_t)h->hw * h->hw * h->cfg.obs_channels : h->cfg.obs_dim;
  // (pageable copies: for these sizes -- 64 KB per transition -- an extra host copy into pinned staging costs
  // more than it saves: 77 -> 90 us for 16 transitions, 150 -> 214 us for 64; measured with scripts/act_bench.py)
  for (int k0 = 0; k0 < n; k0 += h->stg_n) {
    const int m = std::min(h->stg_n, n - k0);
    HIPCHK(hipMemcpyAsync(h->stg_obs, obs + k0 * oe, (size_t)m * oe * 4, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->stg_next, next_obs + k0 * oe, (size_t)m * oe * 4, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->stg_act, act + (int64_t)k0 * h->A, (size_t)m * h->A * 4, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->stg_rew, rew + k0, (size_t)m * 4, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->stg_done, done + k0, (size_t)m * 4, hipMemcpyHostToDevice, h->stream));
    if (int e = replay_add_dev(h, h->stg_obs, h->stg_act, h->stg_rew, h->stg_next, h->stg_done, m)) return e;
    HIPCHK(hipStreamSynchronize(h->stream));   // staging buffers are reused by the next chunk
  }
  return GRL_OK;
}

int64_t grl_replay_size(grl_handle h) { return h ? h->rp_size : -1; }

static int stage_noise(grl_handle h, const int64_t* idx, const float* eps, int step) {
  HIPCHK(hipMemcpyAsync(h->idx_buf, idx + (int64_t)step * h->B, (size_t)h->B * 8, hipMemcpyDeviceToDevice, h->stream));
  const int64_t per = h->cfg.algo == GRL_ALGO_SAC ? (int64_t)h->B * h->A : (int64_t)h->B;   // Q: importance weights
  HIPCHK(hipMemcpyAsync(h->eps_buf, eps + (int64_t)step * per, (size_t)per * 4, hipMemcpyDeviceToDevice, h->stream));
  return GRL_OK;
}

int grl_compute_grads(grl_handle h, const int64_t* idx, const float* eps) {
  if (!h) return fail(GRL_ERR_INVALID, "null handle");
  if ((idx == nullptr) != (eps == nullptr)) return fail(GRL_ERR_INVALID, "idx and eps must both be given or both be NULL");
  if (h->rp_size < 1) return fail(GRL_ERR_STATE, "replay buffer is empty");
  if (idx) {
    if (int e = stage_noise(h, idx, eps, 0)) return e;
    if (int e = h->run_seq("grads_explicit", {&h->ops_gather, &h->ops_grads})) return e;
  } else {
    if (int e = h->run_seq("grads_rng", {&h->ops_rng, &h->ops_grads})) return e;
  }
  HIPCHK(hipGetLastError());
  return GRL_OK;
}

int grl_compute_grads_staged(grl_handle h, int stage, const int64_t* idx, const float* eps) {
  if (!h) return fail(GRL_ERR_INVALID, "null handle");
  if (stage != 0 && stage != 1) return fail(GRL_ERR_INVALID, "stage must be 0 or 1");
  if (!h->staged_ok) {          // plans without a staged form: everything in stage 0, one bucket (grl_grad_ranges)
    if (stage == 1) return GRL_OK;
    return grl_compute_grads(h, idx, eps);
  }
  if (stage == 1) {
    if (int e = h->run_seq("grads_stage1", {&h->ops_stage1})) return e;
    HIPCHK(hipGetLastError());
    return GRL_OK;
  }
  if ((idx == nullptr) != (eps == nullptr)) return fail(GRL_ERR_INVALID, "idx and eps must both be given or both be NULL");
  if (h->rp_size < 1) return fail(GRL_ERR_STATE, "replay buffer is empty");
  if (idx) {
    if (int e = stage_noise(h, idx, eps, 0)) return e;
    if (int e = h->run_seq("grads_stage0_explicit", {&h->ops_gather, &h->ops_stage0})) return e;
  } else {
    if (int e = h->run_seq("grads_stage0_rng", {&h->ops_rng, &h->ops_stage0})) return e;
  }
  HIPCHK(hipGetLastError());
  return GRL_OK;
}

int grl_grad_ranges(grl_handle h, int bucket, int cap, int64_t* offsets, int64_t* numels) {
  if (!h || !offsets || !numels) return fail(GRL_ERR_INVALID, "null argument");
  if (bucket != 0 && bucket != 1) return fail(GRL_ERR_INVALID, "bucket must be 0 or 1");
  std::vector<std::pair<int64_t, int64_t>> r;
  if (!h->staged_ok) {
    if (bucket == 0) r.push_back({0, h->n_train});
  } else if (bucket == 0) {      // dense: fc + heads of the policy net, fc + vf / qf1 / qf2 heads of the value net
    r.push_back({h->ex[0].fw, h->vf_off - h->ex[0].fw});
    r.push_back({h->ex[1].fw, h->ent_off - h->ex[1].fw});
  } else {                       // convolutions of both nets + the entropy coefficient (its gradient comes with the loss sums)
    r.push_back({h->ex[0].w[0], h->ex[0].fw - h->ex[0].w[0]});
    r.push_back({h->ex[1].w[0], h->ex[1].fw - h->ex[1].w[0]});
    r.push_back({h->ent_off, h->n_train - h->ent_off});
  }
  if ((int)r.size() > cap) return fail(GRL_ERR_INVALID, "range buffer too small");
  for (size_t k = 0; k < r.size(); ++k) { offsets[k] = r[k].first; numels[k] = r[k].second; }
  return (int)r.size();
}

int grl_apply_grads(grl_handle h, float grad_scale) {
  if (!h) return fail(GRL_ERR_INVALID, "null handle");
  if (grad_scale != h->apply_graph_scale) {   // the scale is baked into the captured kernel arguments
    auto it = h->graphs.find("apply");
    if (it != h->graphs.end()) { (void)hipGraphExecDestroy(it->second); h->graphs.erase(it); }
    h->apply_graph_scale = grad_scale;
  }
  h->grad_scale = grad_scale;
  if (int e = h->run_seq("apply", {&h->ops_apply})) return e;
  HIPCHK(hipGetLastError());
  return GRL_OK;
}

int grl_train_step(grl_handle h, int n_steps, const int64_t* idx, const float* eps) {
  if (!h || n_steps < 1) return fail(GRL_ERR_INVALID, "bad argument");
  if (h->cfg.algo == GRL_ALGO_AE) return fail(GRL_ERR_STATE, "auto-encoder handles train with grl_ae_train_step");
  if ((idx == nullptr) != (eps == nullptr)) return fail(GRL_ERR_INVALID, "idx and eps must both be given or both be NULL");
  if (h->rp_size < 1) return fail(GRL_ERR_STATE, "replay buffer is empty");
  h->grad_scale = 1.f;
  if (!idx && n_steps >= 2 && h->prefetch_ok && !h->prof) {   // device RNG, several updates: prefetching sequences (plan_sac)
    for (int s = 0; s < n_steps; ++s) {
      const bool first = s == 0, last = s == n_steps - 1;
      if (int e = h->run_seq(first ? "pf_first" : (last ? "pf_last" : "pf_mid"),
                             {first ? &h->ops_pf_first : (last ? &h->ops_pf_last : &h->ops_pf_mid)}))
        return e;
    }
    HIPCHK(hipGetLastError());
    return GRL_OK;
  }
  for (int s = 0; s < n_steps; ++s) {
    if (idx) {
      if (int e = stage_noise(h, idx, eps, s)) return e;
      if (!h->ops_grads_apply.empty()) {
        if (int e = h->run_seq("full_explicit", {&h->ops_gather, &h->ops_grads_apply})) return e;
      } else if (int e = h->run_seq("full_explicit", {&h->ops_gather, &h->ops_grads, &h->ops_apply})) return e;
    } else {
      if (!h->ops_grads_apply.empty()) {
        if (int e = h->run_seq("full_rng", {&h->ops_rng, &h->ops_grads_apply})) return e;
      } else if (int e = h->run_seq("full_rng", {&h->ops_rng, &h->ops_grads, &h->ops_apply})) return e;
    }
  }
  HIPCHK(hipGetLastError());
  return GRL_OK;
}

int grl_train_step_per(grl_handle h, int n_steps, double beta, const double* u) {
  if (!h || n_steps < 1) return fail(GRL_ERR_INVALID, "bad argument");
  if (!h->per_on) return fail(GRL_ERR_STATE, "prioritised replay is not enabled (grl_config.q_per)");
  if (h->rp_size < 1) return fail(GRL_ERR_STATE, "replay buffer is empty");
  h->grad_scale = 1.f;
  if (!(beta > 0.0)) return fail(GRL_ERR_INVALID, "beta must be positive (PrioritizedReplayBuffer.sample asserts beta > 0)");
  if (h->rp_size < 2) return fail(GRL_ERR_STATE, "prioritised sampling needs at least two stored transitions (sum(0, len - 1))");
  HIPCHK(hipMemcpyAsync(&h->per.st->beta, &beta, 8, hipMemcpyHostToDevice, h->stream));
  for (int s = 0; s < n_steps; ++s) {
    if (u) {
      HIPCHK(hipMemcpyAsync(h->per_u, u + (int64_t)s * h->B, (size_t)h->B * 8, hipMemcpyDeviceToDevice, h->stream));
      if (!h->ops_grads_apply_per.empty()) {
        if (int e = h->run_seq("per_u", {&h->ops_per_u_g, &h->ops_grads_apply_per})) return e;
      } else if (!h->ops_grads_apply.empty()) {
        if (int e = h->run_seq("per_u", {&h->ops_per_u, &h->ops_gather, &h->ops_grads_apply, &h->ops_per_update})) return e;
      } else if (int e = h->run_seq("per_u", {&h->ops_per_u, &h->ops_gather, &h->ops_grads, &h->ops_apply, &h->ops_per_update})) return e;
    } else {
      if (!h->ops_grads_apply_per.empty()) {
        if (int e = h->run_seq("per_rng", {&h->ops_per_rng_g, &h->ops_grads_apply_per})) return e;
      } else if (!h->ops_grads_apply.empty()) {
        if (int e = h->run_seq("per_rng", {&h->ops_per_rng, &h->ops_gather, &h->ops_grads_apply, &h->ops_per_update})) return e;
      } else if (int e = h->run_seq("per_rng", {&h->ops_per_rng, &h->ops_gather, &h->ops_grads, &h->ops_apply, &h->ops_per_update})) return e;
    }
  }
  HIPCHK(hipGetLastError());
  return GRL_OK;
}

int grl_ae_train_step(grl_handle h, const float* imgs, int n_steps) {
  if (!h || !imgs || n_steps < 1) return fail(GRL_ERR_INVALID, "bad argument");
  if (h->cfg.algo != GRL_ALGO_AE) return fail(GRL_ERR_STATE, "not an auto-encoder handle (grl_config.algo)");
  const size_t per = (size_t)h->B * 4096;
  for (int s = 0; s < n_steps; ++s) {
    HIPCHK(hipMemcpyAsync(h->ae_x, imgs + per * s, per * 4, hipMemcpyDeviceToDevice, h->stream));
    if (int e = h->run_seq("ae_step", {&h->ops_ae})) return e;
  }
  HIPCHK(hipGetLastError());
  return GRL_OK;
}

int grl_ae_reconstruct(grl_handle h, const float* imgs, float* out) {
  if (!h || !imgs || !out) return fail(GRL_ERR_INVALID, "null argument");
  if (h->cfg.algo != GRL_ALGO_AE) return fail(GRL_ERR_STATE, "not an auto-encoder handle (grl_config.algo)");
  const size_t per = (size_t)h->B * 4096;
  HIPCHK(hipMemcpyAsync(h->ae_x, imgs, per * 4, hipMemcpyDeviceToDevice, h->stream));
  if (int e = h->run_seq("ae_fwd", {&h->ops_ae_fwd})) return e;
  HIPCHK(hipMemcpyAsync(out, h->ae_out, per * 4, hipMemcpyDeviceToDevice, h->stream));
  HIPCHK(hipGetLastError());
  return GRL_OK;
}

int grl_get_metrics(grl_handle h, grl_metrics* out) {
  if (!h || !out) return fail(GRL_ERR_INVALID, "null argument");
  DevScalars s;
  HIPCHK(hipStreamSynchronize(h->stream));
  HIPCHK(hipMemcpy(&s, h->sc, sizeof(s), hipMemcpyDeviceToHost));
  out->policy_loss = s.policy_loss; out->qf1_loss = s.qf1_loss; out->qf2_loss = s.qf2_loss;
  out->value_loss = s.value_loss; out->ent_coef_loss = s.ent_loss; out->ent_coef = s.ent_coef;
  out->entropy = s.entropy; out->mean_qf1 = s.mean_qf1; out->mean_v = s.mean_v;
  return GRL_OK;
}

int grl_act(grl_handle h, const float* obs, int n, int flags, const float* eps, float* out) {
  if (!h || !obs || !out || n < 1) return fail(GRL_ERR_INVALID, "bad argument");
  const int deterministic = flags & 1;
  const bool raw = (flags & 2) != 0;      // raw observations: VecNormalize applied on the device (grl_norm_update statistics)
  if (raw && h->ops_act_norm.empty()) return fail(GRL_ERR_STATE, "this handle has no normalising act path");
  if (n > h->NA) return fail(GRL_ERR_INVALID, "n exceeds act_batch");
  const bool q = h->cfg.algo != GRL_ALGO_SAC;   // DQN / BDQ: Q-values [n, D*bins]
  if (!q && !deterministic && !eps) return fail(GRL_ERR_INVALID, "stochastic action needs eps");
  const int64_t oe = (!q && h->cnn) ? (int64_t)h->hw * h->hw * h->cfg.obs_channels : h->cfg.obs_dim;
  const size_t n_in = (size_t)n * oe, n_eps = (!q && !deterministic) ? (size_t)n * h->A : 0;
  const size_t n_out = q ? (size_t)n * h->qD * h->qN : (size_t)n * h->A;
  if (int e = pin_reserve(h, n_in + n_eps, n_out)) return e;
  memcpy(h->pin_in, obs, n_in * 4);
  if (n_eps) memcpy(h->pin_in + n_in, eps, n_eps * 4);
  HIPCHK(hipMemcpyAsync(h->stg_obs, h->pin_in, n_in * 4, hipMemcpyHostToDevice, h->stream));
  if (n_eps) HIPCHK(hipMemcpyAsync(h->a_eps, h->pin_in + n_in, n_eps * 4, hipMemcpyHostToDevice, h->stream));
  if (q) {
    if (int e = h->run_seq("act", {&h->ops_act})) return e;
  } else {
    // the launches cover act_batch rows whatever n is (rows beyond n hold stale observations: computed, not returned)
    if (int e = h->run_seq(std::string(deterministic ? "act_det" : "act_sto") + (raw ? "_n" : ""),
                           {raw ? &h->ops_act_norm : &h->ops_act, deterministic ? &h->ops_act_det : &h->ops_act_sto}))
      return e;
  }
  HIPCHK(hipMemcpyAsync(h->pin_out, q ? h->q_aout : h->a_out, n_out * 4, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  HIPCHK(hipGetLastError());
  memcpy(out, h->pin_out, n_out * 4);
  return GRL_OK;
}

// ---------------------------------------------------------------------------------------------- data parallel, in-graph
static size_t dp_layout(int64_t n, size_t* src_off, size_t* res_off) {
  const size_t ctl = (size_t)rup((int64_t)sizeof(DpCtl), 256);
  const size_t arr = (size_t)rup(n * 4, 256);
  *src_off = ctl;
  *res_off = ctl + arr;
  return ctl + 2 * arr;
}

int grl_allreduce_init(grl_handle h, int rank, int world, void* handle_out) {
  if (!h || !handle_out) return fail(GRL_ERR_INVALID, "null argument");
  if (h->cfg.algo != GRL_ALGO_SAC) return fail(GRL_ERR_STATE, "the exchange step is defined for SAC handles");
  if (world < 1 || world > DP_MAX_WORLD || rank < 0 || rank >= world) return fail(GRL_ERR_INVALID, "bad rank / world size");
  if (h->dp_buf) return fail(GRL_ERR_STATE, "grl_allreduce_init was already called on this handle");
  size_t so, ro;
  const size_t bytes = dp_layout(h->n_train, &so, &ro);
  // fine-grained: flag and data stores of a peer become visible to a kernel that is already running
  hipError_t e = hipExtMallocWithFlags(&h->dp_buf, bytes, hipDeviceMallocFinegrained);
  if (e != hipSuccess) { h->dp_buf = nullptr; return fail(GRL_ERR_HIP, std::string("exchange buffer: ") + hipGetErrorString(e)); }
  HIPCHK(hipMemset(h->dp_buf, 0, bytes));
  hipIpcMemHandle_t mh;
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "grl.h documents 64-byte handles");
  e = hipIpcGetMemHandle(&mh, h->dp_buf);
  if (e != hipSuccess) {
    (void)hipFree(h->dp_buf); h->dp_buf = nullptr;
    return fail(GRL_ERR_HIP, std::string("hipIpcGetMemHandle: ") + hipGetErrorString(e));
  }
  memcpy(handle_out, &mh, 64);
  memset(&h->dp, 0, sizeof(h->dp));
  h->dp.rank = rank; h->dp.world = world; h->dp.n = h->n_train;
  h->dp.chunk = rup((h->n_train + world - 1) / world, 4);
  h->dp.grads = h->grads;
  return GRL_OK;
}

int grl_allreduce_connect(grl_handle h, const void* handles) {
  if (!h || !handles) return fail(GRL_ERR_INVALID, "null argument");
  if (!h->dp_buf) return fail(GRL_ERR_STATE, "call grl_allreduce_init first");
  if (h->dp_on) return fail(GRL_ERR_STATE, "already connected");
  size_t so, ro;
  dp_layout(h->n_train, &so, &ro);
  DpArgs& d = h->dp;
  for (int p = 0; p < d.world; ++p) {
    char* base = (char*)h->dp_buf;
    if (p != d.rank) {
      hipIpcMemHandle_t mh;
      memcpy(&mh, (const char*)handles + 64 * (size_t)p, 64);
      void* ptr = nullptr;
      hipError_t e = hipIpcOpenMemHandle(&ptr, mh, hipIpcMemLazyEnablePeerAccess);
      if (e != hipSuccess) return fail(GRL_ERR_HIP, "hipIpcOpenMemHandle (rank " + std::to_string(p) + "): " + hipGetErrorString(e));
      h->dp_peer[p] = ptr;
      base = (char*)ptr;
    }
    d.ctl[p] = (DpCtl*)base;
    d.src[p] = (float*)(base + so);
    d.res[p] = (float*)(base + ro);
  }
  const DpArgs da = d;
  grl_ctx* self = h;
  {
    Op op; op.tag = "dp_publish"; op.bytes = 8.0 * (double)d.n;
    const int blocks = (int)std::min<int64_t>(512, std::max<int64_t>(1, (d.n / 4 + 255) / 256));
    op.run = [da, blocks](hipStream_t s) { hipLaunchKernelGGL(dp_publish_kernel, dim3(blocks), dim3(256), 0, s, da); };
    h->ops_dp.push_back(op);
  }
  {
    Op op; op.tag = "dp_reduce_push"; op.bytes = 8.0 * (double)d.chunk * d.world;
    const int blocks = (int)std::min<int64_t>(256, std::max<int64_t>(1, (d.chunk / 4 + 255) / 256));
    op.run = [da, blocks](hipStream_t s) { hipLaunchKernelGGL(dp_reduce_push_kernel, dim3(blocks), dim3(256), 0, s, da); };
    h->ops_dp.push_back(op);
  }
  {
    Op op; op.tag = "dp_apply"; op.bytes = (double)h->n_train * 4 * 7 + (double)h->n_polyak * 4 * 2;
    op.run = [self, da](hipStream_t s) {
      AdamArgs aa;
      memset(&aa, 0, sizeof(aa));
      aa.params = self->params; aa.grads = da.res[da.rank]; aa.m = self->adam_m; aa.v = self->adam_v;
      aa.n_train = self->n_train; aa.sc = self->sc; aa.grad_scale = 1.f / (float)da.world; aa.tau = self->cfg.tau; aa.eps = 1e-8f;
      aa.src_ofs = self->vf_off; aa.n_polyak = self->n_polyak; aa.target = self->params + self->tgt_off;
      const int blocks = (int)std::min<int64_t>(1024, (self->n_train + 255) / 256);
      hipLaunchKernelGGL(dp_apply_kernel, dim3(blocks), dim3(256), 0, s, aa, da);
    };
    h->ops_dp.push_back(op);
  }
  h->dp_on = true;
  return GRL_OK;
}

int grl_train_step_allreduce(grl_handle h, int n_steps, const int64_t* idx, const float* eps) {
  if (!h || n_steps < 1) return fail(GRL_ERR_INVALID, "bad argument");
  if (!h->dp_on) return fail(GRL_ERR_STATE, "call grl_allreduce_init / grl_allreduce_connect first");
  if ((idx == nullptr) != (eps == nullptr)) return fail(GRL_ERR_INVALID, "idx and eps must both be given or both be NULL");
  if (h->rp_size < 1) return fail(GRL_ERR_STATE, "replay buffer is empty");
  for (int s = 0; s < n_steps; ++s) {
    if (idx) {
      if (int e = stage_noise(h, idx, eps, s)) return e;
      if (int e = h->run_seq("dp_explicit", {&h->ops_gather, &h->ops_grads, &h->ops_dp})) return e;
    } else if (int e = h->run_seq("dp_rng", {&h->ops_rng, &h->ops_grads, &h->ops_dp})) return e;
  }
  HIPCHK(hipGetLastError());
  return GRL_OK;
}

int grl_allreduce_status(grl_handle h, int64_t* exchanges, int* error) {
  if (!h || !h->dp_buf) return fail(GRL_ERR_STATE, "no exchange buffer");
  HIPCHK(hipStreamSynchronize(h->stream));
  DpCtl c;
  HIPCHK(hipMemcpy(&c, h->dp_buf, sizeof(c), hipMemcpyDeviceToHost));
  if (exchanges) *exchanges = c.epoch;
  if (error) *error = (int)c.error;
  if (c.error) return fail(GRL_ERR_STATE, "an exchange timed out waiting for a peer (the replicas are no longer in step)");
  return GRL_OK;
}

int grl_q_update_target(grl_handle h) {
  if (!h) return fail(GRL_ERR_INVALID, "null handle");
  if (h->cfg.algo == GRL_ALGO_SAC) return fail(GRL_ERR_STATE, "SAC has no hard target update");
  HIPCHK(hipMemcpyAsync(h->params + h->tgt_off, h->params + h->q_online_off, (size_t)h->q_online_n * 4,
                        hipMemcpyDeviceToDevice, h->stream));
  return GRL_OK;
}

int grl_encoder_load(grl_handle h, const float* const* w, const int64_t* numels, int n_arrays) {
  if (!h || !w || !numels || n_arrays != 8) return fail(GRL_ERR_INVALID, "expected 8 weight arrays");
  if (h->cfg.algo != GRL_ALGO_SAC) return fail(GRL_ERR_STATE, "grl_encoder_load is for SAC handles (auto-encoder handles encode with their own parameters)");
  const int64_t wn[8] = {7 * 7 * 32, 32, 5 * 5 * 32 * 32, 32, 3 * 3 * 32 * 32, 32, 2048 * 100, 100};
  for (int k = 0; k < 8; ++k)
    if (numels[k] != wn[k]) return fail(GRL_ERR_INVALID, "encoder weight " + std::to_string(k) + " has the wrong size");
  HIPCHK(hipStreamSynchronize(h->stream));
  for (int k = 0; k < 8; ++k) HIPCHK(hipMemcpy(h->enc_w[k], w[k], (size_t)wn[k] * 4, hipMemcpyHostToDevice));
  h->enc_loaded = true;
  return GRL_OK;
}

int grl_encode(grl_handle h, const float* depth, int n, float* out) {
  if (!h || !depth || !out || n < 1) return fail(GRL_ERR_INVALID, "bad argument");
  if (!h->enc_loaded) return fail(GRL_ERR_STATE, "grl_encoder_load has not been called");
  if (n > h->NA) return fail(GRL_ERR_INVALID, "n exceeds act_batch");
  if (int e = pin_reserve(h, (size_t)n * 4096, (size_t)n * 100)) return e;
  memcpy(h->pin_in, depth, (size_t)n * 4096 * 4);
  HIPCHK(hipMemcpyAsync(h->ex_in, h->pin_in, (size_t)n * 4096 * 4, hipMemcpyHostToDevice, h->stream));
  if (int e = h->run_seq("encode", {&h->ops_enc})) return e;
  HIPCHK(hipMemcpyAsync(h->pin_out, h->eout, (size_t)n * 100 * 4, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  HIPCHK(hipGetLastError());
  memcpy(out, h->pin_out, (size_t)n * 100 * 4);
  return GRL_OK;
}

int64_t grl_debug_fetch(grl_handle h, const char* name, float* out, int64_t cap) {
  if (!h || !name || !out) return fail(GRL_ERR_INVALID, "null argument");
  auto it = h->dbg.find(name);
  if (it == h->dbg.end()) return fail(GRL_ERR_INVALID, std::string("unknown tensor ") + name);
  const int64_t n = std::min(cap, it->second.second);
  if (hipStreamSynchronize(h->stream) != hipSuccess) return fail(GRL_ERR_HIP, "sync failed");
  hipError_t e = hipMemcpy(out, it->second.first, (size_t)n * 4, hipMemcpyDeviceToHost);
  if (e != hipSuccess) return fail(GRL_ERR_HIP, hipGetErrorString(e));
  return n;
}

int64_t grl_debug_store(grl_handle h, const char* name, const float* in, int64_t n) {
  if (!h || !name || !in) return fail(GRL_ERR_INVALID, "null argument");
  auto it = h->dbg.find(name);
  if (it == h->dbg.end()) return fail(GRL_ERR_INVALID, std::string("unknown tensor ") + name);
  if (n > it->second.second) return fail(GRL_ERR_INVALID, std::string("too many values for ") + name);
  if (hipStreamSynchronize(h->stream) != hipSuccess) return fail(GRL_ERR_HIP, "sync failed");
  hipError_t e = hipMemcpy(const_cast<float*>(it->second.first), in, (size_t)n * 4, hipMemcpyHostToDevice);
  if (e != hipSuccess) return fail(GRL_ERR_HIP, hipGetErrorString(e));
  return n;
}

int grl_profile_enable(grl_handle h, int on) {
  if (!h) return fail(GRL_ERR_INVALID, "null handle");
  h->prof = on != 0;
  if (on) h->prof_acc.clear();
  return GRL_OK;
}

int grl_profile_query(grl_handle h, const char* name, double* avg_ms, int64_t* launches) {
  if (!h || !name) return fail(GRL_ERR_INVALID, "null argument");
  auto it = h->prof_acc.find(name);
  if (it == h->prof_acc.end()) return fail(GRL_ERR_INVALID, std::string("no profile for ") + name);
  if (avg_ms) *avg_ms = it->second.n ? it->second.ms / it->second.n : 0.0;
  if (launches) *launches = it->second.n;
  return GRL_OK;
}

/* list profiled tags: writes "tag:avg_ms:launches:flops_per_launch:bytes_per_launch:executed_flops_per_launch\n" lines */
int grl_profile_dump(grl_handle h, char* buf, int cap) {
  if (!h || !buf || cap < 1) return fail(GRL_ERR_INVALID, "bad argument");
  std::string s;
  for (auto& kv : h->prof_acc) {
    char line[256];
    snprintf(line, sizeof(line), "%s:%.6f:%lld:%.0f:%.0f:%.0f\n", kv.first.c_str(),
             kv.second.n ? kv.second.ms / kv.second.n : 0.0, (long long)kv.second.n,
             kv.second.n ? kv.second.flops / kv.second.n : 0.0, kv.second.n ? kv.second.bytes / kv.second.n : 0.0,
             kv.second.n ? kv.second.flops_exec / kv.second.n : 0.0);
    s += line;
  }
  strncpy(buf, s.c_str(), cap - 1);
  buf[cap - 1] = 0;
  return GRL_OK;
}

}  // extern "C"
