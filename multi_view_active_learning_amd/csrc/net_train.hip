// Training-step executor: forward in train mode (batch-statistics BatchNorm) and the full
// backward pass of the heat-map network, as ONE C call each (reference: strategy.py:460-487;
// forward pose_estimators/hrnet.py:468-501 in train mode, backward = autograd).
//
// Per operator  out = act(((bn(conv(x, w)) up) + res1) + res2):
//   forward   conv on the matrix cores (raw z) -> batch statistics -> fused normalise + residual
//             + ReLU (+ nearest upsample) stream;
//   backward  mask / residual-gradient scatter / window-sum + dgamma, dbeta (two-stage float64
//             reduction) -> dz in place -> weight gradient (MFMA split-K, conv_wgrad.hip) ->
//             data gradient = the SAME forward MFMA kernel on tap-flipped, channel-swapped
//             weights (stride-2 convs read dz as a zero-dilated input), accumulated in place
//             into the producer's gradient buffer.
// Activation gradients accumulate in an arena laid out like the activations; the plan marks the
// first writer of every slot (mval_train_op.first_touch: bit 0 data gradient, bit 1 / 2 residual
// gradients), which stores instead of accumulating, so the arena is never zero-filled and fan-out
// (residual skips, HRNet fuse layers, transitions) needs no special casing.
#include "conv_p2.h"

extern "C" int mval_bn_batch_stats(const float*, int64_t, int, float, float, float*, float*, float*, float*, double*,
                                   void*);
extern "C" int mval_bn_apply_fwd(const float*, const float*, const float*, const float*, const float*, const float*,
                                 const float*, float*, int, int, int, int, int, int, void*);
extern "C" int mval_bn_apply_fwd_mask(const float*, const float*, const float*, const float*, const float*, const float*, const float*, float*,
                                      int, int, int, int, int, int, uint32_t*, uint8_t*, void*);
extern "C" int mval_bn_bwd_fused_mask(const float*, const float*, const uint8_t*, const float*, const float*, const float*, const float*,
                                      const float*, float*, float*, float*, float*, float*, double*, float*, int, int, int, int, int, int,
                                      uint32_t*, void*);
extern "C" int mval_bn_apply_fwd_p2(const float*, const float*, const float*, const float*, const float*, const float*, const float*, float*,
                                    void*, uint32_t*, int, int, int, int, int, int, uint32_t*, uint8_t*, const uint32_t*, const uint32_t*, void*);
extern "C" int mval_bn_apply_fwd_p2_res(const float*, const float*, const float*, const float*, const float*, const float*, const float*, float*,
                                        void*, uint32_t*, int, int, int, int, int, int, uint32_t*, uint8_t*, const uint32_t*, const uint32_t*,
                                        const void*, const uint32_t*, const void*, const uint32_t*, void*);
extern "C" int mval_bn_bwd_fused_p2(const float*, const float*, const uint8_t*, const float*, const float*, const float*, const float*,
                                    const float*, float*, float*, float*, float*, float*, double*, float*, int, int, int, int, int, int,
                                    uint32_t*, void*, uint32_t*, float*, uint32_t*, void*);
extern "C" int mval_bn_finalize_stats(const double*, int, int64_t, int, float, float, float*, float*, float*, float*, void*);
extern "C" int mval_bn_bwd_fused(const float*, const float*, const float*, const float*, const float*, const float*, const float*,
                                 float*, float*, float*, float*, float*, double*, float*, int, int, int, int, int, int, uint32_t*,
                                 void*);
#include <stdlib.h>
// The A/B switches of round 4's BatchNorm restructuring are decided ONCE, by the plan (engine_train.TrainPlan), and travel in
// mval_train_op.p2_flags: bit 7 = batch statistics by the separate pass over z instead of the forward conv's epilogue partials
// (MVAL_TRAIN_EPI_STATS=0), bit 6 = round 3's backward pair (masked copy to gz, dz in place) instead of mval_bn_bwd_fused
// (MVAL_TRAIN_BWD_FUSED=0).  (Round 4 read the environment here on every call as well: a switch flipped between plan build and a later
// step sent ops whose fp32 output was never written down the path that reads it -- ADVICE round 4.)
int mval_conv_wgrad_on(const float* x, const float* dz, float* dw, float* ws, int N, int Hin, int Win, int Cin, int Hout, int Wout, int Cout,
                       int k, int stride, int pad, int x_nchw, const uint32_t* x_amax_row, const uint32_t* dz_amax_row, hipStream_t s);
void mval_conv_wgrad_set_p2_x(const void* planes, const unsigned* rows);  // conv_wgrad.hip: x as P2 planes for the next weight gradient
void mval_conv_wgrad_set_p2_dz(const void* planes, const unsigned* rows);  // ... and dz
void mval_conv_wgrad_set_z_x(const float* mean, const float* invstd, const float* gamma, const float* beta, float sqrt_m1);  // conv_wgrad_bf3.hip: x = relu(BatchNorm(z))
struct WgradReduceJob {  // (conv_wgrad.hip)
  const float* slabs;
  float* dw;
  int PS, T, Cin, Cout, parts;
};
void mval_conv_wgrad_defer(WgradReduceJob* out);  // the next mval_conv_wgrad_on records its slab reduction instead of launching it
int mval_wgrad_reduce_jobs(const WgradReduceJob* jobs, int n, hipStream_t s);
extern "C" size_t mval_conv_wgrad_workspace_floats(int cin, int cout, int k);
void mval_bn_bwd_set_presummed(const double* part, const float* gmaxc, int nslots);  // train_ops.hip: the next fused BatchNorm backward finds its reduction done
// ---- measurement mode (bench.py, the c3 line's per-kernel roofline): hipEvents around every launch group of a
// training step, summed per kernel family.  Off unless mval_train_timing() armed it; the events are resolved (one
// stream synchronisation) at the end of each forward / backward call.
#include <vector>
enum { TT_CONV_FWD = 0, TT_BN_STATS, TT_BN_APPLY, TT_BN_BWD, TT_WGRAD, TT_DGRAD, TT_N };
static float* g_tt_out = nullptr;
static float* g_tt_ops = nullptr;  // optional [n_ops][TT_N] per-operator breakdown (tools/train_op_times.py)
static int g_tt_n_ops = 0, g_tt_op = -1;
struct TtSpan { int cat, op; hipEvent_t a, b; };
static std::vector<TtSpan> g_tt_spans;
struct TtScope {
  int cat; hipStream_t s; hipEvent_t a = nullptr, b = nullptr;
  TtScope(int c, hipStream_t st) : cat(c), s(st) {
    if (!g_tt_out) return;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    (void)hipEventRecord(a, s);
  }
  ~TtScope() {
    if (!a) return;
    (void)hipEventRecord(b, s);
    g_tt_spans.push_back({cat, g_tt_op, a, b});
  }
};
static void tt_flush() {
  if (!g_tt_out) return;
  for (auto& sp : g_tt_spans) {
    (void)hipEventSynchronize(sp.b);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, sp.a, sp.b);
    g_tt_out[sp.cat] += ms;
    if (g_tt_ops && sp.op >= 0 && sp.op < g_tt_n_ops) g_tt_ops[sp.op * TT_N + sp.cat] += ms;
    (void)hipEventDestroy(sp.a);
    (void)hipEventDestroy(sp.b);
  }
  g_tt_spans.clear();
}
extern "C" int mval_train_timing(float* ms_per_family) {
  g_tt_out = ms_per_family;
  g_tt_ops = nullptr;
  g_tt_n_ops = 0;
  if (ms_per_family)
    for (int i = 0; i < TT_N; i++) ms_per_family[i] = 0.f;
  return 0;
}
// The same with a per-operator breakdown: ms_per_op_family is a HOST array [n_ops][6], indexed by the op's position in
// the list handed to mval_train_forward (segmented backward calls must pass the op base, mval_train_timing_base).
static int g_tt_base = 0;
extern "C" int mval_train_timing_ops(float* ms_per_family, float* ms_per_op_family, int n_ops) {
  mval_train_timing(ms_per_family);
  g_tt_ops = ms_per_family ? ms_per_op_family : nullptr;
  g_tt_n_ops = g_tt_ops ? n_ops : 0;
  for (int i = 0; i < g_tt_n_ops * TT_N; i++) g_tt_ops[i] = 0.f;
  return 0;
}
extern "C" int mval_train_timing_base(int first_op) {
  g_tt_base = first_op;
  return 0;
}

// ---- bound-slack probe of the P2 training plan (engine_train.TrainPlan: first step of a plan, then every few hundred steps) ----
// While armed, every P2 tensor a step writes -- an op's output planes after its BatchNorm apply, its dz planes after its BatchNorm
// backward -- is measured by mval_p2_plane_stats into dev_out[op][0 | 1][4] (dwords, zeroed by the caller; op = position in the list
// handed to mval_train_forward; backward calls on a sub-range pass their base with mval_train_timing_base).
extern "C" int mval_p2_plane_stats(const void*, const uint32_t*, int, int, int, uint32_t*, void*);
static uint32_t* g_probe = nullptr;
static int g_probe_n = 0;
extern "C" int mval_train_p2_probe(uint32_t* dev_out, int n_ops) {
  g_probe = n_ops > 0 ? dev_out : nullptr;
  g_probe_n = g_probe ? n_ops : 0;
  return 0;
}

static void geometry(ConvArgs& a, const mval_op& op, int n_images) {
  a.N = n_images;
  a.Hin = op.hin; a.Win = op.win; a.Cin = op.cin;
  a.Hout = op.hout; a.Wout = op.wout; a.Cout = op.cout;
  a.k = op.k; a.stride = op.stride; a.pad = op.pad;
  a.up = 0; a.relu = 0; a.in_nchw = op.in_nchw; a.out_nchw = 0;
  a.dil = 1;
  a.th = a.tw = a.tn = a.tw_log2 = a.thw_log2 = a.tiles_x = a.tiles_y = 0;
  a.G_total = (op.cin + 15) / 16;
  a.NS_total = (op.cout + 15) / 16;
  a.res1 = a.res2 = nullptr;
  // bias-free accumulation: training gradients amplify a coherent -1 ulp (conv_mfma_split.hip).  (The fp16 split passes
  // every training test without it as well, at the same speed: 84.30 vs 84.32 ms per C3 step; kept.)
  a.precise = 1;
}

// ConvTranspose2d forward on the matrix cores: stride-1 conv over the zero-dilated input (net.hip)
static void deconv_as_conv(ConvArgs& a, const mval_op& op) {
  a.dil = op.stride;
  a.stride = 1;
  a.pad = op.k - 1 - op.pad;
}

static int run_conv(const ConvArgs& a0, int algo, hipStream_t s, const char* what) {
  ConvArgs a = a0;
  if (algo == MVAL_ALGO_MFMA_BF3 || algo == MVAL_ALGO_MFMA_H2) {
    // bf16x3: scale-free; fp16x2: the caller set a.in_amax (ONE magnitude row for the tensor: in_amax_stride 0) and
    // a.w_unscale.  Both have the bias-free accumulate mode (a.precise) the training forward needs.
    a.planes = algo == MVAL_ALGO_MFMA_H2 ? 2 : 3;
    if (a.planes == 2 && (!a.in_amax || !a.w_unscale)) {
      mval_set_error("%s: the fp16-split conv needs its input's magnitude row and the packed weights' trailer", what);
      return -1;
    }
    if (mval_launch_conv_split(a, s)) {
      mval_set_error("%s: no bf16x3 MFMA configuration (k%d cin%d cout%d dil%d)", what, a.k, a.Cin, a.Cout, a.dil);
      return -1;
    }
  } else if (algo == MVAL_ALGO_MFMA) {
    if (mval_launch_conv_mfma(a, s)) {
      mval_set_error("%s: no MFMA configuration (k%d cin%d cout%d dil%d)", what, a.k, a.Cin, a.Cout, a.dil);
      return -1;
    }
  } else if (mval_launch_conv_stem(a, s)) {
    mval_launch_conv_direct(a, MVAL_OP_CONV, s);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    mval_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return -2;
  }
  return 0;
}

// Lanes (round 5): ops of one phase that sit on different lanes (graph.py: HRNet branches, fuse outputs) are independent; with
// n_lanes > 1 an op whose p2_flags carries MVAL_TRAIN_LANE_FWD (forward) / MVAL_TRAIN_LANE_BWD (backward) runs on its lane's side
// stream with its lane's slice of every scratch buffer, phases join on the caller's stream (MvalLaneWalk, as the inference executor).
// The deep branches' kernels (8 x 8 / 16 x 16 maps: a few MB per tensor) and the ~880 one-workgroup-per-channel finalize launches
// leave most of the chip idle on their own; next to another branch's kernels they fill its tails.  What each op computes, and the
// order in which every gradient slot is written, does not change: the plan marks a phase for the backward only if each gradient
// slot it writes has all its writers on one lane.
// MVAL_TRAIN_LANE_ORD (backward): the op belongs to a phase whose lanes DO share gradient slots (the chains of a fuse layer all add into
// the branches' output gradients; a transition's convs into the last branch's).  Its lanes still run on their streams, but every kernel
// that writes a gradient slot first waits for the slot's previous writer of this phase (in list order, the order of the one-stream pass)
// and leaves an event behind for the next one: each slot sees its first-touch store and its accumulations in the same order as before,
// whichever stream they come from.  Nothing in a phase READS a slot another lane of the phase writes (an op reads its own output's
// gradient, written by its consumers: later phases, or its own lane's chain).
struct SlotOrder {
  struct Entry { int64_t off; hipEvent_t ev; hipStream_t s; };
  std::vector<Entry> last;
  int phase = -0x7fffffff;
  static std::vector<hipEvent_t>& pool() {
    static std::vector<hipEvent_t> p[16];
    int dev = 0;
    (void)hipGetDevice(&dev);
    return p[dev & 15];
  }
  bool keep = false;  // (free-running lanes: the events carry the dependencies across phases)
  void enter(int ph) {
    if (ph != phase && !keep) {  // (phases join on the caller's stream: everything before is ordered)
      last.clear();
      phase = ph;
    }
  }
  void before(int64_t off, hipStream_t s) {
    if (off < 0) return;
    for (auto& e : last)
      if (e.off == off && e.s != s) (void)hipStreamWaitEvent(s, e.ev, 0);
  }
  void after(int64_t off, hipStream_t s) {
    if (off < 0) return;
    for (auto& e : last)
      if (e.off == off) {
        (void)hipEventRecord(e.ev, s);
        e.s = s;
        return;
      }
    auto& p = pool();
    if (last.size() >= p.size()) {
      hipEvent_t ev = nullptr;
      (void)hipEventCreateWithFlags(&ev, hipEventDisableTiming);
      p.push_back(ev);
    }
    hipEvent_t ev = p[last.size()];
    (void)hipEventRecord(ev, s);
    last.push_back({off, ev, s});
  }
};

// MVAL_TRAIN_LANE_FREE (backward, on every op of the call): no join at the phase changes at all -- the side streams fork once from the
// caller's stream and join once at the end; EVERY dependency travels through the slot events: an op waits for the last writer of the
// gradient slot it reads (its own output's gradient; the writers of a slot are chained, so the last one implies them all) and orders its
// own slot writes as under MVAL_TRAIN_LANE_ORD.  A short lane (the 8 x 8 branch) then runs ahead into the next fuse layer's chain while
// the 64 x 64 branch is still in its blocks.
struct LanesFree {
  MvalLanes* L;
  hipStream_t main_s;
  bool used[MVAL_MAX_LANES] = {false, false, false, false};
  LanesFree(MvalLanes* lanes, hipStream_t s) : L(lanes), main_s(s) {
    if (L) (void)hipEventRecord(L->fork_ev, main_s);
  }
  hipStream_t stream_for(int lane) {
    if (!L || lane <= 0 || lane >= MVAL_MAX_LANES) return main_s;
    if (!used[lane]) {
      (void)hipStreamWaitEvent(L->side[lane], L->fork_ev, 0);
      used[lane] = true;
    }
    return L->side[lane];
  }
  void finish() {
    if (!L) return;
    for (int l = 1; l < MVAL_MAX_LANES; l++)
      if (used[l]) {
        (void)hipEventRecord(L->join_ev[l], L->side[l]);
        (void)hipStreamWaitEvent(main_s, L->join_ev[l], 0);
        used[l] = false;
      }
  }
  ~LanesFree() { finish(); }
};

static int lane_of(const mval_train_op& t, int bit, int n_lanes) {
  return (n_lanes > 1 && (t.p2_flags & bit) && t.op.lane > 0 && t.op.lane < n_lanes) ? t.op.lane : 0;
}

#ifdef MVAL_TRAIN_ABLATE
#include <stdio.h>
extern int g_train_ablate;
static int g_train_fwd_calls = 0, g_train_ablated = 0;
#endif
static int train_forward(const mval_train_op* ops, int n_ops, int n_images, float* arena, const float* params,
                         int64_t ones_off, int64_t zeros_off, const float* input_nchw, float* output_nchw,
                         double* ws0, int64_t ws_doubles, int n_lanes, float momentum, float eps, void* stream0) {
  MVAL_REQUIRE(ops && n_ops > 0 && n_images > 0 && n_lanes >= 1 && n_lanes <= MVAL_MAX_LANES, "mval_train_forward: bad arguments");
  if (g_tt_out) n_lanes = 1;  // (measurement mode times every launch group on its own: one stream)
  MvalLanes* L = n_lanes > 1 ? mval_device_lanes() : nullptr;
  if (n_lanes > 1) MVAL_REQUIRE(L != nullptr, "mval_train_forward: could not create the side streams");
  // (A forward without joins at the phase changes -- every op behind the events of its producers, as the backward's MVAL_TRAIN_LANE_FREE --
  // measured the same or slower: C3 61.3-61.6 ms against 61.0-61.3 with the joins, profiles/r05/train_lanes_free2.log.)
  MvalLaneWalk walk(L, mval_stream(stream0));
#ifdef MVAL_TRAIN_ABLATE
  {
    const char* e = getenv("MVAL_TRAIN_ABL");
    g_train_ablate = e ? atoi(e) : 0;
    g_train_fwd_calls++;
    if (g_train_fwd_calls == 4 && g_train_ablate) fprintf(stderr, "[ablate %d] forward applies skipped per step: %d\n", g_train_ablate, g_train_ablated);
    g_train_ablated = 0;
  }
#endif
  for (int i = 0; i < n_ops; i++) {
    g_tt_op = i;
    const mval_train_op& t = ops[i];
    const mval_op& op = t.op;
    const int lane = lane_of(t, MVAL_TRAIN_LANE_FWD, n_lanes);
    hipStream_t s = walk.stream_for(op.phase, lane);
    void* stream = reinterpret_cast<void*>(s);
    double* ws = ws0 + (int64_t)lane * ws_doubles;
    const bool epi_stats = !(t.p2_flags & 128);  // (bit 7: batch statistics by the separate pass over z -- the plan's decision, MVAL_TRAIN_EPI_STATS=0)
    ConvArgs a = {};
    geometry(a, op, n_images);
    a.in = op.in_off >= 0 ? arena + op.in_off : input_nchw;
    if (op.kind == MVAL_OP_MAXPOOL) {
      MVAL_REQUIRE(op.out_off >= 0 && op.in_off >= 0, "mval_train_forward: op %d: max-pool on an external buffer", i);
      a.w = a.scale = a.shift = nullptr;
      a.out = arena + op.out_off;
      mval_launch_conv_direct(a, MVAL_OP_MAXPOOL, s);
      if (t.out_amax_off > 0)  // (one row for the tensor: "one image" of the whole size)
        mval_launch_amax(a.out, (int64_t)n_images * op.hout * op.wout * op.cout, 1, reinterpret_cast<unsigned*>(arena + t.out_amax_off), s);
      continue;
    }
    if (op.kind == MVAL_OP_DECONV) {
      MVAL_REQUIRE(t.has_bn && op.algo == MVAL_ALGO_MFMA, "mval_train_forward: op %d: transposed conv needs BN + the MFMA form", i);
      deconv_as_conv(a, op);
    }
    a.w = params + op.w_off;
    if (op.algo == MVAL_ALGO_MFMA_H2) {
      MVAL_REQUIRE(op.in_amax_off > 0, "mval_train_forward: op %d: fp16-split conv without its input's magnitude row", i);
      a.in_amax = reinterpret_cast<const unsigned*>(arena + op.in_amax_off);
      a.w_unscale = a.w + mval_packed_weight_floats(MVAL_PACK_MFMA16_H2, op.cout, op.cin, op.k) - 4;
    }
    float* out = op.out_off >= 0 ? arena + op.out_off : output_nchw;
    if (t.has_bn) {
      a.out = arena + t.z_off;
      a.scale = params + ones_off;
      a.shift = params + zeros_off;
      int rc;
      // batch statistics: (sum, sum of squares) partials per conv workgroup from the epilogue that stores z, where the
      // kernel the launcher picks can keep them (tiles > 0 afterwards); else the separate pass over z
      int tiles = 0;
      if (epi_stats && op.kind == MVAL_OP_CONV) {
        a.bn_part = ws;
        a.bn_part_cap = ws_doubles;
        a.bn_tiles_host = &tiles;
      }
      if (t.fwd_p2) {
        // the conv on the P2 kernels: input planes written by the producer's apply, raw z as fp32 NHWC, statistics partials per
        // (persistent workgroup, pixel wave)
        MVAL_REQUIRE(op.kind == MVAL_OP_CONV && t.in_p2_off > 0 && t.in_p2_rows_off > 0, "mval_train_forward: op %d: P2 conv without its input planes", i);
        P2Args p = {};
        p.in = reinterpret_cast<const _Float16*>(arena + t.in_p2_off);
        p.in_row = reinterpret_cast<const unsigned*>(arena + t.in_p2_rows_off);
        if (t.zin_rel) {  // (round 6) the producer's apply was skipped: this conv applies its BatchNorm + ReLU while staging the raw z
          const mval_train_op& pr = (&t)[t.zin_rel];
          MVAL_REQUIRE(i + t.zin_rel >= 0 && t.zin_rel < 0 && pr.z_out && pr.has_bn && pr.op.cout == op.cin && pr.op.hout == op.hin && pr.op.wout == op.win,
                       "mval_train_forward: op %d: zin_rel %d does not name its producer", i, t.zin_rel);
          p.in = nullptr;
          p.in_z = arena + pr.z_off;
          p.zin_mean = pr.mean; p.zin_invstd = pr.invstd; p.zin_gamma = pr.gamma; p.zin_beta = pr.beta;
          const double Mp = (double)n_images * pr.op.hout * pr.op.wout;
          p.zin_sqrt_m1 = (float)sqrt(Mp > 1 ? Mp - 1.0 : 1.0);
        }
        p.w = params + op.w_off;
        p.w_unscale = p.w + mval_packed_weight_floats(MVAL_PACK_MFMA16_H2, op.cout, op.cin, op.k) - 4;
        p.scale = params + ones_off;
        p.shift = params + zeros_off;
        p.out_nhwc = arena + t.z_off;
        p.N = n_images; p.Hin = op.hin; p.Win = op.win; p.Cin = op.cin; p.Hout = op.hout; p.Wout = op.wout; p.Cout = op.cout;
        p.k = op.k; p.stride = op.stride;
        if (epi_stats) {
          p.bn_part = ws;
          p.bn_part_cap = ws_doubles;
          p.bn_slots_host = &tiles;
        }
        TtScope tt(TT_CONV_FWD, s);
        if (mval_launch_conv_p2(p, s)) {
          mval_set_error("mval_train_forward: op %d: no P2 kernel (k%d s%d cin%d cout%d %dx%d)", i, op.k, op.stride, op.cin, op.cout, op.hin, op.win);
          return -1;
        }
        rc = 0;
      } else {
        TtScope tt(TT_CONV_FWD, s);
        rc = run_conv(a, op.algo, s, "mval_train_forward/conv");
      }
      if (rc) return rc;
      const int64_t M = (int64_t)n_images * op.hout * op.wout;
      {
        TtScope tt(TT_BN_STATS, s);
        if (tiles > 0)
          rc = mval_bn_finalize_stats(ws, tiles, M, op.cout, eps, momentum, t.mean, t.invstd, t.running_mean, t.running_var,
                                      stream);
        else
          rc = mval_bn_batch_stats(a.out, M, op.cout, eps, momentum, t.mean, t.invstd, t.running_mean, t.running_var, ws,
                                   stream);
      }
      if (rc) return rc;
      // (round 6) the one reader of this activation applies the BatchNorm itself (mval_train_op.z_out) -- except on the plan's bound-slack
      // probe steps (first step, then every 1 024th): the planes are written then so that the probe measures this tensor like every other
      if (t.z_out && !(g_probe && i < g_probe_n)) continue;
      TtScope tt(TT_BN_APPLY, s);
#ifdef MVAL_TRAIN_ABLATE
      // measurement build only (profiles/r06: the step-level UPPER BOUND of BatchNorm-apply-in-the-consumer's-staging): from the third
      // forward on, the apply of every residual-free ReLU op whose output exists as planes only is skipped -- its consumers read stale planes
      if ((g_train_ablate & 1) && g_train_fwd_calls > 2 && t.out_p2_off > 0 && (t.p2_flags & 2) && op.res1_off < 0 && op.res2_off < 0 && op.relu && op.up == 0) {
        g_train_ablated++;
        continue;
      }
#endif
      if (t.out_p2_off > 0) {
        // (p2_flags bit 4 / 5) a residual that exists as planes only is read from those
        const bool r1p = (t.p2_flags & 16) && op.res1_off >= 0 && t.res1_p2_off > 0, r2p = (t.p2_flags & 32) && op.res2_off >= 0 && t.res2_p2_off > 0;
        rc = mval_bn_apply_fwd_p2_res(a.out, t.mean, t.invstd, t.gamma, t.beta, (op.res1_off >= 0 && !r1p) ? arena + op.res1_off : nullptr,
                                  (op.res2_off >= 0 && !r2p) ? arena + op.res2_off : nullptr, (t.p2_flags & 2) ? nullptr : out, arena + t.out_p2_off,
                                  reinterpret_cast<uint32_t*>(arena + t.out_p2_rows_off), n_images, op.hout, op.wout, op.cout, op.up, op.relu,
                                  t.out_amax_off > 0 ? reinterpret_cast<uint32_t*>(arena + t.out_amax_off) : nullptr,
                                  t.mask_off > 0 ? reinterpret_cast<uint8_t*>(arena + t.mask_off) : nullptr,
                                  t.res1_amax_off > 0 ? reinterpret_cast<const uint32_t*>(arena + t.res1_amax_off) : nullptr,
                                  t.res2_amax_off > 0 ? reinterpret_cast<const uint32_t*>(arena + t.res2_amax_off) : nullptr,
                                  r1p ? arena + t.res1_p2_off : nullptr, r1p ? reinterpret_cast<const uint32_t*>(arena + t.res1_p2_rows_off) : nullptr,
                                  r2p ? arena + t.res2_p2_off : nullptr, r2p ? reinterpret_cast<const uint32_t*>(arena + t.res2_p2_rows_off) : nullptr, stream);
        if (!rc && g_probe && i < g_probe_n)
          rc = mval_p2_plane_stats(arena + t.out_p2_off, reinterpret_cast<const uint32_t*>(arena + t.out_p2_rows_off), n_images, op.cout,
                                   (op.hout << op.up) * (op.wout << op.up), g_probe + (int64_t)i * 8, stream);
      }
      else
      rc = mval_bn_apply_fwd_mask(a.out, t.mean, t.invstd, t.gamma, t.beta, op.res1_off >= 0 ? arena + op.res1_off : nullptr,
                                  op.res2_off >= 0 ? arena + op.res2_off : nullptr, out, n_images, op.hout, op.wout, op.cout,
                                  op.up, op.relu, t.out_amax_off > 0 ? reinterpret_cast<uint32_t*>(arena + t.out_amax_off) : nullptr,
                                  t.mask_off > 0 ? reinterpret_cast<uint8_t*>(arena + t.mask_off) : nullptr, stream);
      if (rc) return rc;
    } else {
      MVAL_REQUIRE(op.up == 0, "mval_train_forward: op %d: upsample without BatchNorm is not part of any graph", i);
      a.out = out;
      a.out_nchw = op.out_nchw;
      a.relu = op.relu;
      a.scale = params + op.scale_off;
      a.shift = params + op.shift_off;
      a.res1 = op.res1_off >= 0 ? arena + op.res1_off : nullptr;
      a.res2 = op.res2_off >= 0 ? arena + op.res2_off : nullptr;
      TtScope tt(TT_CONV_FWD, s);
      int rc = run_conv(a, op.algo, s, "mval_train_forward/conv");
      if (rc) return rc;
    }
  }
  walk.finish();
  tt_flush();
  return 0;
}

extern "C" int mval_train_forward(const mval_train_op* ops, int n_ops, int n_images, float* arena, const float* params,
                                  int64_t ones_off, int64_t zeros_off, const float* input_nchw, float* output_nchw,
                                  double* ws, int64_t ws_doubles, float momentum, float eps, void* stream) {
  return train_forward(ops, n_ops, n_images, arena, params, ones_off, zeros_off, input_nchw, output_nchw, ws, ws_doubles, 1, momentum, eps, stream);
}

extern "C" int mval_train_forward_lanes(const mval_train_op* ops, int n_ops, int n_images, float* arena, const float* params,
                                        int64_t ones_off, int64_t zeros_off, const float* input_nchw, float* output_nchw,
                                        double* ws, int64_t ws_doubles_per_lane, int n_lanes, float momentum, float eps, void* stream) {
  return train_forward(ops, n_ops, n_images, arena, params, ones_off, zeros_off, input_nchw, output_nchw, ws, ws_doubles_per_lane, n_lanes, momentum,
                       eps, stream);
}

// (measurement only: MVAL_WGRAD_SLAB_ROT, engine_train.py -- the weight gradients' slab workspace of a lane as K regions walked op by op)
static int g_slab_rot = 1;
static int64_t g_slab_region = 0;
extern "C" int mval_train_slab_rotation(int k, int64_t region_floats) {
  g_slab_rot = k > 1 ? k : 1;
  g_slab_region = region_floats;
  return 0;
}

static int train_backward(const mval_train_op* ops, int n_ops, int n_images, float* arena, float* garena,
                          const float* params, int64_t ones_off, int64_t zeros_off, const float* input_nchw,
                          float* gz0, float* wsf0, double* ws0, float* sums0, int n_lanes, int64_t gz_stride, int64_t wsf_stride,
                          int64_t ws_stride, int64_t sums_stride, void* stream0) {
  MVAL_REQUIRE(ops && n_ops > 0 && n_images > 0 && garena && gz0 && wsf0 && ws0 && sums0 && n_lanes >= 1 && n_lanes <= MVAL_MAX_LANES,
               "mval_train_backward: bad arguments");
  const int n_lanes_arg = n_lanes;  // (the workspace holds this many slices whatever the measurement mode does to n_lanes)
  if (g_tt_out) n_lanes = 1;
  MvalLanes* L = n_lanes > 1 ? mval_device_lanes() : nullptr;
  if (n_lanes > 1) MVAL_REQUIRE(L != nullptr, "mval_train_backward: could not create the side streams");
  const bool free_run = n_lanes > 1 && (ops[n_ops - 1].p2_flags & MVAL_TRAIN_LANE_FREE);
  MvalLaneWalk walk(free_run ? nullptr : L, mval_stream(stream0));
  LanesFree lanes_free(free_run ? L : nullptr, mval_stream(stream0));
  SlotOrder order;
  order.keep = free_run;
  // (round 6, MVAL_TRAIN_BSUM) per lane: the slots of reduction partials the last data gradient on that lane left in the lane's `ws` for
  // the op `presum_for` -- the very next op of the list (zin_rel == -1), so nothing else touches that scratch in between
  // (round 6, MVAL_TRAIN_WGRAD_DEFER on the call's last op) every op's weight-gradient slabs go to a region of their own inside the whole
  // workspace (all lanes' slices as ONE arena, walked in list order) and the reductions run as one launch per 64 ops when the call's lanes
  // have joined -- 293 seven-microsecond launches per step become one per backward segment
  const bool defer_reduce = !g_tt_out && wsf_stride > 0 && (ops[n_ops - 1].p2_flags & MVAL_TRAIN_WGRAD_DEFER);
  std::vector<WgradReduceJob> rjobs;
  if (defer_reduce) rjobs.reserve(n_ops);
  int64_t slab_off = 0;
  const int64_t slab_cap = wsf_stride * (int64_t)n_lanes_arg;
  int presum_slots[MVAL_MAX_LANES] = {0, 0, 0, 0};
  const mval_train_op* presum_for[MVAL_MAX_LANES] = {nullptr, nullptr, nullptr, nullptr};
  for (int i = n_ops - 1; i >= 0; i--) {
    g_tt_op = g_tt_base + i;
    const mval_train_op& t = ops[i];
    const mval_op& op = t.op;
    const int lane = lane_of(t, MVAL_TRAIN_LANE_BWD, n_lanes);
    hipStream_t s = free_run ? lanes_free.stream_for(lane) : walk.stream_for(op.phase, lane);
    const bool ord = n_lanes > 1 && (free_run || (t.p2_flags & MVAL_TRAIN_LANE_ORD));
    if (ord) order.enter(op.phase);
    if (free_run) order.before(t.gout_off, s);  // (the op reads its output's gradient: behind the slot's last writer)
    void* stream = reinterpret_cast<void*>(s);
    float* gz = gz0 + (int64_t)lane * gz_stride;
    float* wsf = wsf0 + (int64_t)lane * wsf_stride + (g_slab_rot > 1 ? (int64_t)((g_tt_base + i) % g_slab_rot) * g_slab_region : 0);
    double* ws = ws0 + (int64_t)lane * ws_stride;
    float* sums = sums0 + (int64_t)lane * sums_stride;
    const bool bwd_fused = !(t.p2_flags & 64);  // (bit 6: round 3's backward pair -- the plan's decision, MVAL_TRAIN_BWD_FUSED=0; it reads `out`)
    MVAL_REQUIRE(bwd_fused || !(t.p2_flags & (2 | 4 | 8)), "mval_train_backward: op %d: the round-3 BatchNorm backward with a P2-only output / P2 dz", i);
    MVAL_REQUIRE(t.gout_off >= 0, "mval_train_backward: op %d has no output gradient slot", i);
    if (op.kind == MVAL_OP_MAXPOOL) {
      if (t.gin_off >= 0) {
        if (ord) order.before(t.gin_off, s);
        int rc = mval_maxpool_bwd(garena + t.gout_off, arena + op.in_off, garena + t.gin_off, n_images, op.hin, op.win,
                                  op.cin, op.hout, op.wout, op.k, op.stride, op.pad, !(t.first_touch & 1), stream);
        if (rc) return rc;
        if (ord) order.after(t.gin_off, s);
      }
      continue;
    }
    // the output activation is only needed for the ReLU mask; NCHW outputs (final layer) have none
    const float* outp = op.out_off >= 0 ? arena + op.out_off : nullptr;
    MVAL_REQUIRE(!(op.relu && !outp), "mval_train_backward: op %d: ReLU on an external output", i);
    uint32_t* gz_row = (t.has_bn && t.gz_amax_off > 0) ? reinterpret_cast<uint32_t*>(arena + t.gz_amax_off) : nullptr;
    int rc;
    // (p2_flags bit 2) the data gradient on the P2 kernels: the BatchNorm backward also leaves dz as P2 planes
    const bool dz_p2 = (t.p2_flags & 4) && t.gz_p2_off > 0 && t.gin_off >= 0 && bwd_fused && t.has_bn && op.up == 0;
    if (ord) {  // (the BatchNorm backward scatters the residual gradients)
      order.before(t.gres1_off, s);
      order.before(t.gres2_off, s);
    }
    {
    TtScope tt(TT_BN_BWD, s);
    if (presum_for[lane] == &t && presum_slots[lane] > 0 && bwd_fused && dz_p2)
      mval_bn_bwd_set_presummed(ws, reinterpret_cast<const float*>(ws + (int64_t)op.cout * presum_slots[lane] * 2), presum_slots[lane]);
    presum_for[lane] = nullptr;
    presum_slots[lane] = 0;
    if (bwd_fused && t.has_bn && op.up == 0 && (op.cout & 3) == 0)
      rc = mval_bn_bwd_fused_p2(garena + t.gout_off, outp, t.mask_off > 0 ? reinterpret_cast<const uint8_t*>(arena + t.mask_off) : nullptr,
                             arena + t.z_off, t.mean, t.invstd, t.gamma, t.beta,
                             t.gres1_off >= 0 ? garena + t.gres1_off : nullptr, t.gres2_off >= 0 ? garena + t.gres2_off : nullptr,
                             (dz_p2 && (t.p2_flags & 8)) ? nullptr : gz,  // (bit 3: the weight gradient reads the planes too: no fp32 dz)
                             t.dgamma, t.dbeta, ws, sums, n_images, op.hout, op.wout, op.cout, op.relu, t.first_touch >> 1, gz_row,
                             dz_p2 ? arena + t.gz_p2_off : nullptr, dz_p2 ? reinterpret_cast<uint32_t*>(arena + t.gz_p2_rows_off) : nullptr,
                             dz_p2 ? arena + t.gz_p2_rows_off + (int64_t)n_images * P2_ROW : nullptr,
                             dz_p2 ? reinterpret_cast<uint32_t*>(arena + t.gz_p2_rows_off + (int64_t)n_images * P2_ROW + 512) : nullptr, stream);
    else
    rc = mval_bn_bwd_amax(garena + t.gout_off, outp, t.has_bn ? arena + t.z_off : nullptr, t.mean, t.invstd, t.gamma,
                              t.gres1_off >= 0 ? garena + t.gres1_off : nullptr,
                              t.gres2_off >= 0 ? garena + t.gres2_off : nullptr, gz, t.dgamma, t.dbeta, ws, sums, n_images,
                              op.hout, op.wout, op.cout, op.up, op.relu, t.has_bn, t.first_touch >> 1, gz_row, stream);
    }
    if (rc) return rc;
    if (ord) {
      order.after(t.gres1_off, s);
      order.after(t.gres2_off, s);
    }
    if (dz_p2 && g_probe && g_tt_base + i < g_probe_n) {
      rc = mval_p2_plane_stats(arena + t.gz_p2_off, reinterpret_cast<const uint32_t*>(arena + t.gz_p2_rows_off), n_images, op.cout,
                               op.hout * op.wout, g_probe + (int64_t)(g_tt_base + i) * 8 + 4, stream);
      if (rc) return rc;
    }
    const float* x = op.in_off >= 0 ? arena + op.in_off : input_nchw;
    if (op.kind == MVAL_OP_DECONV) {
      // ConvTranspose2d(k, s, p): y = scatter of x through W[cin][cout][k][k].  With the roles swapped it
      // is the conv  x' = conv(y', W as [cout' = cin][cin' = cout], stride s, pad p), so
      //   dW = weight gradient of that conv with input dz (at the output resolution) and "dz" = x;
      //   dx = that conv applied to dz (plain stride-2 conv on the matrix cores).
      float* wsd = wsf;
      if (defer_reduce) {
        const int64_t need = ((int64_t)mval_conv_wgrad_workspace_floats(op.cout, op.cin, op.k) + 63) & ~(int64_t)63;
        MVAL_REQUIRE(slab_off + need <= slab_cap, "mval_train_backward: op %d: the slab workspace does not hold this call's deferred reductions", i);
        wsd = wsf0 + slab_off;
        slab_off += need;
        rjobs.emplace_back();
        mval_conv_wgrad_defer(&rjobs.back());
      }
      rc = mval_conv_wgrad(gz, x, t.dweight, wsd, n_images, op.hout, op.wout, op.cout, op.hin, op.win, op.cin, op.k,
                           op.stride, op.pad, 0, stream);
      if (rc) return rc;
      if (t.gin_off >= 0) {
        ConvArgs a = {};
        a.N = n_images;
        a.Hin = op.hout; a.Win = op.wout; a.Cin = op.cout;
        a.Hout = op.hin; a.Wout = op.win; a.Cout = op.cin;
        a.k = op.k; a.stride = op.stride; a.pad = op.pad;
        a.up = 0; a.relu = 0; a.in_nchw = 0; a.out_nchw = 0;
        a.dil = 1;
        a.th = a.tw = a.tn = a.tw_log2 = a.thw_log2 = a.tiles_x = a.tiles_y = 0;
        a.G_total = (op.cout + 15) / 16;
        a.NS_total = (op.cin + 15) / 16;
        a.in = gz;
        a.w = params + t.wd_off;
        a.scale = params + ones_off;
        a.shift = params + zeros_off;
        a.out = garena + t.gin_off;
        a.res1 = (t.first_touch & 1) ? nullptr : a.out;
        a.res2 = nullptr;
        if (ord) order.before(t.gin_off, s);
        rc = run_conv(a, MVAL_ALGO_MFMA, s, "mval_train_backward/deconv dgrad");
        if (rc) return rc;
        if (ord) order.after(t.gin_off, s);
      }
      continue;
    }
    // fp16-split weight gradient when the magnitudes of both operands are at hand (x's row from its producer, dz's
    // from this op's BatchNorm backward)
    const uint32_t* x_row = (gz_row && op.in_amax_off > 0) ? reinterpret_cast<const uint32_t*>(arena + op.in_amax_off) : nullptr;
    {
    TtScope tt(TT_WGRAD, s);
    if (dz_p2 && (t.p2_flags & 8))
      mval_conv_wgrad_set_p2_dz(arena + t.gz_p2_off, reinterpret_cast<const unsigned*>(arena + t.gz_p2_rows_off));
    const float* xw = x;
    if (t.zin_rel) {  // (round 6) x = relu(BatchNorm(z)) of the producer, applied by the weight gradient's staging
      const mval_train_op& pr = (&t)[t.zin_rel];
      MVAL_REQUIRE(dz_p2 && (t.p2_flags & 8), "mval_train_backward: op %d: zin_rel needs the weight gradient that reads dz from planes", i);
      const double Mp = (double)n_images * pr.op.hout * pr.op.wout;
      mval_conv_wgrad_set_z_x(pr.mean, pr.invstd, pr.gamma, pr.beta, (float)sqrt(Mp > 1 ? Mp - 1.0 : 1.0));
      xw = arena + pr.z_off;
    } else if (t.fwd_p2 && (t.p2_flags & 1))  // (wgrad_p2: this op's input exists as P2 planes and its weight gradient reads those)
      mval_conv_wgrad_set_p2_x(arena + t.in_p2_off, reinterpret_cast<const unsigned*>(arena + t.in_p2_rows_off));
    float* wsl = wsf;
    if (defer_reduce) {
      const int64_t need = ((int64_t)mval_conv_wgrad_workspace_floats(op.cin, op.cout, op.k) + 63) & ~(int64_t)63;
      MVAL_REQUIRE(slab_off + need <= slab_cap, "mval_train_backward: op %d: the slab workspace (%lld floats) does not hold this call's deferred reductions", i,
                   (long long)slab_cap);
      wsl = wsf0 + slab_off;
      slab_off += need;
      rjobs.emplace_back();
      mval_conv_wgrad_defer(&rjobs.back());
    }
    rc = mval_conv_wgrad_on(xw, gz, t.dweight, wsl, n_images, op.hin, op.win, op.cin, op.hout, op.wout, op.cout, op.k, op.stride, op.pad,
                            op.in_nchw, (x_row && !t.zin_rel) ? x_row : nullptr, (x_row && !t.zin_rel) ? gz_row : nullptr, s);
    }
    if (rc) return rc;
    if (t.gin_off >= 0) {
      MVAL_REQUIRE(t.dgrad_algo != MVAL_ALGO_MFMA_H2 || gz_row, "mval_train_backward: op %d: fp16-split data gradient without dz's magnitude row", i);
      if (ord) order.before(t.gin_off, s);
      TtScope tt(TT_DGRAD, s);
      if (dz_p2) {
        // dx (+)= conv(dz, flip(W)^T) on conv_p2.hip: dz planes in, fp32 NHWC gradient slot out (accumulated in place unless first writer)
        P2Args p = {};
        p.in = reinterpret_cast<const _Float16*>(arena + t.gz_p2_off);
        p.in_row = reinterpret_cast<const unsigned*>(arena + t.gz_p2_rows_off);
        p.w = params + t.wd_off;
        p.w_unscale = p.w + mval_packed_weight_floats(MVAL_PACK_MFMA16_H2, op.cin, op.cout, op.k) - 4;
        p.scale = params + ones_off;
        p.shift = params + zeros_off;
        p.out_nhwc = garena + t.gin_off;
        p.acc_nhwc = !(t.first_touch & 1);
        p.N = n_images; p.Hin = op.hout; p.Win = op.wout; p.Cin = op.cout; p.Hout = op.hin; p.Wout = op.win; p.Cout = op.cin;
        p.k = op.k; p.stride = 1;
        int slots = 0;
        if ((t.p2_flags & MVAL_TRAIN_BSUM) && i > 0) {
          // this launch is the LAST writer of the producer's output gradient (the producer is the op in front of this one, so every other
          // reader of its output sits behind this op in the list and has written already; under lanes `order` keeps that sequence): it also
          // keeps that op's BatchNorm backward reduction.  Without residuals the ReLU mask comes from z; with residuals from the kept bits.
          const mval_train_op& pr = ops[i - 1];
          const bool has_res = pr.op.res1_off >= 0 || pr.op.res2_off >= 0;
          if (pr.has_bn && pr.op.relu && (!has_res || pr.mask_off > 0) && pr.op.up == 0 && pr.gout_off == t.gin_off && pr.gz_p2_rows_off > 0 &&
              !(pr.p2_flags & 64) && (pr.p2_flags & 4) && pr.gz_p2_off > 0 && pr.gin_off >= 0 && (pr.op.cout & 3) == 0 &&
              lane_of(pr, MVAL_TRAIN_LANE_BWD, n_lanes) == lane) {
            p.bs_z = arena + pr.z_off;
            p.bs_mask = has_res ? reinterpret_cast<const unsigned char*>(arena + pr.mask_off) : nullptr;
            p.bs_mean = pr.mean; p.bs_invstd = pr.invstd; p.bs_gamma = pr.gamma; p.bs_beta = pr.beta;
            p.bs_part = ws;
            p.bs_cap = ws_stride;  // (doubles of this lane's scratch; mval_train_backward without lanes passes 0: the plain data gradient)
            p.bs_slots_host = &slots;
            p.bs_bound_slot = reinterpret_cast<unsigned*>(arena + pr.gz_p2_rows_off + (int64_t)n_images * P2_ROW + 512);
          }
        }
        if (mval_launch_conv_p2(p, s)) {
          mval_set_error("mval_train_backward: op %d: no P2 kernel for the data gradient (k%d cin%d cout%d)", i, op.k, op.cin, op.cout);
          return -1;
        }
        if (slots > 0) {
          presum_slots[lane] = slots;
          presum_for[lane] = &ops[i - 1];
        }
        rc = 0;
      } else if (t.dgrad_form == 1)  // (dgrad_form: the four-parity form of a stride-2 3x3 data gradient)
        rc = mval_conv_dgrad_parity(gz, params + t.wd_off, params + ones_off, params + zeros_off, garena + t.gin_off,
                                    !(t.first_touch & 1), n_images, op.hin, op.win, op.cin, op.hout, op.wout, op.cout, t.dgrad_algo,
                                    gz_row, stream);
      else
      rc = mval_conv_dgrad_scaled(gz, params + t.wd_off, params + ones_off, params + zeros_off, garena + t.gin_off,
                                  !(t.first_touch & 1), n_images,
                                  op.hin, op.win, op.cin, op.hout, op.wout, op.cout, op.k, op.stride, op.pad, t.dgrad_algo,
                                  gz_row, stream);
      if (rc) return rc;
      if (ord) order.after(t.gin_off, s);
    }
  }
  walk.finish();
  lanes_free.finish();
  if (!rjobs.empty()) {  // (every lane has joined the caller's stream: the slabs are complete)
    int rc = mval_wgrad_reduce_jobs(rjobs.data(), (int)rjobs.size(), mval_stream(stream0));
    if (rc) return rc;
  }
  tt_flush();
  return 0;
}

extern "C" int mval_train_backward(const mval_train_op* ops, int n_ops, int n_images, float* arena, float* garena,
                                   const float* params, int64_t ones_off, int64_t zeros_off, const float* input_nchw,
                                   float* gz, float* wsf, double* ws, float* sums, void* stream) {
  return train_backward(ops, n_ops, n_images, arena, garena, params, ones_off, zeros_off, input_nchw, gz, wsf, ws, sums, 1, 0, 0, 0, 0, stream);
}

extern "C" int mval_train_backward_lanes(const mval_train_op* ops, int n_ops, int n_images, float* arena, float* garena,
                                         const float* params, int64_t ones_off, int64_t zeros_off, const float* input_nchw,
                                         float* gz, float* wsf, double* ws, float* sums, int n_lanes, int64_t gz_floats_per_lane,
                                         int64_t wsf_floats_per_lane, int64_t ws_doubles_per_lane, int64_t sums_floats_per_lane, void* stream) {
  return train_backward(ops, n_ops, n_images, arena, garena, params, ones_off, zeros_off, input_nchw, gz, wsf, ws, sums, n_lanes, gz_floats_per_lane,
                        wsf_floats_per_lane, ws_doubles_per_lane, sums_floats_per_lane, stream);
}

// dx (+)= conv(dz [zero-dilated by the stride], flip(W)^T), stride 1, pad k-1-p: the forward
// kernels on weights packed with mode 2 (mval_pack_conv_weights), cin' = cout, cout' = cin.
extern "C" int mval_conv_dgrad(const float* dz, const float* w_packed, const float* ones, const float* zeros, float* dx,
                               int accumulate, int N, int hin, int win, int cin, int hout, int wout, int cout, int k,
                               int stride, int pad, int algo, void* stream) {
  MVAL_REQUIRE(algo != MVAL_ALGO_MFMA_H2, "mval_conv_dgrad: the fp16-split form needs dz's magnitude row (mval_conv_dgrad_scaled)");
  return mval_conv_dgrad_scaled(dz, w_packed, ones, zeros, dx, accumulate, N, hin, win, cin, hout, wout, cout, k, stride, pad,
                                algo, nullptr, stream);
}

// Data gradient of Conv2d(k3, s2, p1) on even-sized inputs as FOUR 2x2 stride-1 convs over dz, one per parity of dx, in
// one launch (the split kernels' parity form, as the inference plans run transposed convs): 16 tap-pixels per dz pixel
// instead of the 36 of the zero-dilated form (three of four staged values zero), and the fp16 split applies (stride 1).
// w_packed: mval_pack_conv_weights(pack, 4, w, ..., cout' = cin, cin' = cout, k = 4).
static void dgrad_parity_args(ConvArgs& a, int N, int hin, int win, int cin, int hout, int wout, int cout, int algo) {
  a.N = N;
  a.Hin = hout; a.Win = wout; a.Cin = cout;
  a.Hout = hout; a.Wout = wout; a.Cout = cin;  // the parity grid; dx is (hout << 1) x (wout << 1)
  a.k = 2; a.stride = 1; a.pad = 1; a.dil = 1;
  a.os_log2 = 1;
  a.G_total = (cout + 15) / 16;
  a.NS_total = (cin + 15) / 16;
  a.planes = algo == MVAL_ALGO_MFMA_H2 ? 2 : 3;
  const int pack = algo == MVAL_ALGO_MFMA_H2 ? MVAL_PACK_MFMA16_H2 : MVAL_PACK_MFMA16_BF3;
  a.par_w_stride = (int)((mval_packed_weight_floats(pack, cin, cout, 4) & ~(size_t)7) / 4);
}
extern "C" int mval_conv_dgrad_parity_supported(int N, int hin, int win, int cin, int hout, int wout, int cout, int algo) {
  if ((algo != MVAL_ALGO_MFMA_BF3 && algo != MVAL_ALGO_MFMA_H2) || N <= 0 || (hin & 1) || (win & 1) || hout * 2 != hin || wout * 2 != win ||
      (cin & 3) || (cout % 32 != 0 && cout != 48))
    return 0;
  ConvArgs a = {};
  dgrad_parity_args(a, N, hin, win, cin, hout, wout, cout, algo);
  return mval_conv_split_supported(a);
}
extern "C" int mval_conv_dgrad_parity(const float* dz, const float* w_packed, const float* ones, const float* zeros, float* dx,
                                      int accumulate, int N, int hin, int win, int cin, int hout, int wout, int cout, int algo,
                                      const uint32_t* dz_amax_row, void* stream) {
  MVAL_REQUIRE(dz && w_packed && ones && zeros && dx, "mval_conv_dgrad_parity: bad arguments");
  MVAL_REQUIRE(mval_conv_dgrad_parity_supported(N, hin, win, cin, hout, wout, cout, algo),
               "mval_conv_dgrad_parity: k3 s2 p1 on even input sizes, cin %% 4 == 0, cout %% 32 == 0 (or 48), split algos only");
  MVAL_REQUIRE(algo != MVAL_ALGO_MFMA_H2 || dz_amax_row, "mval_conv_dgrad_parity: the fp16 split needs dz's magnitude row");
  ConvArgs a = {};
  dgrad_parity_args(a, N, hin, win, cin, hout, wout, cout, algo);
  a.in = dz;
  a.w = w_packed;
  if (algo == MVAL_ALGO_MFMA_H2) {
    a.in_amax = dz_amax_row;
    a.w_unscale = w_packed + mval_packed_weight_floats(MVAL_PACK_MFMA16_H2, cin, cout, 4) - 4;
  }
  a.scale = ones;
  a.shift = zeros;
  a.out = dx;
  a.res1 = accumulate ? dx : nullptr;
  return run_conv(a, algo, mval_stream(stream), "mval_conv_dgrad_parity");
}

extern "C" int mval_conv_dgrad_scaled(const float* dz, const float* w_packed, const float* ones, const float* zeros, float* dx,
                                      int accumulate, int N, int hin, int win, int cin, int hout, int wout, int cout, int k,
                                      int stride, int pad, int algo, const uint32_t* dz_amax_row, void* stream) {
  MVAL_REQUIRE(dz && w_packed && ones && zeros && dx && N > 0, "mval_conv_dgrad: bad arguments");
  MVAL_REQUIRE(algo == MVAL_ALGO_MFMA || algo == MVAL_ALGO_MFMA_BF3 || stride == 1,
               "mval_conv_dgrad: strided data gradient needs an MFMA kernel");
  MVAL_REQUIRE(algo != MVAL_ALGO_MFMA_H2 || (stride == 1 && dz_amax_row), "mval_conv_dgrad: fp16-split form: stride 1 with dz's magnitude row");
  ConvArgs a = {};
  a.N = N;
  a.Hin = hout; a.Win = wout; a.Cin = cout;
  a.Hout = hin; a.Wout = win; a.Cout = cin;
  a.k = k; a.stride = 1; a.pad = k - 1 - pad;
  a.up = 0; a.relu = 0; a.in_nchw = 0; a.out_nchw = 0;
  a.dil = stride;
  a.th = a.tw = a.tn = a.tw_log2 = a.thw_log2 = a.tiles_x = a.tiles_y = 0;
  a.G_total = (cout + 15) / 16;
  a.NS_total = (cin + 15) / 16;
  a.precise = 0;  // measured: only the FORWARD needs bias-free accumulation (its outputs enter batch statistics)
  a.in = dz;
  a.w = w_packed;
  if (algo == MVAL_ALGO_MFMA_H2) {  // the data-gradient conv has cin' = cout, cout' = cin
    a.in_amax = dz_amax_row;
    a.w_unscale = w_packed + mval_packed_weight_floats(MVAL_PACK_MFMA16_H2, cin, cout, k) - 4;
  }
  a.scale = ones;
  a.shift = zeros;
  a.out = dx;
  a.res1 = accumulate ? dx : nullptr;  // in place: each element is read and written by one thread
  a.res2 = nullptr;
  return run_conv(a, algo, mval_stream(stream), "mval_conv_dgrad");
}
