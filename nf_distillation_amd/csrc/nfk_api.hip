// libnfk C ABI (include/nfk.h): layer-level composites that chain the nfk kernels on the
// caller's stream, plus thin op-level wrappers used by the unit parity tests.
#include "nfk_internal.h"

using namespace nfk;

namespace {
inline hipStream_t S_(void* s) { return reinterpret_cast<hipStream_t>(s); }
inline size_t align4(size_t n) { return (n + 3) & ~size_t(3); }
int splitk_of(int bf16, int M, int N, int K);
#define NFK_TRY(expr)            \
  do {                           \
    const int rc_ = (expr);      \
    if (rc_ != NFK_OK) return rc_; \
  } while (0)

struct StepBwdWs {
  size_t ga3, gz, ga2, ga1, d1slab, col2, col1, tail, wslab, x3, total;
  int S1, S3w, S2w, S1w, mb1, mb2;   // mb1 / mb2: rows RESERVED for the column-sum partials (max over the paths)
  int mb1g, mb2g;                     // rows the plain dgrad GEMMs' epilogues write
  // fused backward: conv3's / conv1's weight gradients as plain GEMMs on the im2col rows the fused kernels leave
  // (P = x3 [M][K3p] from nfk_fusedb.hip, Q = x1 [M][K1p] from nfk_fused2.hip): their split counts and row lengths
  int S3f, S1f, K3p, K1p;
  size_t ws3, ws2;   // sizes of the first two weight-gradient slab regions inside wslab
};
StepBwdWs step_bwd_layout(const nfk_step2d* s) {
  StepBwdWs w;
  const size_t M = (size_t)s->B * s->H * s->W;
  const int Ch = s->C / 2, hid = s->hidden;
  w.S1 = splitk_of(s->bf16, (int)M, Ch, 9 * hid);
  w.S3w = s->bf16 ? wgrad3_splits_bf16((int)M, s->Cout, hid, s->W) : wgrad_splits((int)M, s->Cout, 9 * hid);
  w.S2w = s->bf16 ? wgrad_splits_bf16((int)M, hid, hid) : wgrad_splits((int)M, hid, hid);
  w.S1w = s->bf16 ? wgrad_splits_bf16((int)M, hid, 9 * Ch) : wgrad_splits((int)M, hid, 9 * Ch);
  // column-sum partials of the two dgrad epilogues: one row pair per m-block of THAT launch's tiling
  w.mb2 = s->bf16 ? conv_mblocks_bf16((int)M, hid, 9 * s->Cout) : conv_mblocks((int)M, hid);   // conv3 dgrad -> col2
  w.mb1 = s->bf16 ? conv_mblocks_bf16((int)M, hid, hid) : conv_mblocks((int)M, hid);           // conv2 dgrad -> col1
  w.mb1g = w.mb1; w.mb2g = w.mb2;
  const bool fused = s->bf16 && fusedb_grid(*s) > 0;
  const int fg = fused ? actnorm_sum_parts((int)M) : 0;   // the fused backward's partial rows
  if (fg > w.mb2) w.mb2 = fg;
  if (fg > w.mb1) w.mb1 = fg;
  w.K3p = fused ? fusedb_k3p(*s) : 0;
  w.K1p = fused ? fused2_k1p(*s) : 0;
  w.S3f = fused ? wgrad_splits_bf16((int)M, w.K3p, hid) : 0;
  w.S1f = fused ? wgrad_splits_bf16((int)M, hid, w.K1p) : 0;
  size_t o = 0;
  w.ga3 = o; o += align4(M * s->Cout);
  w.gz = o; o += align4(M * s->C);
  w.ga2 = o; o += align4(M * hid);
  w.ga1 = o; o += align4(M * hid);
  w.d1slab = o; o += align4((size_t)w.S1 * M * Ch);
  w.col2 = o; o += align4((size_t)w.mb2 * 2 * hid);
  w.col1 = o; o += align4((size_t)w.mb1 * 2 * hid);
  // coupling-tail partial rows: one per sample, or one per channel-mix workgroup of the previous step (chained backward)
  w.tail = o; o += align4((size_t)(s->B > 256 ? s->B : 256) * 2 * s->Cout);
  // the three weight gradients' split slabs, each in a region of its own (reduced together by step_tail_kernel)
  const size_t e3 = (size_t)w.S3w * s->Cout * 9 * hid, e3f = (size_t)w.S3f * w.K3p * hid;
  const size_t e1 = (size_t)w.S1w * hid * 9 * Ch, e1f = (size_t)w.S1f * hid * w.K1p;
  w.ws3 = align4(e3 > e3f ? e3 : e3f);
  w.ws2 = align4((size_t)w.S2w * hid * hid);
  w.wslab = o; o += w.ws3 + w.ws2 + align4(e1 > e1f ? e1 : e1f);
  w.x3 = o; o += align4((M * w.K3p + 1) / 2);   // bf16 [M][K3p]
  w.total = o;
  return w;
}

ConvGemmArgs conv_args(const float* X, int ldx, int B, int H, int W, int Cin, int taps, const float* Wt, int N,
                       float* out, int ldo, int epi, int bf16 = 0, int xf32 = 1, int obf16 = 0) {
  ConvGemmArgs a{};
  a.X = X; a.ldx = ldx; a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.taps = taps;
  a.Wt = Wt; a.Kp = bf16 ? padded_k_bf16(taps * Cin) : padded_k(taps * Cin); a.N = N;
  a.out = out; a.ldo = ldo; a.slab_stride = (long)B * H * W * ldo; a.epi = epi;
  a.xf32 = xf32; a.obf16 = obf16;
  return a;
}
inline int run_conv(int bf16, const ConvGemmArgs& a, hipStream_t st) { return bf16 ? conv_gemm_bf16(a, st) : conv_gemm(a, st); }
int splitk_of(int bf16, int M, int N, int K) {
  return bf16 ? conv_splitk_bf16(M, N, K, EPI_RAW) : conv_splitk(M, N, K, EPI_RAW);
}
}  // namespace

// Optional auxiliary stream (nfk_aux_create): the weight-gradient GEMMs of a step's backward run on it, concurrently with
// the dgrad chain on the caller's stream (fork/join with events; capturable).  The handle travels in nfk_step2d_grads, so
// two engines on two threads / streams do not share any state.
struct nfk_aux {
  hipStream_t stream;
  hipEvent_t ev[4];       // per-call forks of the per-step paths (0..2) and the join (3)
  hipEvent_t lev[16];     // forks of the level backward: a different event for every fork of a backward pass (round 3: with ONE
  int lev_next;           // event re-recorded for each level, a hipGraph that also held a side-stream branch crashed in
                          // hipGraphLaunch -- only in a process that had captured other aux-stream graphs before)
  bool forked;   // work was sent to the auxiliary stream since the last join
};
static int g_split_fused = 1;                          // nfk_tune(NFK_TUNE_SPLIT_FUSED)
static unsigned long long* g_bwd_stamps = nullptr;   // diagnostic (nfk_fused_bwd_set_stamps)
static void fork_to_aux(nfk_aux* ax, hipStream_t main_st, int i) {
  ax->forked = true;
  (void)hipEventRecord(ax->ev[i], main_st);
  (void)hipStreamWaitEvent(ax->stream, ax->ev[i], 0);
}
static void fork_level_to_aux(nfk_aux* ax, hipStream_t main_st) {
  hipEvent_t e = ax->lev[ax->lev_next];
  ax->lev_next = (ax->lev_next + 1) % 16;
  ax->forked = true;
  (void)hipEventRecord(e, main_st);
  (void)hipStreamWaitEvent(ax->stream, e, 0);
}
// nfk_step2d_grads.bwd_mode -> 0 separate launches, 1 dual launches for small steps, 2 dual launches, 3 fused backward
static inline int bwd_mode_of(const nfk_step2d_grads* g) { return g->bwd_mode == NFK_BWD_DEFAULT ? 3 : g->bwd_mode - 1; }

extern "C" {

nfk_aux* nfk_aux_create(void* stream) {
  if (!stream) return nullptr;
  nfk_aux* ax = new nfk_aux{};
  ax->stream = reinterpret_cast<hipStream_t>(stream);
  for (auto& e : ax->ev)
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { delete ax; return nullptr; }
  for (auto& e : ax->lev)
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { delete ax; return nullptr; }
  return ax;
}
void nfk_aux_destroy(nfk_aux* ax) {
  if (!ax) return;
  for (auto& e : ax->ev) (void)hipEventDestroy(e);
  for (auto& e : ax->lev) (void)hipEventDestroy(e);
  delete ax;
}

void nfk_tune(int knob, int value) {
  switch (knob) {
    case NFK_TUNE_F32_TILES: set_tuning(value); break;
    case NFK_TUNE_BF16_BIG_TILE_MIN_K64: set_tuning_bf16(value); break;
    case NFK_TUNE_WGRAD_MAX_SPLITS: set_tuning_bf16(100 + value); break;
    case NFK_TUNE_WGRAD_TARGET_WGS: set_tuning_bf16(900 + value); break;
    case NFK_TUNE_WGRAD_WIDE_MAX_SPLITS: set_tuning_bf16(2000 + value); break;
    case NFK_TUNE_NARROW_HALO_DGRAD: set_tuning_bf16(2900 + (value & 1)); break;
    case NFK_TUNE_WGRAD_HALO: set_tuning_bf16(2910 + (value & 1)); break;
    case NFK_TUNE_WGRAD_HALO_MAX_SPLITS: set_tuning_bf16(3000 + value); break;
    case NFK_TUNE_RUN_HALF_TILES: fused2_set_half_tiles(value); fusedb_set_small_tiles(value); break;
    case NFK_TUNE_SPLIT_FUSED: g_split_fused = value & 1; break;
    case NFK_TUNE_TAB_RESIDENT: tab_set_resident(value); break;
    default: break;
  }
}
void nfk_profile_begin(void) { prof_begin(); }
void nfk_profile_end(double* ms, long long* launches, double* flops, double* bytes) {
  prof_end(ms, launches, flops, bytes);
}

const char* nfk_version(void) { return "nfk 0.1 (gfx950, f32 MFMA)"; }

int nfk_padded_k(int K) { return padded_k(K); }
int nfk_conv_splitk(int M, int N, int K) { return conv_splitk(M, N, K, EPI_RAW); }
int nfk_chanmix_blocks(int M, int C) { return chanmix_parts(M, C); }

size_t nfk_step2d_fwd_ws_floats(const nfk_step2d* s) {
  const size_t M = (size_t)s->B * s->H * s->W;
  return align4((size_t)splitk_of(s->bf16, (int)M, s->Cout, 9 * s->hidden) * M * s->Cout);
}
size_t nfk_step2d_bwd_ws_floats(const nfk_step2d* s) { return step_bwd_layout(s).total; }
size_t nfk_split2d_fwd_ws_floats(const nfk_split2d* s) {   // (the inverse always takes the GEMM path; the fused forward
  const size_t M = (size_t)s->B * s->H * s->W;             //  needs B * tiles-per-sample floats, far less)
  return align4((size_t)conv_splitk((int)M, s->C, 9 * (s->C / 2), EPI_RAW) * M * s->C);
}
size_t nfk_split2d_bwd_ws_floats(const nfk_split2d* s) {
  const size_t M = (size_t)s->B * s->H * s->W;
  const int Ch = s->C / 2;
  const size_t gemm = align4(M * s->C) + align4((size_t)conv_splitk((int)M, Ch, 9 * s->C, EPI_RAW) * M * Ch) +
                      align4((size_t)s->B * 2 * s->C) + align4((size_t)wgrad_splits((int)M, s->C, 9 * Ch) * s->C * 9 * Ch);
  const size_t nt = (size_t)split2d_fused_tiles(*s);
  const size_t fused = align4(nt * 2 * s->C) + align4(nt * s->C * 9 * Ch);
  return gemm > fused ? gemm : fused;
}
int nfk_split2d_fwd_tiles(const nfk_split2d* s) { return nfk_split2d_fused_ok(s) ? split2d_fused_tps(*s) : 0; }
int nfk_split2d_fused_ok(const nfk_split2d* s) { return (s->w && g_split_fused && split2d_fused_tiles(*s) > 0) ? 1 : 0; }

int nfk_prep(const nfk_prep_desc* descs, int nsteps, int want_inverse, void* stream) {
  return launch_prep(descs, nsteps, want_inverse, S_(stream));
}
int nfk_prep_repack(const nfk_prep_desc* descs, int nsteps, int want_inverse, const nfk_repack_desc* f32_descs, int n_f32,
                    const nfk_repack_desc* bf16_descs, int n_bf16, void* stream) {
  return launch_prep_repack(descs, nsteps, want_inverse, f32_descs, n_f32, bf16_descs, n_bf16, S_(stream));
}
int nfk_prep_bwd(const nfk_prep_desc* descs, const nfk_prep_grad* grads, int nsteps, const float* g_logdet, int B,
                 int any_dense, void* stream) {
  return launch_prep_bwd(descs, grads, nsteps, g_logdet, B, any_dense, S_(stream));
}
int nfk_prep_inv_bwd(const nfk_prep_desc* descs, const nfk_prep_grad* grads, int nsteps, const float* g_logdet, int B,
                     void* stream) {
  return launch_prep_inv_bwd(descs, grads, nsteps, g_logdet, B, S_(stream));
}
int nfk_repack(const float* w, int Cout, int Cin, int taps, float* wt, float* wd, void* stream) {
  if (taps != 1 && taps != 9) return NFK_ERR_ARG;
  return repack_weights(w, Cout, Cin, taps, wt, wd, S_(stream));
}

int nfk_repack_batch(const nfk_repack_desc* descs, int n, void* stream) { return repack_batch(descs, n, S_(stream)); }
int nfk_repack_batch_bf16(const nfk_repack_desc* descs, int n, void* stream) {
  return repack_batch_bf16(descs, n, S_(stream));
}
int nfk_padded_k_bf16(int K) { return padded_k_bf16(K); }

int nfk_chanmix(const float* x, int ldx, const float* Wm, const float* cv, float* z, int ldz, int M, int C,
                void* stream) {
  ChanMixArgs a{};
  a.x = x; a.ldx = ldx; a.Wm = Wm; a.cv = cv; a.z = z; a.ldz = ldz; a.M = M; a.C = C;
  return launch_chanmix(a, S_(stream));
}

int nfk_conv(const float* x, int ldx, int B, int H, int W, int Cin, int taps, const float* wt, int N, float* out,
             int ldo, int epi, const float* e0, const float* e1, void* stream) {
  if (epi != EPI_RAW && epi != EPI_AN_RELU) return NFK_ERR_ARG;
  ConvGemmArgs a = conv_args(x, ldx, B, H, W, Cin, taps, wt, N, out, ldo, epi);
  a.e0 = e0; a.e1 = e1;
  return conv_gemm(a, S_(stream));
}

int nfk_coupling_fwd(const float* a3, int S, const float* bias3, const float* es3, const float* z, int ldz,
                     float* zout, int ldo, float* hsave, const float* ld_in, const float* dld, float* ld_out,
                     int B, int HW, int C, int Cout, int additive, void* stream) {
  CouplingArgs a{};
  a.a3 = a3; a.S = S; a.slab_stride = (long)B * HW * Cout; a.bias3 = bias3; a.es3 = es3;
  a.z = z; a.ldz = ldz; a.zout = zout; a.ldo = ldo; a.hsave = hsave;
  a.ld_in = ld_in; a.dld = dld; a.ld_out = ld_out; a.has_ld = ld_out != nullptr;
  a.HW = HW; a.C = C; a.Cout = Cout; a.additive = additive;
  return launch_coupling_fwd(a, B, S_(stream));
}
int nfk_coupling_inv(const float* a3, int S, const float* bias3, const float* es3, const float* y, int ldy,
                     float* z, int ldz, const float* ld_in, const float* dld, float* ld_out, int B, int HW, int C,
                     int Cout, int additive, void* stream) {
  CouplingArgs a{};
  a.a3 = a3; a.S = S; a.slab_stride = (long)B * HW * Cout; a.bias3 = bias3; a.es3 = es3;
  a.z = y; a.ldz = ldy; a.zout = z; a.ldo = ldz;
  a.ld_in = ld_in; a.dld = dld; a.ld_out = ld_out; a.has_ld = ld_out != nullptr;
  a.HW = HW; a.C = C; a.Cout = Cout; a.additive = additive;
  return launch_coupling_inv(a, B, S_(stream));
}

int nfk_squeeze(float* x, int ldx, const float* noise, float* out, int B, int C, int H, int W, int from_nchw,
                void* stream) {
  if (from_nchw) return launch_squeeze_nchw(x, noise, out, B, C, H, W, S_(stream));
  if (noise) return NFK_ERR_ARG;
  return launch_squeeze_nhwc(x, ldx, out, B, C, H, W, S_(stream));
}
int nfk_unsqueeze(const float* y, float* out, int B, int C, int H, int W, int to_nchw, void* stream) {
  return launch_unsqueeze(y, nullptr, out, B, C, H, W, to_nchw, S_(stream));
}
int nfk_unsqueeze_add(const float* y, const float* y2, float* out, int B, int C, int H, int W, int to_nchw, void* stream) {
  return launch_unsqueeze(y, y2, out, B, C, H, W, to_nchw, S_(stream));
}

int nfk_prior_fwd(const float* z, int ldz, const float* prior, long prior_bstride, const float* ld,
                  float* objective, float* bpd, int B, int HW, int C, float bpd_scale, void* stream) {
  PriorArgs a{};
  a.z = z; a.ldz = ldz; a.ph = prior; a.ph_bstride = prior_bstride; a.ld = ld; a.objective = objective;
  a.bpd = bpd; a.HW = HW; a.C = C; a.bpd_scale = bpd_scale;
  return launch_prior_fwd(a, B, S_(stream));
}
int nfk_prior_bwd(const float* z, int ldz, const float* prior, long prior_bstride, const float* g_bpd, float* g_z,
                  int ldgz, float* g_ld, float* g_prior, int B, int HW, int C, float bpd_scale, void* stream) {
  PriorBwdArgs a{};
  a.z = z; a.ldz = ldz; a.ph = prior; a.ph_bstride = prior_bstride; a.gbpd = g_bpd; a.gz = g_z; a.ldgz = ldgz;
  a.gld = g_ld; a.gph = g_prior; a.HW = HW; a.C = C; a.bpd_scale = bpd_scale;
  return launch_prior_bwd(a, B, S_(stream));
}

// ------------------------------------------------------------------------------------------
// FlowStep composites
// ------------------------------------------------------------------------------------------
static int coupling_net_fwd(const nfk_step2d* s, const float* zin, int ldz, float* h1, float* h2, float* ws,
                            hipStream_t st) {
  const int Ch = s->C / 2, hid = s->hidden;
  const int bf = s->bf16;  // bf16 mode: h1/h2 are bf16 in HBM, weights bf16, z stays fp32
  ConvGemmArgs c1 = conv_args(zin, ldz, s->B, s->H, s->W, Ch, 9, s->wt1, hid, h1, hid, EPI_AN_RELU, bf, 1, bf);
  c1.e0 = s->b1; c1.e1 = s->es1;
  NFK_TRY(run_conv(bf, c1, st));
  ConvGemmArgs c2 = conv_args(h1, hid, s->B, s->H, s->W, hid, 1, s->wt2, hid, h2, hid, EPI_AN_RELU, bf, !bf, bf);
  c2.e0 = s->b2; c2.e1 = s->es2;
  NFK_TRY(run_conv(bf, c2, st));
  ConvGemmArgs c3 = conv_args(h2, hid, s->B, s->H, s->W, hid, 9, s->wt3, s->Cout, ws, s->Cout, EPI_RAW, bf, !bf, 0);
  return run_conv(bf, c3, st);
}

int nfk_step2d_fwd(const nfk_step2d* s, const float* x, int ldx, const float* ld_in, float* z, float* h1,
                   float* h2, float* hsave, float* ws, float* zout, float* ld_out, void* stream) {
  hipStream_t st = S_(stream);
  const int M = s->B * s->H * s->W, HW = s->H * s->W;
  if (s->C < 2 || s->hidden < 1) return NFK_ERR_ARG;
  ChanMixArgs m{};
  m.x = x; m.ldx = ldx; m.Wm = s->Wf; m.cv = s->cf; m.z = z; m.ldz = s->C; m.M = M; m.C = s->C;
  NFK_TRY(launch_chanmix(m, st));
  NFK_TRY(coupling_net_fwd(s, z, s->C, h1, h2, ws, st));
  const int S = splitk_of(s->bf16, M, s->Cout, 9 * s->hidden);
  return nfk_coupling_fwd(ws, S, s->bias3, s->es3, z, s->C, zout, s->C, hsave, ld_in, s->dld, ld_out, s->B, HW,
                          s->C, s->Cout, s->additive, stream);
}

int nfk_step2d_inv(const nfk_step2d* s, const float* y, int ldy, const float* ld_in, float* z, float* h1,
                   float* h2, float* hsave, float* ws, float* xout, float* ld_out, void* stream) {
  hipStream_t st = S_(stream);
  const int M = s->B * s->H * s->W, HW = s->H * s->W;
  if (!s->Wi || !s->ci) return NFK_ERR_ARG;
  NFK_TRY(coupling_net_fwd(s, y, ldy, h1, h2, ws, st));
  const int S = splitk_of(s->bf16, M, s->Cout, 9 * s->hidden);
  CouplingArgs a{};
  a.a3 = ws; a.S = S; a.slab_stride = (long)M * s->Cout; a.bias3 = s->bias3; a.es3 = s->es3;
  a.z = y; a.ldz = ldy; a.zout = z; a.ldo = s->C; a.hsave = hsave;
  a.ld_in = ld_in; a.dld = s->dld; a.ld_out = ld_out; a.has_ld = ld_out != nullptr;
  a.HW = HW; a.C = s->C; a.Cout = s->Cout; a.additive = s->additive;
  NFK_TRY(launch_coupling_inv(a, s->B, st));
  ChanMixArgs m{};
  m.x = z; m.ldx = s->C; m.Wm = s->Wi; m.cv = s->ci; m.z = xout; m.ldz = s->C; m.M = M; m.C = s->C;
  return launch_chanmix(m, st);
}

// Backward of nfk_step2d_inv (see include/nfk.h).  The inverse step is  y -> [coupling^-1 with NN(y1)] -> z -> z Wi + ci -> x;
// the coupling net is evaluated on y1 exactly as the forward evaluates it on z1, so its backward is the forward step's:
// the same dgrad / weight-gradient GEMMs on the activations nfk_step2d_inv kept, in their plain (one launch per GEMM) form.
int nfk_step2d_inv_bwd(const nfk_step2d* s, const nfk_step2d_grads* g, const float* y, int ldy, const float* z,
                       const float* h1, const float* h2, const float* hsave, const float* g_xout, const float* g_ld,
                       float* g_y, float* ws, void* stream) {
  hipStream_t st = S_(stream);
  const int B = s->B, H = s->H, W = s->W, C = s->C, Ch = C / 2, hid = s->hidden, Cout = s->Cout;
  const int M = B * H * W, HW = H * W, bf = s->bf16;
  if (!s->Wi || !hsave || !g->mix_part) return NFK_ERR_ARG;
  const StepBwdWs L = step_bwd_layout(s);
  float *ga3 = ws + L.ga3, *gz = ws + L.gz, *ga2 = ws + L.ga2, *ga1 = ws + L.ga1, *d1 = ws + L.d1slab;
  float *col2 = ws + L.col2, *col1 = ws + L.col1, *tail = ws + L.tail, *wslab = ws + L.wslab;
  // inverse channel mix: g_z = Wi^T g_x, partial (g_Wi, g_ci) rows
  ChanMixArgs m{};
  m.x = g_xout; m.ldx = C; m.Wm = s->Wi; m.cv = nullptr; m.z = gz; m.ldz = C; m.M = M; m.C = C; m.trans = 1;
  m.xin = z; m.ldxin = C; m.part = g->mix_part;
  NFK_TRY(launch_chanmix(m, st));
  CouplingBwdArgs cb{};
  cb.gout = gz; cb.ldg = C; cb.gld = g_ld; cb.z = y; cb.ldz = ldy; cb.h = hsave; cb.es3 = s->es3; cb.ga3 = ga3;
  cb.gz = g_y; cb.ldgz = C; cb.part = tail; cb.HW = HW; cb.C = C; cb.Cout = Cout; cb.additive = s->additive;
  NFK_TRY(launch_coupling_inv_bwd(cb, B, st));
  auto conv_bwd = [&](const ConvGemmArgs& dg, const WgradArgs& wg, int Sw, int Co, int Ci, int taps, float* gw) -> int {
    NFK_TRY(bf ? wgrad_gemm_bf16(wg, Sw, st) : wgrad_gemm(wg, Sw, st));
    NFK_TRY(wgrad_finalize(wg.out, Sw, wg.slab_stride, Co, Ci, taps, gw, st));
    return run_conv(bf, dg, st);
  };
  WgradArgs w3{};
  w3.P = ga3; w3.ldp = Cout; w3.NI = Cout; w3.Q = h2; w3.ldq = hid; w3.Cq = hid; w3.taps = 9;
  w3.B = B; w3.H = H; w3.W = W; w3.out = wslab; w3.slab_stride = (long)Cout * 9 * hid; w3.qbf16 = bf;
  ConvGemmArgs d3 = conv_args(ga3, Cout, B, H, W, Cout, 9, s->wd3, hid, ga2, hid, EPI_DGRAD, bf, 1, bf);
  d3.e1 = s->es2; d3.hprev = h2; d3.ldh = hid; d3.colsum = col2;
  NFK_TRY(conv_bwd(d3, w3, L.S3w, Cout, hid, 9, g->g_w3));
  WgradArgs w2{};
  w2.P = ga2; w2.ldp = hid; w2.NI = hid; w2.Q = h1; w2.ldq = hid; w2.Cq = hid; w2.taps = 1;
  w2.B = B; w2.H = H; w2.W = W; w2.out = wslab + L.ws3; w2.slab_stride = (long)hid * hid; w2.pbf16 = bf; w2.qbf16 = bf;
  ConvGemmArgs d2 = conv_args(ga2, hid, B, H, W, hid, 1, s->wd2, hid, ga1, hid, EPI_DGRAD, bf, !bf, bf);
  d2.e1 = s->es1; d2.hprev = h1; d2.ldh = hid; d2.colsum = col1;
  NFK_TRY(conv_bwd(d2, w2, L.S2w, hid, hid, 1, g->g_w2));
  WgradArgs w1{};
  w1.P = ga1; w1.ldp = hid; w1.NI = hid; w1.Q = y; w1.ldq = ldy; w1.Cq = Ch; w1.taps = 9;
  w1.B = B; w1.H = H; w1.W = W; w1.out = wslab + L.ws3 + L.ws2; w1.slab_stride = (long)hid * 9 * Ch; w1.pbf16 = bf;
  ConvGemmArgs d1a = conv_args(ga1, hid, B, H, W, hid, 9, s->wd1, Ch, d1, Ch, EPI_RAW, bf, !bf, 0);
  NFK_TRY(conv_bwd(d1a, w1, L.S1w, hid, Ch, 9, g->g_w1));
  nfk_small_grad_args sg{};
  sg.col1 = col1; sg.col2 = col2; sg.tail = tail; sg.mb1 = L.mb1g; sg.mb2 = L.mb2g; sg.B = B;
  sg.n1 = hid; sg.n2 = hid; sg.n3 = Cout;
  sg.g_l1 = g->g_l1; sg.g_b1 = g->g_b1; sg.g_l2 = g->g_l2; sg.g_b2 = g->g_b2; sg.g_bias3 = g->g_bias3; sg.g_l3 = g->g_l3;
  NFK_TRY(launch_small_grads(sg, st));
  // dL/dy1 = dL/dz1 (y1 passes through the coupling) + the coupling net's input gradient
  return launch_add_slabs(gz, C, nullptr, 0, d1, L.S1, (long)M * Ch, g_y, C, M, Ch, st);
}

// What a step's backward still owes when its weight-gradient launch has been merged with the channel-mix launch
// (wgrad_tail_bf16): the reduction of its split slabs and of the ActNorm layers' partial rows -- done by the NEXT step's
// merged launch, or by nfk_step2d_pending_flush after the last one.
struct PendingTail {
  nfk_small_grad_args sg;   // col1 / col2 segments only (n3 = 0)
  StepTailArgs f;
  // where the owed slabs / partial rows sit: the workspace of the step that left them and its layout's geometry.  A later
  // call may only carry them along in its merged launch if nothing in that launch writes the same workspace under ANOTHER
  // layout (the chained coupling backward of `next` writes next->ws + {ga3, gz, tail} of the NEW geometry: at a level
  // boundary those offsets fall inside the old level's slab regions)
  const float* ws;
  int B, H, W, C, hidden, Cout;
};
static_assert(sizeof(PendingTail) <= sizeof(((nfk_step2d_pending*)nullptr)->blob), "nfk_step2d_pending.blob too small");
static PendingTail* pending_of(nfk_step2d_pending* p) { return reinterpret_cast<PendingTail*>(p->blob); }
static int pending_flush(nfk_step2d_pending* p, hipStream_t st) {
  if (!p || !p->valid) return NFK_OK;
  p->valid = 0;
  return launch_step_tail(pending_of(p)->sg, pending_of(p)->f, st);
}

// ga / wslab_ext: see nfk_step2d_bwd_deferred (both NULL = everything inside ws, auxiliary stream joined per call)
static int step2d_bwd_impl(const nfk_step2d* s, const nfk_step2d_grads* g, const float* x, int ldx, const float* z,
                           const float* h1, const float* h2, const float* hsave, const float* g_out,
                           const float* g_ext, const float* g_ld, float* g_x, float* ws, float* ga, float* wslab_ext,
                           void* stream, int coupling_rows = 0, const nfk_step2d_next* next = nullptr,
                           int* next_rows = nullptr, nfk_step2d_pending* pend = nullptr) {
  hipStream_t st = S_(stream);
  const int B = s->B, H = s->H, W = s->W, C = s->C, Ch = C / 2, hid = s->hidden, Cout = s->Cout;
  const int M = B * H * W, HW = H * W, bf = s->bf16;
  const StepBwdWs L = step_bwd_layout(s);
  float *ga3 = ws + L.ga3, *gz = ws + L.gz, *ga2 = ws + L.ga2, *ga1 = ws + L.ga1, *d1 = ws + L.d1slab;
  float *col2 = ws + L.col2, *col1 = ws + L.col1, *tail = ws + L.tail, *wslab = ws + L.wslab;
  nfk_aux* const ax = g->aux;
  const int bwd_mode = bwd_mode_of(g);
  const bool deferred = ga != nullptr && wslab_ext != nullptr && ax != nullptr && ax->stream != st;
  float* x3 = ws + L.x3;
  if (deferred) {   // the tensors the weight-gradient GEMMs read live in this step's own buffer
    ga3 = ga;
    ga2 = ga3 + align4((size_t)M * Cout);
    ga1 = ga2 + align4((size_t)M * hid);
    x3 = ga1 + align4((size_t)M * hid);
    wslab = wslab_ext;
  }

  CouplingBwdArgs cb{};
  cb.gout = g_out; cb.ldg = C; cb.gext = g_ext; cb.ldge = C; cb.gld = g_ld; cb.z = z; cb.ldz = C; cb.h = hsave;
  cb.es3 = s->es3; cb.ga3 = ga3; cb.gz = gz; cb.ldgz = C; cb.part = tail; cb.HW = HW; cb.C = C; cb.Cout = Cout;
  cb.additive = s->additive;
  const int cpl_tiles = coupling_bwd_tiles(B, HW);   // stand-alone coupling backward: B * cpl_tiles workgroups / partial rows
  cb.tiles = cpl_tiles;
  if (coupling_rows <= 0) NFK_TRY(launch_coupling_bwd(cb, B, st));   // else: done by the previous call of the chain
  // Each conv's weight gradient (slabs -> finalize) and dgrad read the same output gradient and are independent.
  // bf16 "dual" mode runs the two GEMMs as ONE launch on the caller's stream; otherwise the weight gradient goes to
  // the auxiliary stream (if one is set), forked here and joined at the end of the call / by nfk_aux_join.
  const bool dual = bf && (bwd_mode >= 2 || (bwd_mode == 1 && M <= 4096));
  // fused backward: needs the ReLU masks the fused forward leaves (g->relu_masks) and a supported geometry
  const int fgrid = (bf && bwd_mode == 3 && g->relu_masks) ? fusedb_grid(*s) : 0;
  hipStream_t ws_st = st;  // stream of the wgrad GEMMs
  const bool use_aux = !dual && ax && ax->stream != st;
  if (use_aux) ws_st = ax->stream;
  int fork_i = 0, conv_i = 0;
  StepTailArgs tail_f{};   // slab reductions deferred to the one launch at the end of the step (dual mode)
  auto conv_bwd = [&](const ConvGemmArgs& dg, const WgradArgs& wg, int Sw, int Co, int Ci, int taps, float* gw) -> int {
    const int k = conv_i++;
    if (dual) {
      const int rc = dual_dgrad_wgrad_bf16(dg, wg, Sw, st);
      if (rc == NFK_OK) {
        tail_f.slabs[k] = wg.out; tail_f.gradW[k] = gw; tail_f.stride[k] = wg.slab_stride;
        tail_f.S[k] = Sw; tail_f.Cout[k] = Co; tail_f.Cin[k] = Ci; tail_f.taps[k] = taps;
        return NFK_OK;
      }
      if (rc != NFK_ERR_ARG) return rc;   // NFK_ERR_ARG: this pair of tile shapes is not instantiated
    }
    if (use_aux) fork_to_aux(ax, st, fork_i++);
    NFK_TRY(bf ? wgrad_gemm_bf16(wg, Sw, ws_st) : wgrad_gemm(wg, Sw, ws_st));
    NFK_TRY(wgrad_finalize(wg.out, Sw, wg.slab_stride, Co, Ci, taps, gw, ws_st));
    return run_conv(bf, dg, st);
  };

  int mb1 = L.mb1g, mb2 = L.mb2g;
  // conv3 (block.4)
  WgradArgs w3{};
  w3.P = ga3; w3.ldp = Cout; w3.NI = Cout; w3.Q = h2; w3.ldq = hid; w3.Cq = hid; w3.taps = 9;
  w3.B = B; w3.H = H; w3.W = W; w3.out = wslab; w3.slab_stride = (long)Cout * 9 * hid;
  w3.qbf16 = bf;
  ConvGemmArgs d3 = conv_args(ga3, Cout, B, H, W, Cout, 9, s->wd3, hid, ga2, hid, EPI_DGRAD, bf, 1, bf);
  d3.e1 = s->es2; d3.hprev = h2; d3.ldh = hid; d3.colsum = col2;
  if (!fgrid) NFK_TRY(conv_bwd(d3, w3, L.S3w, Cout, hid, 9, g->g_w3));

  // conv2 (block.2)
  WgradArgs w2{};
  w2.P = ga2; w2.ldp = hid; w2.NI = hid; w2.Q = h1; w2.ldq = hid; w2.Cq = hid; w2.taps = 1;
  w2.B = B; w2.H = H; w2.W = W; w2.out = wslab + L.ws3; w2.slab_stride = (long)hid * hid;
  w2.pbf16 = bf; w2.qbf16 = bf;
  ConvGemmArgs d2 = conv_args(ga2, hid, B, H, W, hid, 1, s->wd2, hid, ga1, hid, EPI_DGRAD, bf, !bf, bf);
  d2.e1 = s->es1; d2.hprev = h1; d2.ldh = hid; d2.colsum = col1;
  if (!fgrid) NFK_TRY(conv_bwd(d2, w2, L.S2w, hid, hid, 1, g->g_w2));

  // conv1 (block.0)
  WgradArgs w1{};
  w1.P = ga1; w1.ldp = hid; w1.NI = hid; w1.Q = z; w1.ldq = C; w1.Cq = Ch; w1.taps = 9;
  w1.B = B; w1.H = H; w1.W = W; w1.out = wslab + L.ws3 + L.ws2; w1.slab_stride = (long)hid * 9 * Ch;
  w1.pbf16 = bf;
  ConvGemmArgs d1a = conv_args(ga1, hid, B, H, W, hid, 9, s->wd1, Ch, d1, Ch, EPI_RAW, bf, !bf, 0);
  if (!fgrid) NFK_TRY(conv_bwd(d1a, w1, L.S1w, hid, Ch, 9, g->g_w1));
  bool merged = false;
  ActnormSumArgs an_m{};
  if (fgrid) {
    // fused backward: the whole dgrad chain in one launch (g_a2 / g_a1 stay in LDS between the convolutions and are
    // written once for the weight gradients), then the three weight-gradient GEMMs in one launch
    FusedBwdCall fc{};
    fc.s = *s; fc.ga3 = ga3; fc.masks = reinterpret_cast<const unsigned char*>(g->relu_masks);
    fc.ga2 = ga2; fc.ga1 = ga1; fc.d1 = d1; fc.stamps = g_bwd_stamps;
    fc.x3save = x3;
    NFK_TRY(fusedb_launch(fc, st));
    // Deferred mode (an auxiliary stream and per-step buffers: nfk_step2d_bwd_deferred / _chain): the weight-gradient
    // launch, the reduction of its slabs and the ActNorm gradients leave the caller's stream -- nothing on the chain
    // dgrad(s) -> channel mix(s) + coupling backward(s-1) -> dgrad(s-1) ... reads them -- and form a chain of their own
    // on the auxiliary stream, forked here once per step and joined once per backward (nfk_aux_join).  Their per-channel
    // partial sums (col1 / col2) then live behind the slabs in `wslab`, which only that stream touches.
    hipStream_t wst = st;
    if (deferred) {
      fork_to_aux(ax, st, 0);
      wst = ax->stream;
      col2 = wslab + (L.x3 - L.wslab);
      col1 = col2 + align4((size_t)L.mb2 * 2 * hid);
    }
    // conv3 and conv1 as plain pixel-major GEMMs: the 3x3 taps sit in the im2col rows the two fused kernels built anyway
    // (dgrad3's of g_a3 above, conv1's of z1 behind the ReLU masks of the forward), not in nine shifted passes / a
    // per-element gather:  gW3[co][ci][tap] = sum_m x3[m][(8 - tap) Cout + co] h2[m][ci],
    //                      gW1[co][ci][tap] = sum_m g_a1[m][co] x1[m][tap C/2 + ci]
    w3.P = x3; w3.ldp = L.K3p; w3.NI = L.K3p; w3.pbf16 = 1; w3.taps = 1; w3.slab_stride = (long)L.K3p * hid;
    w1.Q = reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(g->relu_masks) + (size_t)2 * M * (hid / 8));
    w1.ldq = L.K1p; w1.Cq = L.K1p; w1.taps = 1; w1.qbf16 = 1; w1.slab_stride = (long)hid * L.K1p;
    ActnormSumArgs an{};
    an.ga2 = ga2; an.h2 = h2; an.ga1 = ga1; an.h1 = h1; an.es2 = s->es2; an.es1 = s->es1; an.col2 = col2; an.col1 = col1;
    an.M = M; an.hid = hid; an.nparts = actnorm_sum_parts(M);
    mb1 = mb2 = an.nparts;
    merged = pend != nullptr && !deferred;   // the GEMMs share ONE launch with the channel mix below (wgrad_tail_bf16)
    an_m = an;
    if (!merged)
      NFK_TRY(triple_wgrad_bf16(w3, L.S3f, w2, L.S2w, w1, L.S1f, an, wst, (double)hid * (9.0 * Cout + hid + 9.0 * Ch)));
    const WgradArgs* ws3[3] = {&w3, &w2, &w1};
    const int Sw[3] = {L.S3f, L.S2w, L.S1f}, Co[3] = {Cout, hid, hid}, Ci[3] = {hid, hid, Ch}, tp[3] = {9, 1, 9};
    const int md[3] = {1, 0, 0}, lj[3] = {hid, 0, L.K1p};
    float* gw[3] = {g->g_w3, g->g_w2, g->g_w1};
    for (int k = 0; k < 3; ++k) {
      tail_f.slabs[k] = ws3[k]->out; tail_f.gradW[k] = gw[k]; tail_f.stride[k] = ws3[k]->slab_stride;
      tail_f.S[k] = Sw[k]; tail_f.Cout[k] = Co[k]; tail_f.Cin[k] = Ci[k]; tail_f.taps[k] = tp[k];
      tail_f.mode[k] = md[k]; tail_f.ldj[k] = lj[k];
    }
    if (deferred) {   // slab reductions + the two ActNorm layers' gradients: behind the GEMMs on the auxiliary stream
      nfk_small_grad_args sa{};
      sa.col1 = col1; sa.col2 = col2; sa.mb1 = mb1; sa.mb2 = mb2; sa.n1 = hid; sa.n2 = hid; sa.n3 = 0;
      sa.g_l1 = g->g_l1; sa.g_b1 = g->g_b1; sa.g_l2 = g->g_l2; sa.g_b2 = g->g_b2;
      NFK_TRY(launch_step_tail(sa, tail_f, wst));
      tail_f = StepTailArgs{};   // nothing of it is left for the caller's stream
    }
  }
  const bool fused_deferred = fgrid && deferred;

  nfk_small_grad_args sg{};
  sg.col1 = col1; sg.col2 = col2; sg.tail = tail; sg.mb1 = mb1; sg.mb2 = mb2;
  sg.B = coupling_rows > 0 ? coupling_rows : B * cpl_tiles;
  sg.n1 = fused_deferred ? 0 : hid; sg.n2 = fused_deferred ? 0 : hid; sg.n3 = Cout;
  sg.g_l1 = g->g_l1; sg.g_b1 = g->g_b1; sg.g_l2 = g->g_l2; sg.g_b2 = g->g_b2; sg.g_bias3 = g->g_bias3;
  sg.g_l3 = g->g_l3;
  // (fused + deferred: the caller's stream still runs the one launch of coupling-tail gradients + channel mix + the next
  //  step's coupling backward, only without slab work)
  const bool have_tail = fused_deferred || (tail_f.S[0] | tail_f.S[1] | tail_f.S[2]) != 0;
  if (pend && pend->valid) {
    // this call cannot take the previous step's tail along (no merged launch), or must not: the owed step belongs to
    // another level, so its slabs sit under another layout in a workspace this launch's chained writes may touch
    const PendingTail* pt = pending_of(pend);
    const bool same_geom = pt->B == B && pt->H == H && pt->W == W && pt->C == C && pt->hidden == hid && pt->Cout == Cout;
    if (!merged || !same_geom) NFK_TRY(pending_flush(pend, st));
  }

  // actnorm + permutation backward: g_x = Wf^T g_z, partial (gWf, gcf) for nfk_prep_bwd
  ChanMixArgs m{};
  m.x = gz; m.ldx = C; m.Wm = s->Wf; m.cv = nullptr; m.z = g_x; m.ldz = C; m.M = M; m.C = C; m.trans = 1;
  // conv1's dgrad leaves L.S1 split-K slabs -- or one, when the dual launch ran it through the halo-tile kernel
  const int S1_used = (fgrid || (dual && narrow_halo_dgrad_active(bf, M, H, W, Ch, hid))) ? 1 : L.S1;
  m.ga = g_out; m.lda = C; m.gext = g_ext; m.ldge = C; m.slab = d1; m.S = S1_used; m.slab_stride = (long)M * Ch;
  m.Ch = Ch; m.xin = x; m.ldxin = ldx; m.part = g->mix_part;
  // dual mode: the step tail (slab reductions + per-channel gradients) and this channel mix share one launch
  if (next_rows) *next_rows = 0;
  CouplingBwdArgs nb{};
  bool chained = false;
  if (next && next->s && have_tail) {
    const nfk_step2d* n = next->s;
    if (!next_rows || n->B != B || n->H != H || n->W != W || n->C != C || !next->ws || next->ws == ws || (deferred && next->ga == ga))
      return NFK_ERR_ARG;
    const StepBwdWs Ln = step_bwd_layout(n);
    nb.gext = next->g_ext; nb.ldge = C; nb.gld = g_ld; nb.z = next->z; nb.ldz = C; nb.h = next->hsave;
    nb.es3 = n->es3; nb.ga3 = (deferred && next->ga) ? next->ga : next->ws + Ln.ga3; nb.gz = next->ws + Ln.gz;
    nb.ldgz = C; nb.part = next->ws + Ln.tail; nb.HW = HW; nb.C = C; nb.Cout = n->Cout; nb.additive = n->additive;
    chained = true;
  }
  if (merged) {
    // ONE launch: this step's GEMMs || channel mix + next step's coupling backward + this step's coupling-tail gradients
    // || the PREVIOUS step's slab / partial-row reductions (its GEMMs finished two launches ago); this step's own
    // reductions become the next call's (or nfk_step2d_pending_flush's)
    nfk_small_grad_args sx = sg;       // coupling-tail segment (n3): this step's; col1 / col2 segments: the previous step's
    StepTailArgs fx{};
    sx.n1 = sx.n2 = 0;
    if (pend->valid) {
      const PendingTail* pt = pending_of(pend);
      sx.col1 = pt->sg.col1; sx.col2 = pt->sg.col2; sx.mb1 = pt->sg.mb1; sx.mb2 = pt->sg.mb2; sx.n1 = pt->sg.n1; sx.n2 = pt->sg.n2;
      sx.g_l1 = pt->sg.g_l1; sx.g_b1 = pt->sg.g_b1; sx.g_l2 = pt->sg.g_l2; sx.g_b2 = pt->sg.g_b2;
      fx = pt->f;
    }
    const bool want_chain = chained;
    int rc = wgrad_tail_bf16(w3, L.S3f, w2, L.S2w, w1, L.S1f, an_m, sx, fx, m, chained ? &nb : nullptr, st,
                             (double)hid * (9.0 * Cout + hid + 9.0 * Ch));
    if (rc == NFK_ERR_ARG && chained) {   // the coupling backward does not fit beside the channel mix: unchained
      chained = false;
      rc = wgrad_tail_bf16(w3, L.S3f, w2, L.S2w, w1, L.S1f, an_m, sx, fx, m, nullptr, st, (double)hid * (9.0 * Cout + hid + 9.0 * Ch));
    }
    if (rc == NFK_ERR_ARG) {
      // the merged launch does not exist for this geometry (channel mix too wide for its LDS, operand rows not a multiple
      // of 8): the three GEMMs as their own launch, what the previous step owes now, then the unmerged tail below
      NFK_TRY(triple_wgrad_bf16(w3, L.S3f, w2, L.S2w, w1, L.S1f, an_m, st, (double)hid * (9.0 * Cout + hid + 9.0 * Ch)));
      NFK_TRY(pending_flush(pend, st));
      merged = false;
      chained = want_chain;
    } else if (rc != NFK_OK) {
      return rc;
    } else {
      if (chained && next_rows) *next_rows = chanmix_parts(M, C);
      PendingTail* pt = pending_of(pend);
      pt->sg = sg; pt->sg.n3 = 0; pt->sg.tail = nullptr;
      pt->f = tail_f;
      pt->ws = ws; pt->B = B; pt->H = H; pt->W = W; pt->C = C; pt->hidden = hid; pt->Cout = Cout;
      pend->valid = 1;
    }
  }
  if (!merged && have_tail) {
    int rc = launch_tail_chanmix(sg, tail_f, m, chained ? &nb : nullptr, st);
    if (rc == NFK_OK && chained && next_rows) *next_rows = chanmix_parts(M, C);
    if (rc == NFK_ERR_ARG && chained) rc = launch_tail_chanmix(sg, tail_f, m, nullptr, st);   // does not fit: unchained
    if (rc == NFK_ERR_ARG) { NFK_TRY(launch_step_tail(sg, tail_f, st)); NFK_TRY(launch_chanmix(m, st)); }
    else if (rc != NFK_OK) return rc;
  } else if (!merged) {
    NFK_TRY(launch_small_grads(sg, st));
    NFK_TRY(launch_chanmix(m, st));
  }
  if (ws_st != st && !deferred) {  // join: this step's workspace and weight grads are complete when the call's stream moves on
    (void)hipEventRecord(ax->ev[3], ws_st);
    (void)hipStreamWaitEvent(st, ax->ev[3], 0);
    ax->forked = false;
  }
  return NFK_OK;
}

int nfk_step2d_bwd(const nfk_step2d* s, const nfk_step2d_grads* g, const float* x, int ldx, const float* z,
                   const float* h1, const float* h2, const float* hsave, const float* g_out, const float* g_ext,
                   const float* g_ld, float* g_x, float* ws, void* stream) {
  return step2d_bwd_impl(s, g, x, ldx, z, h1, h2, hsave, g_out, g_ext, g_ld, g_x, ws, nullptr, nullptr, stream);
}
int nfk_step2d_bwd_deferred(const nfk_step2d* s, const nfk_step2d_grads* g, const float* x, int ldx, const float* z,
                            const float* h1, const float* h2, const float* hsave, const float* g_out,
                            const float* g_ext, const float* g_ld, float* g_x, float* ws, float* ga, float* wslab,
                            void* stream) {
  return step2d_bwd_impl(s, g, x, ldx, z, h1, h2, hsave, g_out, g_ext, g_ld, g_x, ws, ga, wslab, stream);
}
int nfk_step2d_bwd_chain(const nfk_step2d* s, const nfk_step2d_grads* g, const float* x, int ldx, const float* z,
                         const float* h1, const float* h2, const float* hsave, const float* g_out, const float* g_ext,
                         const float* g_ld, float* g_x, float* ws, float* ga, float* wslab, int coupling_rows,
                         const nfk_step2d_next* next, int* next_rows, nfk_step2d_pending* pending, void* stream) {
  return step2d_bwd_impl(s, g, x, ldx, z, h1, h2, hsave, g_out, g_ext, g_ld, g_x, ws, ga, wslab, stream, coupling_rows,
                         next, next_rows, pending);
}
int nfk_step2d_pending_flush(nfk_step2d_pending* pending, void* stream) { return pending_flush(pending, S_(stream)); }
size_t nfk_step2d_bwd_ga_floats(const nfk_step2d* s) {
  const size_t M = (size_t)s->B * s->H * s->W;
  const StepBwdWs L = step_bwd_layout(s);
  return align4(M * s->Cout) + 2 * align4(M * s->hidden) + align4((M * L.K3p + 1) / 2);   // g_a3, g_a2, g_a1, dgrad3's im2col rows
}
size_t nfk_step2d_bwd_wslab_floats(const nfk_step2d* s) {
  const StepBwdWs L = step_bwd_layout(s);
  return (L.x3 - L.wslab) + align4((size_t)L.mb2 * 2 * s->hidden) + align4((size_t)L.mb1 * 2 * s->hidden);   // slabs, col2, col1
}
// ------------------------------------------------------------------------------------------
// backward of a whole level (include/nfk.h: nfk_level_bwd)
// ------------------------------------------------------------------------------------------
namespace {
struct LevelWs {
  size_t gz0, gz1, tailpart, ga, wslab, total;
  size_t ga_stride, o_ga2, o_ga1, o_x3;            // inside a step's gradient block
  size_t wslab_stride, o_s2, o_s1, o_col2, o_col1;   // inside a step's slab block
  int K3p, K1p, nparts;
};
LevelWs level_ws_layout(const nfk_step2d* s, int nsteps) {
  LevelWs w;
  const size_t M = (size_t)s->B * s->H * s->W, hid = (size_t)s->hidden;
  w.K3p = fusedb_k3p(*s); w.K1p = fused2_k1p(*s); w.nparts = actnorm_sum_parts((int)M);
  w.o_ga2 = align4(M * s->Cout);
  w.o_ga1 = w.o_ga2 + align4((M * hid + 1) / 2);
  w.o_x3 = w.o_ga1 + align4((M * hid + 1) / 2);
  w.ga_stride = w.o_x3 + align4((M * w.K3p + 1) / 2);
  const size_t S = NFK_LEVEL_MAX_SPLITS;
  w.o_s2 = align4(S * w.K3p * hid);
  w.o_s1 = w.o_s2 + align4(S * hid * hid);
  w.o_col2 = w.o_s1 + align4(S * hid * w.K1p);
  w.o_col1 = w.o_col2 + align4((size_t)w.nparts * 2 * hid);
  w.wslab_stride = w.o_col1 + align4((size_t)w.nparts * 2 * hid);
  size_t o = 0;
  w.gz0 = o; o += align4(M * s->C);
  w.gz1 = o; o += align4(M * s->C);
  w.tailpart = o; o += align4((size_t)nsteps * NFK_LEVEL_ROWS * 2 * s->Cout);
  w.ga = o; o += (size_t)nsteps * w.ga_stride;
  w.wslab = o; o += (size_t)nsteps * w.wslab_stride;
  w.total = o;
  return w;
}
}  // namespace

int nfk_level_bwd_ok(const nfk_step2d* first) {
  if (!first || !first->bf16 || first->hidden != 512 || first->B > 256) return 0;
  return fusedb_level_ok(*first);
}
int nfk_level_bwd_parts(const nfk_step2d* first) { return nfk_level_bwd_ok(first) ? fusedb_level_grid(*first) : 0; }
size_t nfk_level_bwd_ws_floats(const nfk_step2d* first, int nsteps) {
  return (first && nsteps > 0) ? level_ws_layout(first, nsteps).total : 0;
}

int nfk_level2d_bwd(const nfk_level_bwd* lv, const nfk_step2d* host_steps, void* stream) {
  hipStream_t st = S_(stream);
  if (!lv || !host_steps || lv->nsteps < 1 || lv->k_lo < 0 || lv->k_hi >= lv->nsteps || lv->k_lo > lv->k_hi) return NFK_ERR_ARG;
  const nfk_step2d* s = host_steps;   // dimensions of the level
  if (!nfk_level_bwd_ok(s)) return NFK_ERR_ARG;
  if (!lv->steps || !lv->grads || !lv->x0 || !lv->zsave || !lv->hsave || !lv->h1save || !lv->h2save || !lv->masks ||
      !lv->g_out || !lv->g_x || !lv->ws || (lv->nsteps > 1 && !lv->zout))
    return NFK_ERR_ARG;
  const int B = s->B, HW = s->H * s->W, M = B * HW, C = s->C, Cout = s->Cout;
  const LevelWs L = level_ws_layout(s, lv->nsteps);
  float* ws = lv->ws;
  const int kh = lv->k_hi;
  // 1. coupling backward of the top step (the steps below get theirs from the launch of the step above)
  CouplingBwdArgs cb{};
  cb.gout = lv->g_out; cb.ldg = C; cb.gext = lv->g_ext; cb.ldge = C; cb.gld = lv->g_ld;
  cb.z = lv->zsave + (long)kh * lv->zsave_stride; cb.ldz = C; cb.h = lv->hsave + (long)kh * lv->hsave_stride;
  cb.es3 = host_steps[kh].es3; cb.ga3 = ws + L.ga + (size_t)kh * L.ga_stride;
  cb.gz = ws + ((kh & 1) ? L.gz1 : L.gz0); cb.ldgz = C;
  cb.part = ws + L.tailpart + (size_t)kh * NFK_LEVEL_ROWS * 2 * Cout;
  cb.HW = HW; cb.C = C; cb.Cout = Cout; cb.additive = s->additive;
  cb.tiles = coupling_bwd_tiles(B, HW);
  if (B * cb.tiles > 256) return NFK_ERR_ARG;
  NFK_TRY(launch_coupling_bwd(cb, B, st));
  // 2. the steps' launches (or the one launch of the run)
  FusedLevelCall fc{};
  fc.lv = *lv; fc.first = *s;
  fc.gz[0] = ws + L.gz0; fc.gz[1] = ws + L.gz1; fc.tailpart = ws + L.tailpart;
  fc.stamps = g_bwd_stamps;
  fc.ga = ws + L.ga; fc.ga_stride = (long)L.ga_stride; fc.o_ga2 = (long)L.o_ga2; fc.o_ga1 = (long)L.o_ga1; fc.o_x3 = (long)L.o_x3;
  nfk_aux* const ax = lv->aux;
  LevelWgradArgs wa{};
  wa.steps = lv->steps; wa.grads = lv->grads;
  wa.B = B; wa.H = s->H; wa.W = s->W; wa.M = M; wa.hid = s->hidden; wa.Cout = Cout; wa.Ch = C / 2; wa.K3p = L.K3p; wa.K1p = L.K1p;
  wa.ga = ws + L.ga; wa.ga_stride = (long)L.ga_stride; wa.o_ga2 = (long)L.o_ga2; wa.o_ga1 = (long)L.o_ga1; wa.o_x3 = (long)L.o_x3;
  wa.h1 = lv->h1save; wa.h2 = lv->h2save; wa.h_stride = lv->h_stride;
  wa.masks = reinterpret_cast<const unsigned char*>(lv->masks); wa.mask_stride = lv->mask_stride;
  wa.o_x1 = (long)2 * M * (s->hidden / 8);
  wa.wslab = ws + L.wslab; wa.wslab_stride = (long)L.wslab_stride; wa.o_s2 = (long)L.o_s2; wa.o_s1 = (long)L.o_s1;
  wa.o_col2 = (long)L.o_col2; wa.o_col1 = (long)L.o_col1;
  wa.tailpart = ws + L.tailpart; wa.rows_top = B * cb.tiles; wa.k_top = kh;
  // steps in chunks: [the chunk's launches] then, on `aux` if there is one, its weight-gradient GEMMs + reductions -- which
  // run beside the next chunk's launches.  One chunk = the whole walk unless the caller asks for less (a run of steps in
  // one launch is always one chunk)
  const bool run = lv->multi && kh > lv->k_lo && fusedb_level_multi_ok(*s);
  const int chunk = (run || lv->wgrad_chunk <= 0) ? kh - lv->k_lo + 1 : lv->wgrad_chunk;
  for (int hi = kh; hi >= lv->k_lo; hi -= chunk) {
    const int lo = hi - chunk + 1 > lv->k_lo ? hi - chunk + 1 : lv->k_lo;
    fc.run_hi = hi; fc.run_lo = lo;
    int grid = 0;
    NFK_TRY(fusedb_level_launch(fc, st, &grid));
    hipStream_t wst = st;
    if (ax && ax->stream != st) { fork_level_to_aux(ax, st); wst = ax->stream; }
    wa.k_lo = lo; wa.k_hi = hi; wa.rows = grid;
    wa.S = level_wgrad_splits(M, s->hidden, L.K3p, L.K1p, hi - lo + 1);
    NFK_TRY(level_wgrad_launch(wa, wst, wst));
    if (lo == lv->k_lo && lv->prep_n > 0 && lv->prep_descs && lv->prep_grads) {   // behind the walk's last reductions
      if (lv->prep_n != kh - lv->k_lo + 1) return NFK_ERR_ARG;
      NFK_TRY(launch_prep_bwd(lv->prep_descs, lv->prep_grads, lv->prep_n, lv->g_ld, B, lv->prep_dense, wst));
    }
  }
  return NFK_OK;
}

void nfk_debug_triple_parts(int skip_mask) { triple_wgrad_debug_skip(skip_mask); }
void nfk_fused_bwd_set_stamps(void* buf) { g_bwd_stamps = reinterpret_cast<unsigned long long*>(buf); }
int nfk_step2d_fused_bwd_ok(const nfk_step2d* s) { return (s->bf16 && fusedb_grid(*s) > 0) ? 1 : 0; }
size_t nfk_step2d_mask_bytes(const nfk_step2d* s) {
  const size_t M = (size_t)s->B * s->H * s->W;
  return 2 * M * (s->hidden / 8) + M * fused2_k1p(*s) * 2;   // ReLU masks of h1 / h2, then conv1's im2col rows (bf16)
}
int nfk_aux_join(nfk_aux* ax, void* stream) {
  hipStream_t st = S_(stream);
  if (!ax) return NFK_OK;
  if (ax->stream != st && ax->forked) {              // nothing to wait for otherwise (and, while `st` is being captured,
    (void)hipEventRecord(ax->ev[3], ax->stream);     // an event of a stream outside the capture must not be waited on)
    (void)hipStreamWaitEvent(st, ax->ev[3], 0);
  }
  ax->forked = false;
  return NFK_OK;
}

// ------------------------------------------------------------------------------------------
// Split2d composites
// ------------------------------------------------------------------------------------------
int nfk_split2d_fwd(const nfk_split2d* s, const float* x, const float* ld_in, float* z1out, float* hsave,
                    float* ws, float* ld_out, void* stream) {
  hipStream_t st = S_(stream);
  const int M = s->B * s->H * s->W, Ch = s->C / 2;
  if (nfk_split2d_fused_ok(s)) {   // conv + prior in one launch, then the per-sample sums of the tiles' partials
    NFK_TRY(split2d_fused_fwd(*s, x, s->w, z1out, hsave, ws, st));
    return ld_out ? nfk_ld_finish(ld_in, ws, 1, s->B, split2d_fused_tps(*s), ld_out, stream) : NFK_OK;
  }
  ConvGemmArgs c = conv_args(x, s->C, s->B, s->H, s->W, Ch, 9, s->wt, s->C, ws, s->C, EPI_RAW);
  NFK_TRY(conv_gemm(c, st));
  SplitArgs a{};
  a.a = ws; a.S = conv_splitk(M, s->C, 9 * Ch, EPI_RAW); a.slab_stride = (long)M * s->C;
  a.bias = s->bias; a.es = s->es; a.x = x; a.ldx = s->C; a.z1out = z1out; a.hsave = hsave;
  a.ld_in = ld_in; a.ld_out = ld_out; a.HW = s->H * s->W; a.C = s->C; a.has_ld = ld_out != nullptr;
  return launch_split_fwd(a, s->B, st);
}
int nfk_split2d_inv(const nfk_split2d* s, const float* z1, const float* eps, float temperature, float* xout,
                    float* ws, void* stream) {
  hipStream_t st = S_(stream);
  const int M = s->B * s->H * s->W, Ch = s->C / 2;
  ConvGemmArgs c = conv_args(z1, Ch, s->B, s->H, s->W, Ch, 9, s->wt, s->C, ws, s->C, EPI_RAW);
  NFK_TRY(conv_gemm(c, st));
  SplitArgs a{};
  a.a = ws; a.S = conv_splitk(M, s->C, 9 * Ch, EPI_RAW); a.slab_stride = (long)M * s->C;
  a.bias = s->bias; a.es = s->es; a.x = z1; a.ldx = Ch; a.z1out = xout; a.HW = s->H * s->W; a.C = s->C;
  a.eps = eps; a.temperature = temperature;
  return launch_split_inv(a, s->B, st);
}
int nfk_split2d_bwd(const nfk_split2d* s, const float* x, const float* hsave, const float* g_z1,
                    const float* g_ext, const float* g_ld, float* g_x, float* g_w, float* g_bias, float* g_logs,
                    float* ws, void* stream) {
  hipStream_t st = S_(stream);
  const int B = s->B, H = s->H, W = s->W, C = s->C, Ch = C / 2, M = B * H * W;
  if (nfk_split2d_fused_ok(s)) {   // everything but the final reductions in one launch (nfk_split.hip)
    const int nt = split2d_fused_tiles(*s);
    float* part = ws;
    float* slab = part + align4((size_t)nt * 2 * C);
    NFK_TRY(split2d_fused_bwd(*s, x, hsave, s->w, g_z1, g_ext, g_ld, g_x, part, slab, st));
    nfk_small_grad_args sg{};
    sg.tail = part; sg.B = nt; sg.n3 = C; sg.g_bias3 = g_bias; sg.g_l3 = g_logs;
    StepTailArgs f{};
    f.slabs[0] = slab; f.gradW[0] = g_w; f.stride[0] = (long)C * 9 * Ch; f.S[0] = nt; f.Cout[0] = C; f.Cin[0] = Ch; f.taps[0] = 9;
    return launch_step_tail(sg, f, st);
  }
  const int Sd = conv_splitk(M, Ch, 9 * C, EPI_RAW), Sw = wgrad_splits(M, C, 9 * Ch);
  float* ga = ws;
  float* dslab = ga + align4((size_t)M * C);
  float* tail = dslab + align4((size_t)Sd * M * Ch);
  float* wslab = tail + align4((size_t)B * 2 * C);
  SplitBwdArgs b{};
  b.gld = g_ld; b.x = x; b.ldx = C; b.h = hsave; b.es = s->es; b.ga = ga; b.gx = g_x; b.ldgx = C; b.part = tail;
  b.HW = H * W; b.C = C;
  NFK_TRY(launch_split_bwd(b, B, st));
  WgradArgs w{};
  w.P = ga; w.ldp = C; w.NI = C; w.Q = x; w.ldq = C; w.Cq = Ch; w.taps = 9; w.B = B; w.H = H; w.W = W;
  w.out = wslab; w.slab_stride = (long)C * 9 * Ch;
  NFK_TRY(wgrad_gemm(w, Sw, st));
  ConvGemmArgs d = conv_args(ga, C, B, H, W, C, 9, s->wd, Ch, dslab, Ch, EPI_RAW);
  NFK_TRY(conv_gemm(d, st));
  NFK_TRY(launch_add_slabs(g_z1, Ch, g_ext, Ch, dslab, Sd, (long)M * Ch, g_x, C, M, Ch, st));
  // the weight gradient's slab reduction and the per-channel gradients in one launch (as at the end of a FlowStep)
  nfk_small_grad_args sg{};
  sg.tail = tail; sg.B = B; sg.n3 = C; sg.g_bias3 = g_bias; sg.g_l3 = g_logs;
  StepTailArgs f{};
  f.slabs[0] = wslab; f.gradW[0] = g_w; f.stride[0] = w.slab_stride; f.S[0] = Sw; f.Cout[0] = C; f.Cin[0] = Ch; f.taps[0] = 9;
  return launch_step_tail(sg, f, st);
}

}  // extern "C"
