/* nfk -- MI355X (gfx950) native kernels for the Glow coupling-flow hot path.
 *
 * C ABI of libnfk.so.  Plain pointers and sizes only: every pointer is a DEVICE pointer to
 * fp32 data unless stated otherwise, `stream` is a hipStream_t passed as void*, every function
 * returns 0 on success (NFK_ERR_* otherwise), nothing is allocated or synchronised inside, all
 * work is enqueued on the caller's stream (so a caller may capture it into a hipGraph).
 *
 * The reference (vklyukin/nf_distillation) has no FFI: its hot path is eager PyTorch.  Each entry
 * point below names the reference code it replaces (file:line under /root/reference).
 *
 * Activation layout: NHWC, i.e. rows of C channels per pixel, M = B*H*W rows ("ld*" = row stride
 * in floats).  1-D (tabular) data is the H = W = 1 case.
 */
#ifndef NFK_H_
#define NFK_H_
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NFK_OK 0
#define NFK_ERR_ARG 1
#define NFK_ERR_LAUNCH 2

enum { NFK_PERM_LU = 0, NFK_PERM_DENSE = 1, NFK_PERM_INDEX = 2 };

/* Parameter products of one FlowStep (ActNorm + permutation) and the exp() of the coupling
 * net's log-scales.  An array of these lives in DEVICE memory; one launch serves all steps.
 * Replaces models/layers.py:107-127 (_ActNorm._scale), :360-402 (InvertibleConv1x1.get_weight),
 * :227 / :260 (exp(logs) of ActNorm2d inside Conv2d, exp(3 logs) of Conv2dZeros). */
typedef struct nfk_prep_desc {
  int32_t C;          /* channels mixed by this step; 0 = no mixing (Split2d / learn_top conv) */
  int32_t pixels;     /* H*W (1 for 1-D): multiplies the parameter log-det                       */
  int32_t is_1d;      /* 1: z = y @ W (models/layers.py:411), 0: z = W y per pixel (:413)        */
  int32_t perm_kind;  /* NFK_PERM_*                                                              */
  int32_t n1, n2, n3; /* lengths of l1,l2 (ActNorm logs of block.0/.2) and l3 (Conv2dZeros logs) */
  int32_t pad_;
  const float *an_bias, *an_logs;                   /* [C]                    */
  const float *lower, *upper, *log_s, *p, *sign_s;  /* LU parametrisation     */
  const float *weight;                              /* dense [C][C]           */
  const int32_t *perm_idx;                          /* [C] forward gather idx */
  const float *l1, *l2, *l3;
  /* outputs */
  float *Wf, *cf;     /* fused forward matrix [C][C] (out-major) and offset [C]        */
  float *Wi, *ci;     /* fused inverse matrix / offset (written when want_inverse)     */
  float *dld;         /* [1] pixels * (sum logs + sum log|s|)                          */
  float *es1, *es2, *es3;
  float *scratch;     /* >= 5*C*C floats */
} nfk_prep_desc;

/* Where the backward of `prep` reads its partial sums and accumulates parameter gradients. */
typedef struct nfk_prep_grad {
  const float *part;  /* [nparts][C*C + C] partial (gWf, gcf) written by nfk_chanmix_bwd */
  int32_t nparts, pad_;
  float *g_an_bias, *g_an_logs, *g_lower, *g_upper, *g_log_s, *g_weight;
} nfk_prep_grad;

/* One 2-D FlowStep (models/flows.py:55-202) with a 3-conv coupling net (models/flows.py:25-34). */
typedef struct nfk_step2d {
  int32_t B, H, W, C, hidden, Cout; /* Cout = C - C/2 (additive) or 2*(C - C/2) (affine) */
  int32_t additive;
  int32_t bf16;                     /* 1: h1/h2 (and their grads) are bf16 in HBM; wt and wd are bf16 [N][Kp64] */
  const float *b1, *b2, *bias3;     /* block.0/.2 actnorm.bias [hidden], block.4 conv.bias [Cout] */
  const float *Wf, *cf, *Wi, *ci, *dld, *es1, *es2, *es3; /* from nfk_prep */
  const float *wt1, *wt2, *wt3;     /* repacked forward weights [N][Kp]  (nfk_repack) */
  const float *wd1, *wd2, *wd3;     /* repacked dgrad weights            (nfk_repack) */
  /* bf16 mode, optional (NULL = absent): the forward weights once more in MFMA-fragment order, so that a wave
   * fetches every A operand as 1 KiB of contiguous bytes (nfk_repack_desc.wf; used by nfk_step2d_fwd_fused) */
  const void *wf1, *wf2, *wf3;
  /* ... and the DGRAD weights in MFMA-fragment order (nfk_repack_desc.wg; used by the fused step backward,
   * csrc/nfk_fusedb.hip): wg3 / wg2 = 32x32x16 A fragments of wd3 [hidden][Kp(9 Cout)] / wd2 [hidden][hidden],
   * wg1 = 16x16x32 A fragments of wd1 [C/2 rounded up to 16][9 hidden] */
  const void *wg1, *wg2, *wg3;
} nfk_step2d;

typedef struct nfk_step2d_grads {   /* per-call pointers of a step backward; the gradients all accumulate (+=) */
  float *g_w1, *g_b1, *g_l1, *g_w2, *g_b2, *g_l2, *g_w3, *g_bias3, *g_l3;
  float *mix_part;                  /* [nfk_chanmix_blocks(M, C)][C*C + C], consumed by nfk_prep_bwd */
  const void *relu_masks;           /* what nfk_step2d_fwd_fused left in `masksave` for this step (ReLU masks, then conv1's
                                     * im2col rows), or NULL (the fused step backward needs them; without, the step runs
                                     * the per-conv launches) */
  int32_t bwd_mode, pad_;           /* NFK_BWD_*: how the bf16 backward of the coupling net is launched */
  struct nfk_aux *aux;              /* optional auxiliary stream for the weight-gradient GEMMs (nfk_aux_create), or NULL */
} nfk_step2d_grads;
/* bf16 backward of a step's coupling net:
 *   NFK_BWD_SEPARATE   every dgrad / weight-gradient GEMM its own launch (weight gradients on `aux` if one is given)
 *   NFK_BWD_DUAL_SMALL a conv's dgrad + weight gradient as ONE launch for steps of at most 4096 pixels
 *   NFK_BWD_DUAL       ... for every step
 *   NFK_BWD_FUSED      (= NFK_BWD_DEFAULT) as DUAL, and where the geometry of nfk_step2d_fwd_fused applies and relu_masks is
 *                      given the whole dgrad chain is ONE launch that keeps the 512-wide gradients in LDS
 *                      (csrc/nfk_fusedb.hip), followed by ONE launch for the step's three weight-gradient GEMMs.
 * The DUAL / FUSED launches ignore `aux`. */
enum { NFK_BWD_DEFAULT = 0, NFK_BWD_SEPARATE = 1, NFK_BWD_DUAL_SMALL = 2, NFK_BWD_DUAL = 3, NFK_BWD_FUSED = 4 };
/* An auxiliary stream plus the events that fork work to it and join it: created once per engine (two engines on two
 * threads share nothing), passed per call in nfk_step2d_grads.aux.  `stream` must outlive the handle. */
typedef struct nfk_aux nfk_aux;
nfk_aux *nfk_aux_create(void *stream);
void nfk_aux_destroy(nfk_aux *aux);

/* Split2d (models/layers.py:293-313): Conv2dZeros(C/2 -> C) prior on the dropped half. */
typedef struct nfk_split2d {
  int32_t B, H, W, C;
  const float *bias, *es;           /* conv.bias [C], exp(3 logs) [C] (prep es3) */
  const float *wt, *wd;             /* repacked weights */
  const float *w;                   /* conv.weight itself, [C][C/2][3][3]: read by the fused forward / backward (one launch each);
                                     * NULL -> the generic GEMM path */
} nfk_split2d;

/* One 1-D (tabular) FlowStep: ActNorm1d + x @ W + get_block_1d MLP + coupling tail
 * (models/flows.py:37-52, 142-202 with is_1d).  W[l]/b[l] are the nn.Linear weights [out][in] / biases of
 * block.{0,2,4,6,8,10}; Wf/cf/Wi/ci/dld come from nfk_prep. */
typedef struct nfk_tab_step {
  int32_t B, d, hid, ncond, additive;
  int32_t made;   /* 1: MADE / MAF block instead of a coupling (north_star's "MAF MADE blocks"; no reference code exists:
                   * SURVEY.md section 0 F1, parity unpinned): W[0] is [hid][d + ncond], W[5] is [2 d][hid], both already
                   * multiplied with the autoregressive masks by the caller; u_i = (y_i - mu_i) exp(-alpha_i) on every feature,
                   * log|det| = -sum alpha_i; the inverse runs d passes of the MLP.  0: the reference's coupling step */
  const float *W[6];
  const float *b[6];
  const float *Wf, *cf, *Wi, *ci, *dld;
} nfk_tab_step;

typedef struct nfk_tab_grads { /* accumulate (+=) */
  float *gW[6];
  float *gb[6];
} nfk_tab_grads;

/* One conv weight to repack (see nfk_repack); DEVICE array for nfk_repack_batch. */
typedef struct nfk_repack_desc {
  const float *w;
  float *wt, *wd; /* wd may be NULL */
  int32_t Cout, Cin, taps;
  int32_t frag;   /* layout of wf: 1 = 32x32x16 A fragments [Cout/32][Kp/16][64 lanes][8], Cout % 32 == 0;
                   *               2 = 16x16x32 A fragments [ceil(Cout/16)][Kp/32][64 lanes][8], zero rows behind Cout */
  void *wf;       /* bf16 repack only, may be NULL: fragment-ordered copy of wt (Kp = nfk_padded_k_bf16(taps*Cin)) */
  void *wg;       /* bf16 repack only, may be NULL: fragment-ordered copy of wd (rows = Cin, Kp = nfk_padded_k_bf16(taps*Cout)),
                   * layout `gfrag` as `frag`; for gfrag = 2 the rows behind Cin are NOT written (keep the buffer zeroed) */
  int32_t gfrag;
  int32_t flags;  /* bf16 repack only.  NFK_REPACK_FRAG_ONLY: the row-major copies wt / wd are not written where a fragment-
                   * ordered copy of the same weights (wf / wg) is -- for models whose every step runs the fused kernels, which
                   * read the latter only (the reverse pass and the per-conv launches read wt / wd: repack without the flag) */
} nfk_repack_desc;
#define NFK_REPACK_FRAG_ONLY 1

typedef struct nfk_small_grad_args {
  const float *col1, *col2, *tail;
  int32_t mb1, mb2, B, n1, n2, n3;
  float *g_l1, *g_b1, *g_l2, *g_b2, *g_bias3, *g_l3;
} nfk_small_grad_args;

/* ---- sizes ------------------------------------------------------------------------------- */
int nfk_padded_k(int K);                          /* K rounded up to the k-chunk (32)            */
int nfk_conv_splitk(int M, int N, int K);         /* slabs written by a raw-output conv          */
int nfk_chanmix_blocks(int M, int C);         /* partial (gWf,gcf) slabs a step backward writes  */
size_t nfk_step2d_fwd_ws_floats(const nfk_step2d *s);   /* a3 slab workspace                   */
size_t nfk_step2d_bwd_ws_floats(const nfk_step2d *s);
size_t nfk_split2d_fwd_ws_floats(const nfk_split2d *s);
size_t nfk_split2d_bwd_ws_floats(const nfk_split2d *s);
int nfk_split2d_fused_ok(const nfk_split2d *s);   /* 1: nfk_split2d_fwd / _bwd take the one-launch path for this geometry */

/* ---- parameter-side kernels ---------------------------------------------------------------- */
/* descs: DEVICE array[nsteps].  One workgroup per step. */
int nfk_prep(const nfk_prep_desc *descs, int nsteps, int want_inverse, void *stream);
int nfk_prep_bwd(const nfk_prep_desc *descs, const nfk_prep_grad *grads, int nsteps, const float *g_logdet,
                 int B, int any_dense, void *stream);
/* Backward of the INVERSE products of nfk_prep (Wi, ci), for a reverse pass that carries gradients (1-D perceptual
 * branch, pl_module.py:229-243): grads[i].part = [nparts][C*C + C] partial (g_Wi, g_ci) of step i (nparts == 0: step
 * skipped); g_logdet = dL/d(logdet of the reverse pass) or NULL.  Accumulates into the same parameter gradients as
 * nfk_prep_bwd.  C <= 74. */
int nfk_prep_inv_bwd(const nfk_prep_desc *descs, const nfk_prep_grad *grads, int nsteps, const float *g_logdet,
                     int B, void *stream);
/* conv.weight [Cout][Cin][taps] -> wt [Cout][Kp(taps*Cin)] and (optional) wd [Cin][Kp(taps*Cout)]
 * (flipped taps).  Replaces nn.ZeroPad2d + nn.Conv2d weight handling (models/layers.py:208-211). */
int nfk_repack(const float *w, int Cout, int Cin, int taps, float *wt, float *wd, void *stream);
/* all conv weights of a model in one launch; descs: DEVICE array[n] */
int nfk_repack_batch(const nfk_repack_desc *descs, int n, void *stream);
/* same, to bf16 with k padded to 64 (throughput mode: nfk_step2d.bf16 = 1) */
int nfk_repack_batch_bf16(const nfk_repack_desc *descs, int n, void *stream);
int nfk_padded_k_bf16(int K);
/* nfk_prep + nfk_repack_batch(f32_descs) + nfk_repack_batch_bf16(bf16_descs) in ONE launch (the head of every training
 * step in bf16 mode: the three are independent of each other).  Any of the three may be empty. */
int nfk_prep_repack(const nfk_prep_desc *descs, int nsteps, int want_inverse, const nfk_repack_desc *f32_descs, int n_f32,
                    const nfk_repack_desc *bf16_descs, int n_bf16, void *stream);

/* ---- op-level entry points (unit-testable primitives) -------------------------------------- */
/* z[m][:] = Wm x[m][:] + cv   -- _ActNorm.forward + InvertibleConv1x1.forward / Permute2d
 * (models/layers.py:129-142, 404-421, 283-290) in either direction, given prep's (Wf,cf)/(Wi,ci). */
int nfk_chanmix(const float *x, int ldx, const float *Wm, const float *cv, float *z, int ldz, int M, int C,
                void *stream);
/* implicit-GEMM convolution on the matrix cores; epi: 0 raw slabs, 1 (acc+e0)*e1 then ReLU.
 * Replaces Conv2d.forward / Conv2dZeros.forward (models/layers.py:223-228, 257-260). */
int nfk_conv(const float *x, int ldx, int B, int H, int W, int Cin, int taps, const float *wt, int N, float *out,
             int ldo, int epi, const float *e0, const float *e1, void *stream);
/* affine / additive coupling tail + per-sample log-det (models/flows.py:157-169 / 182-194) */
int nfk_coupling_fwd(const float *a3, int S, const float *bias3, const float *es3, const float *z, int ldz,
                     float *zout, int ldo, float *hsave, const float *ld_in, const float *dld, float *ld_out,
                     int B, int HW, int C, int Cout, int additive, void *stream);
int nfk_coupling_inv(const float *a3, int S, const float *bias3, const float *es3, const float *y, int ldy,
                     float *z, int ldz, const float *ld_in, const float *dld, float *ld_out, int B, int HW,
                     int C, int Cout, int additive, void *stream);
/* squeeze2d / unsqueeze2d (models/layers.py:32-61).  from_nchw: x is the NCHW model input, optionally
 * with the in-place dequantisation noise of uniform_binning_correction (models/utils.py:38). */
int nfk_squeeze(float *x, int ldx, const float *noise, float *out, int B, int C, int H, int W, int from_nchw,
                void *stream);
int nfk_unsqueeze(const float *y, float *out, int B, int C, int H, int W, int to_nchw, void *stream);
/* unsqueeze2d of y + y2 (same layout): the backward of a squeeze whose output is also a distillation hint receives two
 * gradients (kd_flows.py:42-53 returns that output; pl_module.py:262-282 puts an MSE on it) -- summed on the way through */
int nfk_unsqueeze_add(const float *y, const float *y2, float *out, int B, int C, int H, int W, int to_nchw, void *stream);
/* Glow.normal_flow tail (models/kd_flows.py:136-150): objective = logdet + log N(z; prior), bits/dim */
int nfk_prior_fwd(const float *z, int ldz, const float *prior, long prior_bstride, const float *ld,
                  float *objective, float *bpd, int B, int HW, int C, float bpd_scale, void *stream);
int nfk_prior_bwd(const float *z, int ldz, const float *prior, long prior_bstride, const float *g_bpd,
                  float *g_z, int ldgz, float *g_ld, float *g_prior, int B, int HW, int C, float bpd_scale,
                  void *stream);

/* ---- layer-level composites ---------------------------------------------------------------- */
/* FlowStep.normal_flow (models/flows.py:142-171).  z,h1,h2,hsave are the activations kept for the
 * backward (hsave may be NULL when no backward follows; z/h1/h2 may then be reused scratch). */
int nfk_step2d_fwd(const nfk_step2d *s, const float *x, int ldx, const float *ld_in, float *z, float *h1,
                   float *h2, float *hsave, float *ws, float *zout, float *ld_out, void *stream);
/* FlowStep.reverse_flow (models/flows.py:173-202): y -> x.  ld_* may be NULL.  hsave: NULL, or [M][Cout] where the
 * coupling net's (shift | scale-raw) output is kept for nfk_step2d_inv_bwd (z, h1, h2 must then be kept too). */
int nfk_step2d_inv(const nfk_step2d *s, const float *y, int ldy, const float *ld_in, float *z, float *h1,
                   float *h2, float *hsave, float *ws, float *xout, float *ld_out, void *stream);
/* backward of nfk_step2d_inv -- the reverse pass with gradients (pl_module.py:229-243 runs the student's reverse pass under
 * autograd for the perceptual loss): g_xout = dL/dx, g_ld = dL/d(logdet_out) or NULL -> g_y [M][C] = dL/dy, the conv-net
 * parameter gradients accumulated through `g` as in nfk_step2d_bwd, and the partial (g_Wi, g_ci) rows of the inverse channel
 * mix in g->mix_part for nfk_prep_inv_bwd.  ws: nfk_step2d_bwd_ws_floats floats.  Plain launches on `stream` (this pass is
 * not on the benchmark's path: no fused / merged kernels). */
int nfk_step2d_inv_bwd(const nfk_step2d *s, const nfk_step2d_grads *g, const float *y, int ldy, const float *z,
                       const float *h1, const float *h2, const float *hsave, const float *g_xout, const float *g_ld,
                       float *g_y, float *ws, void *stream);
/* backward of nfk_step2d_fwd: g_out (+ optional g_ext) and g_ld -> g_x, parameter grads. */
int nfk_step2d_bwd(const nfk_step2d *s, const nfk_step2d_grads *g, const float *x, int ldx, const float *z,
                   const float *h1, const float *h2, const float *hsave, const float *g_out,
                   const float *g_ext, const float *g_ld, float *g_x, float *ws, void *stream);
/* Same, for a caller that runs many steps back to back with an auxiliary stream (g->aux): the three gradient tensors the
 * weight-gradient GEMMs read are placed in `ga` (nfk_step2d_bwd_ga_floats floats, a buffer of its own for every step until
 * nfk_aux_join) and their split slabs in `wslab` (nfk_step2d_bwd_wslab_floats floats, may be shared by all steps: only the
 * auxiliary stream touches it), so that the auxiliary stream is NOT joined inside the call and the dgrad chain of the next
 * step starts at once.  nfk_aux_join makes `stream` wait for everything queued on the auxiliary stream; the weight
 * gradients are complete only after it.  Without an auxiliary stream (or with NULL ga / wslab) this is nfk_step2d_bwd. */
int nfk_step2d_bwd_deferred(const nfk_step2d *s, const nfk_step2d_grads *g, const float *x, int ldx, const float *z,
                            const float *h1, const float *h2, const float *hsave, const float *g_out,
                            const float *g_ext, const float *g_ld, float *g_x, float *ws, float *ga, float *wslab,
                            void *stream);
/* Chained form for a run of FlowSteps of one level walked in backward order (models/flows.py:178-186 reversed by
 * autograd).  A step's backward ends in the channel-mix launch that produces g_x -- which is the output gradient of the
 * step that comes next in backward order -- so that launch can also run the NEXT step's coupling-tail backward
 * (models/layers.py:377-392 differentiated) on the pixel tile it holds, and the next call starts at its first GEMM.
 *   next          (optional) describes the step the following call will process: its descriptor, saved z / hsave, its
 *                 external output gradient (hint grads; NULL if none) and the ws / ga buffers THAT call will receive,
 *                 which must differ from this call's (two workspaces used alternately).
 *   coupling_rows 0: this call runs its own coupling backward (as nfk_step2d_bwd); > 0: the previous call already did,
 *                 passing this step as `next`, and left that many partial rows (the value it returned in *next_rows).
 * Same B, H, W, C for both steps.  ga / wslab as in nfk_step2d_bwd_deferred (NULL, NULL = nfk_step2d_bwd). */
/* pending (optional, with the fused backward and no auxiliary-stream deferral): lets a call put its three weight-gradient
 * GEMMs into ONE launch with the channel mix that ends the step (two launches on a step's critical path instead of
 * three).  The reductions that wait for those GEMMs -- split slabs into the weight gradients, ActNorm partial rows -- are
 * then owed: the call stores them in `pending`, the next call (any geometry) takes them into its own merged launch, and
 * nfk_step2d_pending_flush runs what is owed after the last one.  A caller zero-initialises the struct once per
 * backward and must keep the workspace of the owing call untouched until the next call / the flush has been enqueued
 * (two workspaces used alternately do). */
typedef struct nfk_step2d_pending {
  int32_t valid, pad_;
  uint64_t blob[40];
} nfk_step2d_pending;
typedef struct nfk_step2d_next {
  const nfk_step2d *s;
  const float *z, *hsave;   /* saved by that step's forward */
  const float *g_ext;       /* [M][C] or NULL */
  float *ws, *ga;           /* what the following call passes as ws / ga (ga NULL outside deferred mode) */
} nfk_step2d_next;
int nfk_step2d_bwd_chain(const nfk_step2d *s, const nfk_step2d_grads *g, const float *x, int ldx, const float *z,
                         const float *h1, const float *h2, const float *hsave, const float *g_out, const float *g_ext,
                         const float *g_ld, float *g_x, float *ws, float *ga, float *wslab, int coupling_rows,
                         const nfk_step2d_next *next, int *next_rows, nfk_step2d_pending *pending, void *stream);
int nfk_step2d_pending_flush(nfk_step2d_pending *pending, void *stream);
size_t nfk_step2d_bwd_ga_floats(const nfk_step2d *s);
size_t nfk_step2d_bwd_wslab_floats(const nfk_step2d *s);
int nfk_aux_join(nfk_aux *aux, void *stream);
/* 1 if nfk_step2d_bwd* (bwd_mode NFK_BWD_FUSED / DEFAULT) would run this step through the fused backward (needs s->wg1..3
 * and relu_masks), else 0 */
int nfk_step2d_fused_bwd_ok(const nfk_step2d *s);
size_t nfk_step2d_mask_bytes(const nfk_step2d *s);   /* size of nfk_step2d_fwd_fused's masksave */
/* ---- backward of a whole LEVEL (a run of FlowSteps of one geometry; bf16 fused path) ---------------------------------
 * Autograd of models/flows.py:142-171 through all K steps of a level, as the mirror of nfk_run2d_fwd_fused:
 *   per step ONE launch  [dgrad chain -> channel-mix backward (g_x, partial gWf / gcf) -> coupling backward of the step
 *                         below (g_a3, g_z2, partial bias3 / logs3 sums)]
 *   -- or, where the pixel tiles are whole images (no halo rows: nfk_run2d_fused_ok), ONE launch that walks all K steps
 *   with the gradient tile in LDS between them --, then
 *   per level ONE launch with the 3 K weight-gradient GEMMs and the 2 K ActNorm sums (few, long split-K ranges instead of
 *   16 short ones per step), and ONE launch with every reduction (split slabs, partial rows) into the parameter gradients.
 * Everything the forward kept is addressed as base + k * stride (k = step index in forward order), as nfk_fused_run does;
 * `steps` / `grads` are DEVICE arrays, `host_steps` a HOST copy of `steps` (`first` = its element 0).  g_ext (optional) is added to g_out; steps below the
 * last one take no external gradient (a caller with hint gradients inside the level calls per sub-run: k_lo / k_hi).
 * mix_part of step k receives nfk_level_bwd_parts(first) partial slabs (nfk_prep_grad.nparts). */
typedef struct nfk_level_grads {
  float *g_w1, *g_b1, *g_l1, *g_w2, *g_b2, *g_l2, *g_w3, *g_bias3, *g_l3;
  float *mix_part;
} nfk_level_grads;
typedef struct nfk_level_bwd {
  const nfk_step2d *steps;
  const nfk_level_grads *grads;
  int32_t nsteps, ldx0;
  const float *x0;                               /* input of step 0: [M][ldx0] */
  const float *zout; int64_t zout_stride;        /* outputs of the steps: the input of step k >= 1 is zout + (k-1) * stride */
  const float *zsave; int64_t zsave_stride;      /* mixed z                                  (floats) */
  const float *hsave; int64_t hsave_stride;      /* (shift | scale) of the coupling tail      (floats) */
  const void *h1save, *h2save; int64_t h_stride; /* [M][hidden] bf16                          (elements) */
  const void *masks; int64_t mask_stride;        /* ReLU masks + conv1's im2col rows          (bytes) */
  const float *g_out, *g_ext, *g_ld;             /* wrt the output of step k_hi (g_ext may be NULL), wrt the log-det */
  float *g_x;                                    /* out: wrt the input of step k_lo */
  float *ws;                                     /* nfk_level_bwd_ws_floats(first, nsteps) floats */
  int32_t k_lo, k_hi;                            /* steps k_hi, k_hi - 1, ..., k_lo are walked (0 <= k_lo <= k_hi < nsteps) */
  int32_t multi;                                 /* 1: one launch for all of them where the geometry allows, 0: one per step */
  nfk_aux *aux;                                  /* optional: the two per-level launches go to this stream (join: nfk_aux_join) */
  int32_t wgrad_chunk, pad_;                     /* > 0 (one launch per step only): the weight-gradient / reduction launches go out
                                                  * for every `wgrad_chunk` steps walked instead of once per call -- on `aux` they
                                                  * then run beside the remaining steps' launches */
  /* optional: the parameter-side chain (nfk_prep_bwd) of the walked steps -- entry 0 of both arrays = step k_lo, prep_n =
   * k_hi - k_lo + 1 -- goes out behind the level's reductions on the same stream (on `aux`: under the next level's launches
   * instead of at the end of the backward).  The caller then leaves these steps out of its final nfk_prep_bwd
   * (nfk_prep_grad.part == NULL skips a step there). */
  const nfk_prep_desc *prep_descs;
  const nfk_prep_grad *prep_grads;
  int32_t prep_n, prep_dense;
} nfk_level_bwd;
int nfk_level_bwd_ok(const nfk_step2d *first);        /* 1: nfk_level2d_bwd covers this geometry */
int nfk_level_bwd_parts(const nfk_step2d *first);     /* partial (gWf, gcf) slabs per step = workgroups of the step launch */
size_t nfk_level_bwd_ws_floats(const nfk_step2d *first, int nsteps);
int nfk_level2d_bwd(const nfk_level_bwd *lv, const nfk_step2d *host_steps, void *stream);
/* diagnostic: leave parts of the fused backward's weight-gradient launch out, to time the others (bit 0: ActNorm sums,
 * 1: conv2, 2: conv3, 3: conv1); gradients are then wrong.  0 = off */
void nfk_debug_triple_parts(int skip_mask);
/* diagnostic: device buffer [workgroups][16] of shader-clock stamps per phase of the fused backward (NULL = off) */
void nfk_fused_bwd_set_stamps(void *buf);
/* Fused forward of one FlowStep in ONE launch (bf16 mode, 64-pixel row tiles; see csrc/nfk_fused2.hip).
 * nfk_step2d_fused_tiles: 0 if this step's geometry is not supported (use nfk_step2d_fwd), else the number of
 * log-det partial slots per sample.  zsave/h1save/h2save/hsave as in nfk_step2d_fwd (NULL when no backward
 * follows; h1save/h2save are bf16 [M][hidden]).  The per-sample log-det is NOT updated here: the step writes
 * ldpart[B][tiles] and nfk_ld_finish adds the partials of a run of steps to ld_in in fixed order.
 * masksave (optional, with h1save / h2save): [2][M][hidden/8] bytes, bit c%8 of byte [k][m][c/8] = (h_{k+1}[m][c] > 0), the
 * ReLU masks of the two hidden activations -- all the fused step backward needs of them (`relu_masks` of nfk_step2d_grads).
 * split_ways: 4, 2 or 0.  "Hidden split": a grid that would fill only 16-64 CUs gives every pixel tile to that many
 * workgroups, each with its share of conv2 / conv3, partial sums in ws (nfk_step2d_fused_ws_floats(s) floats; 0 when the
 * geometry never splits), finished by the coupling kernel -- zsave must then be non-NULL even when no backward follows.
 * split_ways == 0 or ws == NULL: no split (the step runs on its 16-64 workgroups). */
int nfk_step2d_fused_tiles(const nfk_step2d *s);
size_t nfk_step2d_fused_ws_floats(const nfk_step2d *s);
int nfk_step2d_fwd_fused(const nfk_step2d *s, const float *x, int ldx, float *zsave, void *h1save, void *h2save,
                         float *hsave, float *zout, float *ldpart, float *ws, void *masksave, int split_ways,
                         void *stream);
/* diagnostic: device buffer [workgroups][16] of shader-clock stamps per phase of the fused kernel (NULL = off) */
void nfk_fused_set_stamps(void *buf);
/* A RUN of FlowSteps of one level in ONE launch (models/kd_flows.py:42-53 walks them one by one): for geometries whose
 * pixel tiles are whole images (H*W <= 64: CIFAR levels 2 and 3) a step's output tile is the next step's input tile, so
 * the activations stay in LDS from step to step and only what somebody reads goes to HBM.  Bit-identical to calling
 * nfk_step2d_fwd_fused (split_ways 0) once per step.
 *   steps     DEVICE array [nsteps] of descriptors with identical B, H, W, C, hidden, Cout, additive (nsteps <= 64)
 *   keep      bit k set: the output of step k is written to zout + slot * zout_stride (floats), slot = number of set
 *             bits below k; the last step's bit must be set.  A training forward sets all bits (a step's output is what
 *             the next step's backward reads as its input).
 *   zsave, h1save, h2save, hsave, masksave: per-step buffers as in nfk_step2d_fwd_fused, step k at base + k * stride
 *             (strides in floats / bf16 elements / floats / bytes); all NULL when no backward follows
 *   ldpart    [nsteps][B][tiles] log-det partials, finished by nfk_ld_finish as for single steps
 * nfk_run2d_fused_ok: 1 if this step geometry can run as such a run.  `first`: HOST copy of steps[0]. */
typedef struct nfk_fused_run {
  const nfk_step2d *steps;
  int32_t nsteps, ldx;
  const float *x;
  uint64_t keep;
  float *zout; int64_t zout_stride;
  float *zsave; int64_t zsave_stride;
  void *h1save, *h2save; int64_t h_stride;
  float *hsave; int64_t hsave_stride;
  void *masksave; int64_t mask_stride;
  float *ldpart;
} nfk_fused_run;
int nfk_run2d_fused_ok(const nfk_step2d *s);
int nfk_run2d_fwd_fused(const nfk_fused_run *run, const nfk_step2d *first, void *stream);
int nfk_ld_finish(const float *ld_in, const float *ldpart, int nsteps, int B, int tiles, float *ld_out, void *stream);
/* The same for up to NFK_LD_MAX_PARTS partial buffers in sequence (the steps of a level, the Split2d behind it, the next
 * level, ...): the arithmetic of one nfk_ld_finish per buffer, each continuing from the previous result, in ONE launch
 * (nothing between them reads the running log-det).  `parts` is a host array. */
#define NFK_LD_MAX_PARTS 8
typedef struct nfk_ld_parts {
  const float *part;       /* [nsteps][B][tiles] */
  int32_t nsteps, tiles;
} nfk_ld_parts;
int nfk_ld_finish_multi(const float *ld_in, const nfk_ld_parts *parts, int n, int B, float *ld_out, void *stream);
/* Workgroups per sample whose log-density partials nfk_split2d_fwd leaves at the start of `ws` when it is called with
 * ld_out == NULL on a geometry nfk_split2d_fused_ok accepts (0 otherwise): to be finished by nfk_ld_finish(_multi). */
int nfk_split2d_fwd_tiles(const nfk_split2d *s);
/* Split2d.forward / reverse / backward */
int nfk_split2d_fwd(const nfk_split2d *s, const float *x, const float *ld_in, float *z1out, float *hsave,
                    float *ws, float *ld_out, void *stream);
int nfk_split2d_inv(const nfk_split2d *s, const float *z1, const float *eps, float temperature, float *xout,
                    float *ws, void *stream);
int nfk_split2d_bwd(const nfk_split2d *s, const float *x, const float *hsave, const float *g_z1,
                    const float *g_ext, const float *g_ld, float *g_x, float *g_w, float *g_bias,
                    float *g_logs, float *ws, void *stream);

/* ---- 1-D (tabular) FlowStep: one fused kernel per direction ----------------------------------- */
int nfk_tab_bwd_blocks(const nfk_tab_step *s);    /* slabs of `part` / `part_mix` the backward writes */
int nfk_tab_param_floats(const nfk_tab_step *s);  /* floats per `part` slab                            */
/* FlowStep.normal_flow, 1-D: x [B][d] (+ cond [B][ncond] or NULL) -> zout, ld_out = ld_in + dlogdet */
int nfk_tab_fwd(const nfk_tab_step *s, const float *x, const float *cond, const float *ld_in, float *zout,
                float *ld_out, void *stream);
/* FlowStep.reverse_flow, 1-D: y -> x (ld_* may be NULL) */
int nfk_tab_inv(const nfk_tab_step *s, const float *y, const float *cond, const float *ld_in, float *xout,
                float *ld_out, void *stream);
/* backward: recomputes the forward per row tile in LDS; part [blocks][param_floats], part_mix [blocks][d*d+d] */
int nfk_tab_bwd(const nfk_tab_step *s, const nfk_tab_grads *g, const float *x, const float *cond,
                const float *g_out, const float *g_ld, float *g_x, float *g_cond, float *part, float *part_mix,
                void *stream);

/* backward of nfk_tab_inv (reverse pass with gradients): g_x = dL/dx of the reverse OUTPUT -> g_y (its input), g_cond,
 * MLP parameter gradients (via part, as nfk_tab_bwd) and part_mix = partial (g_Wi, g_ci) for nfk_prep_inv_bwd.
 * g_ld = dL/d(ld_out) or NULL. */
int nfk_tab_inv_bwd(const nfk_tab_step *s, const nfk_tab_grads *g, const float *y, const float *cond,
                    const float *g_x, const float *g_ld, float *g_y, float *g_cond, float *part, float *part_mix,
                    void *stream);

/* ---- optimiser (torch.optim.Adam + clip_grad_norm_, pl_module.py:350-354, train.py:46) ------ */
/* sumsq[0] += sum g^2 over n elements (deterministic two-stage tree); partial: >= 1024 floats */
int nfk_sumsq(const float *g, size_t n, float *partial, float *sumsq, void *stream);

/* Image pipeline on the device (data/src/utils.py:7-25 behind torchvision's ToTensor; data/src/cifar10.py:17-20):
 * preprocess: uint8 [B][H][W][C] -> fp32 [B][C][H][W], x = ((u/255)*255)/256 - 0.5 in the reference's own order of fp32
 * operations (bit-identical), flip[b] != 0 mirrors sample b horizontally (may be NULL);
 * postprocess: fp32 NCHW -> uint8 HWC, clamp(x,-0.5,0.5)+0.5, *255, clamp [0,255], truncation. */
int nfk_preprocess_u8(const unsigned char *u8_hwc, const unsigned char *flip, float *out_nchw, int B, int H, int W, int C,
                      void *stream);
int nfk_postprocess_u8(const float *x_nchw, unsigned char *u8_hwc, int B, int H, int W, int C, void *stream);

/* Distillation loss of the hint-MSE + NLL branch of NFModel.loss (pl_module.py:246-300):
 *   kd[b] = 1/P * sum_p mean_i (s_p[b][i] - t_p[b][i])^2,   result = mean_b (w_nll * nll[b] + w_kd * kd[b]).
 * s_p / t_p: P pairs of dense per-sample-contiguous tensors with per_sample[p] elements per sample (same memory order
 * in both).  fwd: part [P][B] scratch, kd_out [B], out3 = {result, mean nll, mean kd}; fixed-order sums.
 * bwd: g_result is a device scalar; writes g_s[p] (shape of s_p) and g_nll [B] (may be NULL). */
#define NFK_KD_MAX_PAIRS 16
typedef struct nfk_kd_pairs {
  int32_t n, B;         /* n = number of (student, teacher) hint pairs = min(len(s_idx), len(t_idx)): the reference ZIPS
                           the two index lists (pl_module.py:264-267) */
  int32_t divisor, pad_; /* len(student_kd_indices) (pl_module.py:279-281), >= n; 0 means n */
  const float *weights; /* optional per-sample weights multiplied into the loss (pl_module.py:309-310), or NULL */
  const float *s[NFK_KD_MAX_PAIRS], *t[NFK_KD_MAX_PAIRS];
  int64_t per_sample[NFK_KD_MAX_PAIRS];
} nfk_kd_pairs;
typedef struct nfk_kd_grads {
  float *g_s[NFK_KD_MAX_PAIRS];
  float *g_nll;
} nfk_kd_grads;
int nfk_kd_loss_fwd(const nfk_kd_pairs *pairs, const float *nll, float w_nll, float w_kd, float *part, float *kd_out,
                    float *out3, void *stream);
int nfk_kd_loss_bwd(const nfk_kd_pairs *pairs, const float *g_result, float w_nll, float w_kd,
                    const nfk_kd_grads *grads, void *stream);
/* p -= lr_t * m_hat / (sqrt(v_hat) + eps) with g scaled by min(1, clip/(sqrt(sumsq)+1e-6)) * gscale */
int nfk_adam_step(float *p, const float *g, float *m, float *v, size_t n, const float *sumsq, float clip,
                  float gscale, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                  void *stream);

/* ---- optional kernel timing (bench.py roofline leg) ------------------------------------------
 * Between begin/end every GEMM-engine launch is bracketed by hipEvents on its own stream and
 * attributed to a kernel class; `end` synchronises the device and returns, per class, total
 * milliseconds, launches, algorithmic FLOPs (2*M*N*K) and algorithmic bytes.  Arrays of
 * NFK_PROF_CLASSES. */
enum { NFK_PROF_CONV_WIDE128 = 0, NFK_PROF_CONV_WIDE64 = 1, NFK_PROF_CONV_NARROW = 2, NFK_PROF_WGRAD_WIDE = 3,
       NFK_PROF_WGRAD_NARROW = 4,
       /* nfk_step2d_fwd_fused by conv3 width (one template instance each for Glow's shapes: CIFAR levels 1 / 2 / 3):
        * flops of the three convs, bytes = x + z + weights (+ what a training forward saves) */
       NFK_PROF_FUSED_STEP = 5, NFK_PROF_FUSED_STEP_NT2 = 6, NFK_PROF_FUSED_STEP_NT3 = 7,
       /* dgrad + wgrad of one conv in one launch (NFK_BWD_DUAL): 9 classes = conv (block.4, .2, .0) x size bucket
        * (M >= 8192, >= 2048, smaller: CIFAR levels 1 / 2 / 3 at B = 64), one template instance each for Glow's shapes */
       NFK_PROF_DUAL_BWD = 8,
       /* fused step backward (NFK_BWD_FUSED, default in bf16 mode where the geometry allows): the dgrad chain of a
        * FlowStep in one launch (3 classes by size bucket as above) and its three weight-gradient GEMMs in one launch */
       NFK_PROF_FUSED_BWD = 17, NFK_PROF_TRIPLE_WGRAD = 20,
       NFK_PROF_CLASSES = 23 };
void nfk_profile_begin(void);
/* Process-wide experiment knobs of the GEMM launch heuristics (tools/, A/B tests of kernel variants): not part of the data
 * path's interface, defaults are what production uses. */
enum { NFK_TUNE_F32_TILES = 0,               /* fp32 wide conv: 0 default, 1..5 alternative tile configurations */
       NFK_TUNE_BF16_BIG_TILE_MIN_K64 = 1,   /* bf16 128x128 tiles only for K > 64 * value */
       NFK_TUNE_WGRAD_MAX_SPLITS = 2,        /* cap of the weight-gradient split-K */
       NFK_TUNE_WGRAD_TARGET_WGS = 3,        /* workgroups a weight-gradient launch aims at */
       NFK_TUNE_WGRAD_WIDE_MAX_SPLITS = 4,   /* cap for the 512x512 weight gradient */
       NFK_TUNE_NARROW_HALO_DGRAD = 5,       /* 1/0: conv1's dgrad through the halo-tile kernel / the split-K GEMM */
       NFK_TUNE_WGRAD_HALO = 6,              /* 1/0: conv3's weight gradient through the halo-tile kernel / the generic one */
       NFK_TUNE_WGRAD_HALO_MAX_SPLITS = 7,
       NFK_TUNE_RUN_HALF_TILES = 8,          /* 1/0: runs of steps on tiny images use 32-pixel tiles when the grid is small */
       NFK_TUNE_SPLIT_FUSED = 9,             /* 1/0: Split2d forward / backward as one fused fp32 launch each (default 1) */
       NFK_TUNE_TAB_RESIDENT = 10 };         /* 1/0: 1-D kernels keep every weight matrix resident in LDS per workgroup (default 1) / stage per tile */
void nfk_tune(int knob, int value);
void nfk_profile_end(double *ms, long long *launches, double *flops, double *bytes);

/* same, with the step count kept on the device (*step_dev = completed steps, incremented here), so the
 * call sequence is identical every step and can be replayed from a captured hipGraph */
int nfk_adam_step_dev(float *p, const float *g, float *m, float *v, size_t n, const float *sumsq, float clip,
                      float gscale, float lr, float beta1, float beta2, float eps, float weight_decay,
                      int *step_dev, void *stream);

const char *nfk_version(void);

#ifdef __cplusplus
}
#endif
#endif /* NFK_H_ */
