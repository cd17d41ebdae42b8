// Argument structs and launcher prototypes shared between the nfk translation units.
#pragma once
#include "nfk_device.h"
#include "nfk_structs.h"

namespace nfk {

enum { EPI_RAW = 0, EPI_AN_RELU = 1, EPI_DGRAD = 2 };

struct ConvGemmArgs {
  const float* X;   // activations, NHWC rows: X[pixel*ldx + ci]
  int ldx, B, H, W, Cin, taps, M, K;
  const float* Wt;  // repacked weights [N][Kp], zero padded in k
  int Kp, N;
  float* out;       // out[slab*slab_stride + m*ldo + n]
  int ldo;
  long slab_stride;
  int epi;
  const float* e0;     // AN_RELU: bias[n]
  const float* e1;     // AN_RELU / DGRAD: scale[n]
  const float* hprev;  // DGRAD: post-ReLU activation of the layer being differentiated
  int ldh;
  float* colsum;       // DGRAD: [mblocks][2][N] partial column sums (dlogs, dbias)
  int nkchunks;        // ceil(K/32)  (bf16 kernel: ceil(K/64))
  // bf16 mode (conv_gemm_bf16): Wt, and unless xf32 also X, are bf16; hprev is bf16; out is bf16 when obf16
  int xf32, obf16;
};

struct WgradArgs {
  const float* P; int ldp, NI;
  const float* Q; int ldq, Cq, taps;
  int B, H, W, M, J;   // J = taps*Cq
  float* out; long slab_stride;  // out[slab][i*J + j]
  int pbf16, qbf16;              // P / Q stored as bf16 (converted to f32 while staging)
};

struct ChanMixArgs {
  const float* x; int ldx;
  const float* Wm; const float* cv;
  float* z; int ldz;
  int M, C, trans;
  // backward-only gather of g_z: first `Ch` channels = ga[m][c] + sum_s slab[s][m][c], rest from x
  const float* ga; int lda; const float* gext; int ldge; const float* slab; int S; long slab_stride; int Ch;
  // backward-only partial outer products: part[block][C*C + C]
  const float* xin; int ldxin; float* part;
};

struct CouplingArgs {
  const float* a3; int S; long slab_stride;   // conv3 raw partial sums [S][M][Cout]
  const float* bias3; const float* es3;       // [Cout]
  const float* z; int ldz;                    // [M][C] mixed input (z1 | z2)
  float* zout; int ldo;                       // [M][C]
  float* hsave;                               // [M][Cout] or null
  const float* ld_in; const float* dld; float* ld_out;   // [B], scalar, [B]
  int HW, C, Cout, additive, has_ld;
};

struct CouplingBwdArgs {
  const float* gout; int ldg;      // grad wrt step output [M][C] (already includes hint grads)
  const float* gext; int ldge;     // optional extra grad to add (external hint) or null
  const float* gld;                // [B] grad wrt logdet
  const float* z; int ldz;         // saved mixed z [M][C]
  const float* h;                  // saved h [M][Cout]
  const float* es3;
  float* ga3;                      // [M][Cout] = g_h * es3  (operand of dgrad3 / wgrad3)
  float* gz; int ldgz;             // [M][C]: second half written here (g_z2)
  float* part;                     // [B * tiles][2][Cout]: (sum g_a3, 3 * sum g_h h)
  int HW, C, Cout, additive;
  int tiles;                       // stand-alone kernel: workgroups per sample (0 / 1: one); unused by the fused forms
};

struct SplitArgs {
  const float* a; int S; long slab_stride;  // conv raw sums [S][M][C]
  const float* bias; const float* es;       // [C]
  const float* x; int ldx;                  // [M][C] input (z1 | z2)
  float* z1out;                             // [M][C/2]
  float* hsave;                             // [M][C] (mean,logs interleaved) or null
  const float* ld_in; float* ld_out;
  int HW, C, has_ld;
  // inverse only
  const float* eps; float temperature;      // eps [M][C/2] or null
};

struct SplitBwdArgs {
  const float* gld;              // [B]
  const float* x; int ldx;       // saved input [M][C]
  const float* h;                // saved (mean,logs) [M][C]
  const float* es;
  float* ga;                     // [M][C] grad wrt raw conv sums
  float* gx; int ldgx;           // [M][C]: second half written (g_z2)
  float* part;                   // [B][2][C]
  int HW, C;
};

struct PriorArgs {
  const float* z; int ldz;         // [B][HW][C] NHWC (1-D: HW = 1)
  const float* ph; long ph_bstride; // prior tensor [b?][2C][HW] (NCHW order), batch stride (0 = shared)
  const float* ld; float* objective; float* bpd;
  int HW, C; float bpd_scale;      // bpd = -objective * bpd_scale
};

struct PriorBwdArgs {
  const float* z; int ldz; const float* ph; long ph_bstride;
  const float* gbpd;        // [B]
  float* gz; int ldgz;      // [B][HW][C]
  float* gld;               // [B]  = d loss / d logdet
  float* gph;               // [B][2C][HW] or null
  int HW, C; float bpd_scale;
};

// one fused FlowStep forward call (nfk_fused2.hip)
struct FusedCall {
  nfk_step2d s;
  const float* x; int ldx;
  float* zsave;          // [M][C]   mixed z (training) or null
  void* h1save;          // [M][hid] bf16 or null
  void* h2save;          // [M][hid] bf16 or null
  float* hsave;          // [M][Cout] or null
  unsigned char* masksave;  // [2][M][hid/8] ReLU masks of h1 / h2 (bit c%8 of byte c/8), with h1save / h2save, or null
  float* zout;           // [M][C]
  float* ldpart;         // [B][tiles per sample] partial log-det of this step
  float* ws;             // fused2_ws_floats(s) floats: partial conv3 slabs of a hidden-split step (else unused)
  unsigned long long* stamps;  // diagnostic: [grid][16] cycle stamps (null in production)
  int split_ways;              // hidden split requested by the caller: 4, 2 or 0 (none); see nfk_step2d_fwd_fused
};
// one fused FlowStep backward call (nfk_fusedb.hip): the dgrad chain g_a3 -> g_a2 -> g_a1 -> g_z1 contribution
struct FusedBwdCall {
  nfk_step2d s;
  const float* ga3;      // [M][Cout] gradient wrt conv3's raw output (coupling backward)
  const unsigned char* masks;  // [2][M][hid/8] ReLU masks of h1 / h2 left by the fused forward (FusedCall.masksave)
  void* ga2;             // [M][hid] bf16 out: operand of conv2's weight gradient
  void* ga1;             // [M][hid] bf16 out: operand of conv1's weight gradient
  float* d1;             // [M][C/2] out: dgrad1 (added to g_z1 by the channel-mix backward)
  void* x3save;          // [M][fusedb_k3p(s)] bf16 out (or null): dgrad3's im2col rows = P operand of conv3's weight gradient
  unsigned long long* stamps;  // diagnostic: [grid][16] cycle stamps (null in production)
};
// per-channel sums behind the two ActNorm layers' gradients, over a range of pixels (a 4th body of the triple launch):
//   col[part][0][c] = sum_m ga[m][c] h[m][c] / es[c]   (dlogs),   col[part][1][c] = sum_m ga[m][c]   (dbias)
struct ActnormSumArgs {
  const void *ga2, *h2, *ga1, *h1;   // [M][hid] bf16
  const float *es2, *es1;            // [hid]
  float *col2, *col1;                // [nparts][2][hid]
  int M, hid, nparts;
};
int actnorm_sum_parts(int M);
int fusedb_grid(const nfk_step2d& s);    // workgroups (= partial rows in col1 / col2) of the fused backward, 0 = not supported
int fusedb_launch(const FusedBwdCall& c, hipStream_t st);
int fusedb_k3p(const nfk_step2d& s);     // row length of x3save: 9 * Cout rounded up to 64
int fused2_k1p(const nfk_step2d& s);     // row length of the im2col rows the fused forward leaves behind the masks
// the three weight-gradient GEMMs of a step (block.4, .2, .0) in ONE launch; NFK_ERR_ARG: shapes not instantiated
int triple_wgrad_bf16(WgradArgs w3, int S3, WgradArgs w2, int S2, WgradArgs w1, int S1, const ActnormSumArgs& an, hipStream_t st,
                      double alg_nj = 0.0);
void triple_wgrad_debug_skip(int mask);
int fused2_tiles(const nfk_step2d& s);   // log-det partial slots per sample, 0 = shape not supported
void fused2_set_half_tiles(int on);
void fusedb_set_small_tiles(int on);
void tab_set_resident(int on);
int fused2_launch(const FusedCall& c, hipStream_t st);
size_t fused2_ws_floats(const nfk_step2d& s);

void set_tuning(int v);
void prof_begin();
void prof_end(double* ms, long long* launches, double* flops, double* bytes);
int padded_k(int K);
int conv_splitk(int M, int N, int K, int epi);
int conv_mblocks(int M, int N);
int conv_gemm(ConvGemmArgs a, hipStream_t st);
int conv_gemm_bf16(ConvGemmArgs a, hipStream_t st);
int padded_k_bf16(int K);
int conv_splitk_bf16(int M, int N, int K, int epi);
int conv_mblocks_bf16(int M, int N, int K);
void set_tuning_bf16(int v);
int wgrad_splits_bf16(int M, int NI, int J);
int wgrad3_splits_bf16(int M, int NI, int Cq, int W);
int wgrad_gemm_bf16(WgradArgs a, int S, hipStream_t st);
int dual_dgrad_wgrad_bf16(ConvGemmArgs d, WgradArgs w, int SW, hipStream_t st);
bool narrow_halo_dgrad_active(int bf16, int M, int H, int W, int N, int Cin);   // conv1 dgrad writes ONE slab then  // NFK_ERR_ARG: pair not instantiated
int repack_batch_bf16(const nfk_repack_desc* descs, int n, hipStream_t st);
void prof_scope_begin(int cls, double flops, double bytes, hipStream_t st);
void prof_scope_end(int cls, hipStream_t st);
int wgrad_splits(int M, int NI, int J);
int wgrad_gemm(WgradArgs a, int S, hipStream_t st);
int repack_weights(const float* W, int Cout, int Cin, int taps, float* Wt, float* Wd, hipStream_t st);
int repack_batch(const nfk_repack_desc* descs, int n, hipStream_t st);
int wgrad_finalize(const float* slabs, int S, long slab_stride, int Cout, int Cin, int taps, float* gradW,
                   hipStream_t st);
int launch_prep(const PrepDesc* descs_dev, int nsteps, int want_inverse, hipStream_t st);
int launch_prep_repack(const PrepDesc* descs_dev, int nsteps, int want_inverse, const nfk_repack_desc* rf32, int n_f32,
                       const nfk_repack_desc* rbf16, int n_bf16, hipStream_t st);
int launch_prep_bwd(const PrepDesc* descs_dev, const PrepGrad* grads_dev, int nsteps, const float* gld, int B,
                    int any_dense, hipStream_t st);
int launch_prep_inv_bwd(const PrepDesc* descs_dev, const PrepGrad* grads_dev, int nsteps, const float* gld, int B,
                        hipStream_t st);
int launch_chanmix(const ChanMixArgs& a, hipStream_t st);
int chanmix_parts(int M, int C);
int launch_coupling_fwd(const CouplingArgs& a, int B, hipStream_t st);
int launch_coupling_inv(const CouplingArgs& a, int B, hipStream_t st);
int launch_coupling_bwd(const CouplingBwdArgs& a, int B, hipStream_t st);
int coupling_bwd_tiles(int B, int HW);
int launch_coupling_inv_bwd(const CouplingBwdArgs& a, int B, hipStream_t st);
int launch_squeeze_nchw(float* x, const float* noise, float* out, int B, int C, int H, int W, hipStream_t st);
int launch_squeeze_nhwc(const float* x, int ldx, float* out, int B, int C, int H, int W, hipStream_t st);
int launch_unsqueeze(const float* y, const float* y2, float* out, int B, int C, int H, int W, int to_nchw, hipStream_t st);
int launch_split_fwd(const SplitArgs& a, int B, hipStream_t st);
int launch_split_inv(const SplitArgs& a, int B, hipStream_t st);
int launch_split_bwd(const SplitBwdArgs& a, int B, hipStream_t st);
int launch_add_slabs(const float* g, int ldg, const float* gext, int ldge, const float* slab, int S,
                     long slab_stride, float* out, int ldo, long M, int Ch, hipStream_t st);
// nfk_split.hip: Split2d's conv + prior in one launch per direction (fp32, vector ALUs)
int split2d_fused_tiles(const nfk_split2d& s);   // 0: geometry not covered
int split2d_fused_tps(const nfk_split2d& s);
int split2d_fused_fwd(const nfk_split2d& s, const float* x, const float* w, float* z1out, float* hsave, float* ldpart,
                      hipStream_t st);
int split2d_fused_bwd(const nfk_split2d& s, const float* x, const float* h, const float* w, const float* gz1, const float* gext,
                      const float* gld, float* gx, float* part, float* wslab, hipStream_t st);
int launch_prior_fwd(const PriorArgs& a, int B, hipStream_t st);
int launch_prior_bwd(const PriorBwdArgs& a, int B, hipStream_t st);
int launch_small_grads(const SmallGradArgs& a, hipStream_t st);
// three weight-gradient slab reductions (S[k] == 0: none) + the small per-channel gradients in one launch
struct StepTailArgs {
  const float* slabs[3]; float* gradW[3];
  long stride[3];
  int S[3], Cout[3], Cin[3], taps[3];
  // layout of a slab: mode 0: [Cout][ldj] with column tap * Cin + ci (ldj = 0: taps * Cin);
  //                   mode 1: [taps * Cout -> rows (taps - 1 - tap) * Cout + co][ldj] with column ci (conv3 through dgrad3's im2col)
  int mode[3], ldj[3];
  int blocks[3], sg_blocks;   // filled by the launcher
};
int launch_step_tail(const SmallGradArgs& a, StepTailArgs f, hipStream_t st);
// the same plus the step's chanmix backward in ONE launch (NFK_ERR_ARG: chanmix tile does not fit 64 KiB of LDS)
// `next` (optional): coupling backward of the step that follows in backward order, fused into the channel-mix workgroups;
// its partial rows are then one per channel-mix workgroup (chanmix_parts(M, C)), not one per sample
int launch_tail_chanmix(const SmallGradArgs& a, StepTailArgs f, const ChanMixArgs& m, const CouplingBwdArgs* next,
                        hipStream_t st);
// the step's three weight-gradient GEMMs (all operands pixel-major bf16: the fused backward's) + everything of
// launch_tail_chanmix in ONE launch; `f` / the col1, col2 part of `a` describe the PREVIOUS step's slabs and partial rows
int wgrad_tail_bf16(WgradArgs w3, int S3, WgradArgs w2, int S2, WgradArgs w1, int S1, const ActnormSumArgs& an,
                    const SmallGradArgs& a, StepTailArgs f, const ChanMixArgs& m, const CouplingBwdArgs* next, hipStream_t st,
                    double alg_nj);


// ---- level mode (nfk_level2d_bwd) ----
constexpr int NFK_LEVEL_MAX_SPLITS = 4;
constexpr int NFK_LEVEL_ROWS = 1024;   // partial rows / slabs a step of a level may leave (= workgroups of its launch)
struct LevelWgradArgs {
  const nfk_step2d* steps; const nfk_level_grads* grads;   // device arrays
  int k_lo, k_hi;                                          // steps walked
  int B, H, W, M, hid, Cout, Ch, K3p, K1p;
  int S, nb3, nb2, nb1, nparts, per_step;                  // (nb*, nparts, per_step: filled by the launcher)
  const float* ga; long ga_stride, o_ga2, o_ga1, o_x3;     // per-step gradient blocks (floats)
  const void *h1, *h2; long h_stride;                      // bf16 elements
  const unsigned char* masks; long mask_stride, o_x1;      // bytes
  float* wslab; long wslab_stride, o_s2, o_s1, o_col2, o_col1;   // per step: slab3 | slab2 | slab1 | col2 | col1 (floats)
  const float* tailpart; int rows_top, rows, k_top;        // [nsteps][NFK_LEVEL_ROWS][2][Cout]; partial rows of step k_top / the others
};
int level_wgrad_tiles(int hid, int K3p, int K1p);
int level_wgrad_splits(int M, int hid, int K3p, int K1p, int nwalk);
int level_wgrad_launch(LevelWgradArgs a, hipStream_t st, hipStream_t st_tail);
// the dgrad chain + channel-mix backward + coupling backward of the step below, for steps k_hi .. k_lo of a level
struct FusedLevelCall {
  nfk_level_bwd lv;
  nfk_step2d first;
  float* gz[2]; float* tailpart; float* ga; long ga_stride, o_ga2, o_ga1, o_x3;
  unsigned long long* stamps;   // diagnostic: [grid][16] cycle stamps of the last step of the last launch (null in production)
  int run_hi, run_lo;           // the steps to launch now (a sub-range of lv.k_hi .. lv.k_lo, which define the walk)
};
int fusedb_level_ok(const nfk_step2d& s);
int fusedb_level_multi_ok(const nfk_step2d& s);      // 1: level mode covers this geometry; 2: also as ONE launch for all steps
int fusedb_level_grid(const nfk_step2d& s);    // workgroups of a step launch (what the one-launch form uses may differ: see fusedb_level_launch)
int fusedb_level_launch(const FusedLevelCall& c, hipStream_t st, int* grid_used);
}  // namespace nfk
