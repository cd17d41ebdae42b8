// nfk tabular kernels: one fused kernel per direction for the reference's 1-D FlowStep
// (models/flows.py:142-202 with is_1d: ActNorm1d -> x @ W -> 6-layer MLP coupling net
// get_block_1d models/flows.py:37-52 -> affine/additive tail -> per-row log-det).
//
// A workgroup owns a tile of TR rows; every layer's activations for the tile live in LDS, the
// current layer's weights are staged (transposed) in LDS, thread = (row, 4 outputs).  Nothing but
// the step input and output touches HBM.  The backward kernel recomputes the tile's forward into
// LDS (cheaper than saving 7 activations per row), back-propagates layer by layer and
// accumulates weight-gradient partial sums in LDS; one slab per workgroup is written and reduced
// in fixed order by tab_reduce_kernel (deterministic, no atomics).
#include "nfk_internal.h"

namespace nfk {

using TabStep = nfk_tab_step;

__host__ __device__ static inline int pad4(int n) { return (n + 3) & ~3; }
__host__ __device__ static inline int tab_rows(int hid, int d) {
  int m = hid > d ? hid : d;
  const int g4 = (m + 3) / 4;
  int tr = 64;
  while (tr > 1 && tr * g4 > 256) tr >>= 1;
  return tr;
}
// MADE mode (nfk_tab_step.made): the block is autoregressive instead of a coupling -- the MLP (whose weights the caller has
// multiplied with the autoregressive masks) sees ALL d features and returns (mu_i, alpha_i) for every feature, interleaved
// like the coupling's (shift, scale);  u_i = (y_i - mu_i) exp(-alpha_i),  log|det| = -sum_i alpha_i  (MAF, Papamakarios et
// al. 2017, eq. 3-5; the reference repository contains no such code: SURVEY.md section 0 F1).
__host__ __device__ static inline int tab_nfeat(const TabStep& s) { return s.made ? s.d : s.d / 2; }   // features fed to the MLP
__host__ __device__ static inline int tab_nout(const TabStep& s) {
  if (s.made) return 2 * s.d;
  const int C2 = s.d - s.d / 2;
  return s.additive ? C2 : 2 * C2;
}
__host__ __device__ static inline int tab_nin0(const TabStep& s) { return tab_nfeat(s) + s.ncond; }
__host__ __device__ static inline int tab_layer_in(const TabStep& s, int l) { return l == 0 ? tab_nin0(s) : s.hid; }
__host__ __device__ static inline int tab_layer_out(const TabStep& s, int l) { return l == 5 ? tab_nout(s) : s.hid; }
__host__ __device__ static inline int tab_param_floats(const TabStep& s) {
  int t = 0;
  for (int l = 0; l < 6; ++l) t += tab_layer_out(s, l) * (tab_layer_in(s, l) + 1);
  return t;
}

__host__ __device__ static inline int tab_param_off(const TabStep& s, int l) {
  int t = 0;
  for (int k = 0; k < l; ++k) t += tab_layer_out(s, k) * (tab_layer_in(s, k) + 1);
  return t;
}

struct TabLds {
  int TR, XS, HS, OS, IS, GS;
  float *X, *Y, *IN0, *H0, *O, *GA, *GB, *GY, *Wl, *Pl, *PM;
  int HST;   // floats between the five hidden-activation buffers: H(k) = H0 + k * HST.  (No arrays of pointers in here: indexed
             // with a run-time layer number they put the whole struct into scratch memory, a round trip per use.)
  __host__ __device__ float* H(int k) const { return H0 + k * HST; }
  // resident mode (round 3): every weight matrix of the step stays in LDS for the life of the workgroup, in the forward
  // layout [inputs][pad4(outputs) + 4] (slot 0: the channel mix Wf, 1..6: the MLP's layers), with its bias vector
  float* WR0;   // WR(k) = WR0 + tab_wr_off(s, k), BR(k) behind it (tab_wr / tab_br)
  int res;
};

__host__ __device__ static inline int tab_res_stride(int nout) { return pad4(nout) + 4; }   // rows 16-byte aligned, banks spread
__host__ __device__ static inline int tab_w_in(const TabStep& s, int k) { return k == 0 ? s.d : tab_layer_in(s, k - 1); }
__host__ __device__ static inline int tab_w_out(const TabStep& s, int k) { return k == 0 ? s.d : tab_layer_out(s, k - 1); }
// floats in front of resident matrix k (k = 7: all of them): matrix [inputs][stride], then its bias vector, each padded to 4
__host__ __device__ static inline int tab_wr_off(const TabStep& s, int k) {
  int t = 0;
  for (int j = 0; j < k; ++j) t += pad4(tab_w_in(s, j) * tab_res_stride(tab_w_out(s, j))) + pad4(tab_w_out(s, j));
  return t;
}
__host__ __device__ static inline float* tab_wr(const TabStep& s, const TabLds& L, int k) { return L.WR0 + tab_wr_off(s, k); }
__host__ __device__ static inline float* tab_br(const TabStep& s, const TabLds& L, int k) {
  return L.WR0 + tab_wr_off(s, k) + pad4(tab_w_in(s, k) * tab_res_stride(tab_w_out(s, k)));
}

__host__ __device__ __attribute__((always_inline)) static inline size_t tab_layout(const TabStep& s, bool bwd, TabLds& Lr, float* base, bool res = false, int tr = 0) {
  TabLds* const L = &Lr;   // (a reference: a null test on a private-memory address pins the struct in scratch memory)
  const int d = s.d, hid = s.hid, nout = tab_nout(s), nin0 = tab_nin0(s);
  const int TR = tr > 0 ? tr : tab_rows(hid, d);   // (tr: the resident forward's taller tiles)
  int mx = d > hid ? d : hid;
  if (nout > mx) mx = nout;
  if (nin0 > mx) mx = nin0;
  size_t o = 0;
  auto take = [&](size_t n) { size_t r = o; o += (n + 3) & ~size_t(3); return r; };
  // (resident backward: gradient rows 16-byte aligned -- its transposed layers read four of them as float4 beside a weight row)
  // row strides: odd (rows p, p + 1, ... of a tile on different banks: lanes of a wave differ in p) -- "n + 1" alone left d = 63
  // with 64-float rows, every row on the same bank
  const int XS = (d + 1) | 1, HS = (hid + 1) | 1, OS = (nout + 1) | 1, IS = (nin0 + 1) | 1,
            GS = (res && bwd) ? pad4(mx) + 4 : ((mx + 1) | 1);
  size_t oX = take((size_t)TR * XS), oY = take((size_t)TR * XS), oI = take((size_t)TR * IS);
  const size_t hst = ((size_t)TR * HS + 3) & ~size_t(3);
  size_t oH = take(5 * hst);
  size_t oO = take((size_t)TR * OS);
  // staged weights of one dense layer [inputs][pad4(outputs)]: the largest of the layers' shapes (and their transposes in
  // the backward), not mx x mx -- with 2 d outputs (MADE, d = 63) that alone would be 64 KiB
  size_t wmax = (size_t)d * pad4(d);
  {
    const size_t c[5] = {(size_t)nin0 * pad4(hid), (size_t)hid * pad4(hid), (size_t)hid * pad4(nout),
                         (size_t)hid * pad4(nin0), (size_t)nout * pad4(hid)};
#pragma unroll
    for (int i = 0; i < 5; ++i) if (c[i] > wmax) wmax = c[i];
  }
  size_t oW = res ? 0 : take(wmax);
  size_t oGA = 0, oGB = 0, oGY = 0, oP = 0, oPM = 0;
  if (bwd) {
    oGA = take((size_t)TR * GS); oGB = take((size_t)TR * GS); oGY = take((size_t)TR * (res ? GS : XS));
    if (!res) { oP = take((size_t)tab_param_floats(s)); oPM = take((size_t)d * d + d); }   // (resident: in registers)
  }
  size_t oWR = 0;
  if (res) {
    oWR = o;
    o += tab_wr_off(s, 7);
  }
  {
    L->TR = TR; L->XS = XS; L->HS = HS; L->OS = OS; L->IS = IS; L->GS = GS;
    L->X = base + oX; L->Y = base + oY; L->IN0 = base + oI;
    L->H0 = base + oH; L->HST = (int)hst;
    L->O = base + oO; L->Wl = base + oW;
    L->GA = base + oGA; L->GB = base + oGB; L->GY = base + oGY; L->Pl = base + oP; L->PM = base + oPM;
    L->res = res ? 1 : 0;
    L->WR0 = base + oWR;
  }
  return o * sizeof(float);
}

// out[p][o] = act(sum_c in[p][c] * Wm(o,c) + b[o]);  Wm(o,c) = W[o*nin + c]  (trans=0, nn.Linear)
//                                                   or W[c*nout + o]         (trans=1: uses W^T)
// act: 0 none, 1 relu, 2 tanh.  Block-wide; contains barriers.
__device__ void tab_dense(const float* in, int ldi, int nin, const float* __restrict__ W, const float* __restrict__ b,
                          float* out, int ldo, int nout, int act, int trans, float* Wl, int TR) {
  const int np = pad4(nout);
  __syncthreads();
  for (int idx = threadIdx.x; idx < nin * nout; idx += blockDim.x) {
    if (trans) { const int c = idx / nout, o = idx - c * nout; Wl[c * np + o] = W[idx]; }
    else { const int o = idx / nin, c = idx - o * nin; Wl[c * np + o] = W[idx]; }
  }
  __syncthreads();
  const int G4 = np / 4;
  for (int t = threadIdx.x; t < TR * G4; t += blockDim.x) {
    const int p = t / G4, o = (t - p * G4) * 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (b)
#pragma unroll
      for (int r = 0; r < 4; ++r) if (o + r < nout) acc[r] = b[o + r];
    const float* xr = in + p * ldi;
    for (int c = 0; c < nin; ++c) {
      const float xv = xr[c];
      const f32x4 w = *reinterpret_cast<const f32x4*>(Wl + c * np + o);
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] += w[r] * xv;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (o + r < nout) {
        float v = acc[r];
        if (act == 1) v = fmaxf(v, 0.f);
        else if (act == 2) v = tanhf(v);
        out[p * ldo + o + r] = v;
      }
  }
  __syncthreads();
}

// ---- resident mode: weights staged ONCE per workgroup.  (Per tile and layer the staging above is a round trip to L2 for
// ~1 K floats in front of 16 rows x 32 outputs of arithmetic: at d = 63 it was most of the forward's time and, fourteen times
// per tile, of the backward's.)
__device__ __forceinline__ void tab_stage_resident(const TabStep& s, const TabLds& L) {
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    const int nin = tab_w_in(s, k), nout = tab_w_out(s, k), rs = tab_res_stride(nout);
    const float* __restrict__ W = k == 0 ? s.Wf : s.W[k > 0 ? k - 1 : 0];
    const float* __restrict__ b = k == 0 ? s.cf : s.b[k > 0 ? k - 1 : 0];
    float* wr = tab_wr(s, L, k);
    float* br = tab_br(s, L, k);
    for (int idx = threadIdx.x; idx < nin * rs; idx += blockDim.x) {
      const int c = idx / rs, o = idx - c * rs;
      wr[idx] = o < nout ? W[o * nin + c] : 0.f;      // nn.Linear [out][in] -> [in][out + pad], pad columns zero
    }
    for (int o = threadIdx.x; o < nout; o += blockDim.x) br[o] = b[o];
  }
  __syncthreads();
}
// ---- resident mode's contractions run on the matrix cores (exact-f32 v_mfma_f32_16x16x4_f32; tiles of 16 rows: TR % 16 == 0).
// The vector-ALU loops of tab_dense read 20 bytes of LDS per four multiply-adds (one weight quad + one activation per lane); the
// MFMA operand layout reads 8 (one weight, one activation per lane for 4 x 16 x 16 / 64 = 16 multiply-adds per lane).
// Operand a of mfma(a, b, acc): lane l holds a[row = l & 15][k = l >> 4]; b likewise; acc[r] = out[a row 4 (l >> 4) + r][b row l & 15].
// Here a = the weight side (16 output columns), b = the row side (16 rows of the tile).  A wave takes whole 16 x 16 tiles.
typedef float tab_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void tab_dense_mfma(const float* in, int ldi, int nin, const float* Wr, const float* br, float* out,
                                               int ldo, int nout, int act, int TR) {
  const int rs = tab_res_stride(nout), wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r16 = lane & 15, q = lane >> 4;
  const int MT = TR >> 4, NT = (nout + 15) >> 4;
  for (int t = wave; t < MT * NT; t += 4) {
    const int mt = t % MT, nt = t / MT;
    const int p = mt * 16 + r16, oa = nt * 16 + r16, ob = nt * 16 + 4 * q;
    tab_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (br)
#pragma unroll
      for (int r = 0; r < 4; ++r) if (ob + r < nout) acc[r] = br[ob + r];
    const float* wp = Wr + (oa < nout ? oa : 0);
    const float* xp = in + p * ldi;
    const bool wok = oa < nout;
    for (int k0 = 0; k0 < nin; k0 += 16) {   // four k-steps per trip: eight loads in flight, then four MFMAs
      float a[4], b[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = k0 + 4 * j + q;
        const bool ok = c < nin;
        a[j] = (ok && wok) ? wp[c * rs] : 0.f;
        b[j] = ok ? xp[c] : 0.f;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (ob + r < nout) {
        float v = acc[r];
        if (act == 1) v = fmaxf(v, 0.f);
        else if (act == 2) v = tanhf(v);
        out[p * ldo + ob + r] = v;
      }
  }
  __syncthreads();
}
// out[p][c] = sum_o g[p][o] Wr[c][o]  (o < nred; c < nout_t): a = weight rows c, b = gradient rows p, k = o
__device__ __forceinline__ void tab_dense_t_mfma(const float* g, int ldg, int nred, const float* Wr, float* out, int ldo,
                                                 int nout_t, int TR) {
  const int rs = tab_res_stride(nred), wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r16 = lane & 15, q = lane >> 4;
  const int MT = TR >> 4, NT = (nout_t + 15) >> 4;
  for (int t = wave; t < MT * NT; t += 4) {
    const int mt = t % MT, nt = t / MT;
    const int p = mt * 16 + r16, ca = nt * 16 + r16, cb = nt * 16 + 4 * q;
    tab_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const bool wok = ca < nout_t;
    const float* wp = Wr + (wok ? ca : 0) * rs;
    const float* gp = g + p * ldg;
    for (int k0 = 0; k0 < nred; k0 += 16) {
      float a[4], b[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int o = k0 + 4 * j + q;
        const bool ok = o < nred;
        a[j] = (ok && wok) ? wp[o] : 0.f;
        b[j] = ok ? gp[o] : 0.f;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (cb + r < nout_t) out[p * ldo + cb + r] = acc[r];
  }
  __syncthreads();
}
// acc[j] (tile t = wave + 4 j of the layer's OT x CT tiles of 16 x 16) += sum_p g[p][o] in[p][c];  bacc += sum_p g[p][o = thread].
// a = gradient columns o, b = input columns c, k = the tile's rows.  The sums stay in registers for the life of the
// workgroup (at most TAB_WG_TILES tiles per wave and layer: tab_resident_ok), written out once by tab_wgrad_store.
constexpr int TAB_WG_TILES = 4;
__device__ __forceinline__ void tab_wgrad_mfma(const float* g, int ldg, const float* in, int ldi, int nin, int nout,
                                               tab_f32x4 (&acc)[TAB_WG_TILES], float& bacc, int TR) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r16 = lane & 15, q = lane >> 4;
  const int OT = (nout + 15) >> 4, CT = (nin + 15) >> 4;
#pragma unroll
  for (int j = 0; j < TAB_WG_TILES; ++j) {
    const int t = wave + 4 * j;
    if (t < OT * CT) {
      const int ct = t % CT, ot = t / CT;
      const int oa = ot * 16 + r16, cb = ct * 16 + r16;
      const bool aok = oa < nout, bok = cb < nin;
      const float* gp = g + (aok ? oa : 0);
      const float* ip = in + (bok ? cb : 0);
      tab_f32x4 v = acc[j];
      for (int k0 = 0; k0 < TR; k0 += 16) {   // (TR % 16 == 0)
        float a[4], b[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int p = k0 + 4 * e + q;
          a[e] = aok ? gp[p * ldg] : 0.f;
          b[e] = bok ? ip[p * ldi] : 0.f;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b[e], v, 0, 0, 0);
      }
      acc[j] = v;
    }
  }
  if ((int)threadIdx.x < nout) {
    float sacc = 0.f;
#pragma unroll 4
    for (int p = 0; p < TR; ++p) sacc += g[p * ldg + threadIdx.x];
    bacc += sacc;
  }
}
// the sums of one layer -> the workgroup's slab:  P[o * ldp + c],  Pb[o * bstride]
__device__ __forceinline__ void tab_wgrad_store(int nin, int nout, const tab_f32x4 (&acc)[TAB_WG_TILES], float bacc, float* P,
                                                int ldp, float* Pb, int bstride) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r16 = lane & 15, q = lane >> 4;
  const int OT = (nout + 15) >> 4, CT = (nin + 15) >> 4;
#pragma unroll
  for (int j = 0; j < TAB_WG_TILES; ++j) {
    const int t = wave + 4 * j;
    if (t < OT * CT) {
      const int ct = t % CT, ot = t / CT, cb = ct * 16 + r16, ob = ot * 16 + 4 * q;
      if (cb < nin)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (ob + r < nout) P[(ob + r) * ldp + cb] = acc[j][r];
    }
  }
  if ((int)threadIdx.x < nout) Pb[threadIdx.x * bstride] = bacc;
}

// `staged`: the caller has put the tile's rows into L.X already (the resident forward requests them one tile ahead)
__device__ __forceinline__ void tab_forward_tile(const TabStep& s, const TabLds& L, const float* x, const float* cond, int row0,
                                                 bool staged = false) {
  const int d = s.d, TR = L.TR;
  if (!staged)
    for (int idx = threadIdx.x; idx < TR * d; idx += blockDim.x) {
      const int p = idx / d, c = idx - p * d, m = row0 + p;
      L.X[p * L.XS + c] = m < s.B ? x[(long)m * d + c] : 0.f;
    }
  if (L.res) { __syncthreads(); tab_dense_mfma(L.X, L.XS, d, tab_wr(s, L, 0), tab_br(s, L, 0), L.Y, L.XS, d, 0, TR); }
  else tab_dense(L.X, L.XS, d, s.Wf, s.cf, L.Y, L.XS, d, 0, 0, L.Wl, TR);
  const int nin0 = tab_nin0(s), nf = tab_nfeat(s);
  for (int idx = threadIdx.x; idx < TR * nin0; idx += blockDim.x) {
    const int p = idx / nin0, c = idx - p * nin0, m = row0 + p;
    float v;
    if (c < nf) v = L.Y[p * L.XS + c];
    else v = m < s.B ? cond[(long)m * s.ncond + (c - nf)] : 0.f;
    L.IN0[p * L.IS + c] = v;
  }
  const float* in = L.IN0;
  int ldi = L.IS;
  if (L.res) {
    __syncthreads();
#pragma unroll
    for (int l = 0; l < 5; ++l) {
      tab_dense_mfma(in, ldi, tab_layer_in(s, l), tab_wr(s, L, l + 1), tab_br(s, L, l + 1), L.H(l), L.HS, s.hid, l == 4 ? 2 : 1, TR);
      in = L.H(l);
      ldi = L.HS;
    }
    tab_dense_mfma(L.H(4), L.HS, s.hid, tab_wr(s, L, 6), tab_br(s, L, 6), L.O, L.OS, tab_nout(s), 0, TR);
    return;
  }
  for (int l = 0; l < 5; ++l) {
    tab_dense(in, ldi, tab_layer_in(s, l), s.W[l], s.b[l], L.H(l), L.HS, s.hid, l == 4 ? 2 : 1, 0, L.Wl, TR);
    in = L.H(l);
    ldi = L.HS;
  }
  tab_dense(L.H(4), L.HS, s.hid, s.W[5], s.b[5], L.O, L.OS, tab_nout(s), 0, 0, L.Wl, TR);
}

__global__ void __launch_bounds__(256) tab_fwd_kernel(TabStep s, const float* __restrict__ x,
                                                      const float* __restrict__ cond, const float* __restrict__ ld_in,
                                                      float* __restrict__ zout, float* __restrict__ ld_out, int res) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  TabLds L;
  tab_layout(s, false, L, smem, res != 0, res);   // res = rows per tile in resident mode
  if (res) tab_stage_resident(s, L);
  const int d = s.d, Ch = d / 2, C2 = d - Ch, TR = L.TR;
  const int ntiles = (s.B + TR - 1) / TR;
  // resident mode: the NEXT tile's rows are requested (into registers) before this tile's chain of short phases starts
  constexpr int XPF = 4;
  const bool pf = res != 0 && TR * d <= XPF * 256;
  float xpf[XPF];
  auto request = [&](int tile) {
#pragma unroll
    for (int e = 0; e < XPF; ++e) {
      const int idx = threadIdx.x + e * 256;
      const int p = idx / d, c = idx - p * d, m = tile * TR + p;
      xpf[e] = (idx < TR * d && tile < ntiles && m < s.B) ? x[(long)m * d + c] : 0.f;
    }
  };
  if (pf) request(blockIdx.x);
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int row0 = tile * TR;
    if (pf) {
#pragma unroll
      for (int e = 0; e < XPF; ++e) {
        const int idx = threadIdx.x + e * 256;
        const int p = idx / d, c = idx - p * d;
        if (idx < TR * d) L.X[p * L.XS + c] = xpf[e];
      }
      request(tile + gridDim.x);
    }
    tab_forward_tile(s, L, x, cond, row0, pf);
    if (s.made) {   // u = (y - mu) exp(-alpha) on every feature; log-det terms -alpha into X
      for (int idx = threadIdx.x; idx < TR * d; idx += blockDim.x) {
        const int p = idx / d, j = idx - p * d, m = row0 + p;
        const float mu = L.O[p * L.OS + 2 * j], al = L.O[p * L.OS + 2 * j + 1];
        L.X[p * L.XS + j] = -al;
        if (m < s.B) zout[(long)m * d + j] = (L.Y[p * L.XS + j] - mu) * expf(-al);
      }
      __syncthreads();
      if (ld_out)
        for (int p = threadIdx.x; p < TR; p += blockDim.x) {
          const int m = row0 + p;
          if (m < s.B) {
            float t = 0.f;
            for (int j = 0; j < d; ++j) t += L.X[p * L.XS + j];
            ld_out[m] = ld_in[m] + s.dld[0] + t;
          }
        }
      __syncthreads();
      continue;
    }
    // coupling tail; log-scale terms go to GA-less scratch: reuse X (no longer needed) as [TR][XS]
    for (int idx = threadIdx.x; idx < TR * C2; idx += blockDim.x) {
      const int p = idx / C2, j = idx - p * C2, m = row0 + p;
      const float z2 = L.Y[p * L.XS + Ch + j];
      float out, lt = 0.f;
      if (s.additive) out = z2 + L.O[p * L.OS + j];
      else {
        const float t = L.O[p * L.OS + 2 * j + 1] + 2.f;
        out = (z2 + L.O[p * L.OS + 2 * j]) * sigmoidf(t);
        lt = log_sigmoidf(t);
      }
      L.X[p * L.XS + j] = lt;
      if (m < s.B) zout[(long)m * d + Ch + j] = out;
    }
    for (int idx = threadIdx.x; idx < TR * Ch; idx += blockDim.x) {
      const int p = idx / Ch, j = idx - p * Ch, m = row0 + p;
      if (m < s.B) zout[(long)m * d + j] = L.Y[p * L.XS + j];
    }
    __syncthreads();
    if (ld_out)
      for (int p = threadIdx.x; p < TR; p += blockDim.x) {
        const int m = row0 + p;
        if (m < s.B) {
          float t = 0.f;
          for (int j = 0; j < C2; ++j) t += L.X[p * L.XS + j];
          ld_out[m] = ld_in[m] + s.dld[0] + t;
        }
      }
    __syncthreads();
  }
}

// inverse: y (step output) -> x.  z1 = y[:Ch]; (shift, scale) = MLP(z1, cond); z2 = y2/scale - shift; x = Wi z + ci
__global__ void __launch_bounds__(256) tab_inv_kernel(TabStep s, const float* __restrict__ y,
                                                      const float* __restrict__ cond, const float* __restrict__ ld_in,
                                                      float* __restrict__ xout, float* __restrict__ ld_out) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  TabLds L;
  tab_layout(s, false, L, smem);
  const int d = s.d, Ch = d / 2, C2 = d - Ch, TR = L.TR, nin0 = tab_nin0(s);
  const int ntiles = (s.B + TR - 1) / TR;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int row0 = tile * TR;
    __syncthreads();
    for (int idx = threadIdx.x; idx < TR * d; idx += blockDim.x) {
      const int p = idx / d, c = idx - p * d, m = row0 + p;
      L.Y[p * L.XS + c] = m < s.B ? y[(long)m * d + c] : 0.f;
    }
    __syncthreads();
    if (s.made) {
      // autoregressive inverse: y_i = u_i exp(alpha_i) + mu_i with (mu_i, alpha_i) functions of y_<i only (the masks):
      // d passes of the MLP over the tile, pass i fixes feature i.  U = the step output (kept in X), Y starts at zero
      for (int idx = threadIdx.x; idx < TR * d; idx += blockDim.x) {
        const int p = idx / d, c = idx - p * d;
        L.X[p * L.XS + c] = L.Y[p * L.XS + c];
        L.Y[p * L.XS + c] = 0.f;
      }
      __syncthreads();
      for (int i = 0; i < d; ++i) {
        for (int idx = threadIdx.x; idx < TR * nin0; idx += blockDim.x) {
          const int p = idx / nin0, c = idx - p * nin0, m = row0 + p;
          L.IN0[p * L.IS + c] = c < d ? L.Y[p * L.XS + c] : (m < s.B ? cond[(long)m * s.ncond + (c - d)] : 0.f);
        }
        const float* in = L.IN0;
        int ldi = L.IS;
        for (int l = 0; l < 5; ++l) {
          tab_dense(in, ldi, tab_layer_in(s, l), s.W[l], s.b[l], L.H(l), L.HS, s.hid, l == 4 ? 2 : 1, 0, L.Wl, TR);
          in = L.H(l);
          ldi = L.HS;
        }
        tab_dense(L.H(4), L.HS, s.hid, s.W[5], s.b[5], L.O, L.OS, tab_nout(s), 0, 0, L.Wl, TR);
        for (int p = threadIdx.x; p < TR; p += blockDim.x)
          L.Y[p * L.XS + i] = L.X[p * L.XS + i] * expf(L.O[p * L.OS + 2 * i + 1]) + L.O[p * L.OS + 2 * i];
        __syncthreads();
      }
      // after the last pass every alpha_i in O is final (alpha_i depends on y_<i, all fixed by then)
      if (ld_out)
        for (int p = threadIdx.x; p < TR; p += blockDim.x) {
          const int m = row0 + p;
          if (m < s.B) {
            float t = 0.f;
            for (int j = 0; j < d; ++j) t += L.O[p * L.OS + 2 * j + 1];
            ld_out[m] = ld_in[m] - s.dld[0] + t;   // forward added -sum alpha
          }
        }
      tab_dense(L.Y, L.XS, d, s.Wi, s.ci, L.X, L.XS, d, 0, 0, L.Wl, TR);
      for (int idx = threadIdx.x; idx < TR * d; idx += blockDim.x) {
        const int p = idx / d, c = idx - p * d, m = row0 + p;
        if (m < s.B) xout[(long)m * d + c] = L.X[p * L.XS + c];
      }
      continue;
    }
    for (int idx = threadIdx.x; idx < TR * nin0; idx += blockDim.x) {
      const int p = idx / nin0, c = idx - p * nin0, m = row0 + p;
      L.IN0[p * L.IS + c] = c < Ch ? L.Y[p * L.XS + c] : (m < s.B ? cond[(long)m * s.ncond + (c - Ch)] : 0.f);
    }
    const float* in = L.IN0;
    int ldi = L.IS;
    for (int l = 0; l < 5; ++l) {
      tab_dense(in, ldi, tab_layer_in(s, l), s.W[l], s.b[l], L.H(l), L.HS, s.hid, l == 4 ? 2 : 1, 0, L.Wl, TR);
      in = L.H(l);
      ldi = L.HS;
    }
    tab_dense(L.H(4), L.HS, s.hid, s.W[5], s.b[5], L.O, L.OS, tab_nout(s), 0, 0, L.Wl, TR);
    for (int idx = threadIdx.x; idx < TR * C2; idx += blockDim.x) {
      const int p = idx / C2, j = idx - p * C2;
      const float y2 = L.Y[p * L.XS + Ch + j];
      float out, lt = 0.f;
      if (s.additive) out = y2 - L.O[p * L.OS + j];
      else {
        const float t = L.O[p * L.OS + 2 * j + 1] + 2.f;
        out = y2 * (1.f + expf(-t)) - L.O[p * L.OS + 2 * j];
        lt = log_sigmoidf(t);
      }
      L.X[p * L.XS + j] = lt;            // log terms (X is free until the final mix)
      L.Y[p * L.XS + Ch + j] = out;      // z = (z1 | z2) assembled in place (same thread reads and writes it)
    }
    __syncthreads();
    if (ld_out)
      for (int p = threadIdx.x; p < TR; p += blockDim.x) {
        const int m = row0 + p;
        if (m < s.B) {
          float t = 0.f;
          for (int j = 0; j < C2; ++j) t += L.X[p * L.XS + j];
          ld_out[m] = ld_in[m] - s.dld[0] - t;
        }
      }
    tab_dense(L.Y, L.XS, d, s.Wi, s.ci, L.X, L.XS, d, 0, 0, L.Wl, TR);
    for (int idx = threadIdx.x; idx < TR * d; idx += blockDim.x) {
      const int p = idx / d, c = idx - p * d, m = row0 + p;
      if (m < s.B) xout[(long)m * d + c] = L.X[p * L.XS + c];
    }
  }
}

// accumulate  P[o*(nin+1) + c] += sum_p g[p][o] * in[p][c]   and   P[o*(nin+1) + nin] += sum_p g[p][o]
__device__ void tab_wgrad(const float* g, int ldg, const float* in, int ldi, int nin, int nout, float* P, int TR) {
  for (int idx = threadIdx.x; idx < nout * (nin + 1); idx += blockDim.x) {
    const int o = idx / (nin + 1), c = idx - o * (nin + 1);
    float sacc = 0.f;
    if (c < nin) {
#pragma unroll 4
      for (int p = 0; p < TR; ++p) sacc += g[p * ldg + o] * in[p * ldi + c];
    } else {
#pragma unroll 4
      for (int p = 0; p < TR; ++p) sacc += g[p * ldg + o];
    }
    P[idx] += sacc;
  }
}

// MLP backward (layers 5..0) for the tile whose forward activations (IN0, H[0..4]) are in LDS and whose gradient wrt the
// last layer's output is in L.GA.  Accumulates the weight-gradient partial sums into L.Pl; returns the gradient wrt the
// first layer's input IN0 = (z1 | cond) (in L.GA or L.GB).  Block-wide.
__device__ const float* tab_mlp_backward(const TabStep& s, const TabLds& L) {
  const int TR = L.TR;
  float* g = L.GA;
  float* gn = L.GB;
  for (int l = 5; l >= 0; --l) {
    const int nin = tab_layer_in(s, l), no = tab_layer_out(s, l);
    const float* in = l == 0 ? L.IN0 : L.H(l - 1);
    const int ldi = l == 0 ? L.IS : L.HS;
    tab_wgrad(g, L.GS, in, ldi, nin, no, L.Pl + tab_param_off(s, l), TR);
    // grad wrt this layer's input (post-activation of layer l-1)
    tab_dense(g, L.GS, no, s.W[l], nullptr, gn, L.GS, nin, 0, 1, L.Wl, TR);
    if (l > 0) {
      const int act = (l - 1) == 4 ? 2 : 1;
      for (int idx = threadIdx.x; idx < TR * nin; idx += blockDim.x) {
        const int p = idx / nin, c = idx - p * nin;
        const float h = L.H(l - 1)[p * L.HS + c];
        gn[p * L.GS + c] *= act == 1 ? (h > 0.f ? 1.f : 0.f) : (1.f - h * h);
      }
      __syncthreads();
    }
    float* t = g; g = gn; gn = t;
  }
  return g;
}

__global__ void __launch_bounds__(256) tab_bwd_kernel(TabStep s, const float* __restrict__ x,
                                                      const float* __restrict__ cond, const float* __restrict__ gout,
                                                      const float* __restrict__ gld, float* __restrict__ gx,
                                                      float* __restrict__ gcond, float* __restrict__ part,
                                                      float* __restrict__ part_mix) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  TabLds L;
  tab_layout(s, true, L, smem);
  const int d = s.d, Ch = d / 2, C2 = d - Ch, TR = L.TR, nin0 = tab_nin0(s);
  const int NP = tab_param_floats(s), NM = d * d + d;
  for (int i = threadIdx.x; i < NP; i += blockDim.x) L.Pl[i] = 0.f;
  for (int i = threadIdx.x; i < NM; i += blockDim.x) L.PM[i] = 0.f;
  const int ntiles = (s.B + TR - 1) / TR;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int row0 = tile * TR;
    tab_forward_tile(s, L, x, cond, row0);
    // ---- tail backward: grads wrt O (into GA) and wrt y2 (into GY second half)
    if (s.made)   // u = (y - mu) e^-alpha, ld += -alpha:  g_mu = -g_u e^-alpha, g_alpha = -g_u u - g_ld, g_y (direct) = g_u e^-alpha
      for (int idx = threadIdx.x; idx < TR * d; idx += blockDim.x) {
        const int p = idx / d, j = idx - p * d, m = row0 + p;
        const float gu = m < s.B ? gout[(long)m * d + j] : 0.f;
        const float gl = (m < s.B && gld) ? gld[m] : 0.f;
        const float mu = L.O[p * L.OS + 2 * j], al = L.O[p * L.OS + 2 * j + 1], e = expf(-al);
        const float u = (L.Y[p * L.XS + j] - mu) * e;
        L.GA[p * L.GS + 2 * j] = -gu * e;
        L.GA[p * L.GS + 2 * j + 1] = m < s.B ? -gu * u - gl : 0.f;
        L.GY[p * L.XS + j] = gu * e;
      }
    else
    for (int idx = threadIdx.x; idx < TR * C2; idx += blockDim.x) {
      const int p = idx / C2, j = idx - p * C2, m = row0 + p;
      const float g2 = m < s.B ? gout[(long)m * d + Ch + j] : 0.f;
      const float gl = (m < s.B && gld) ? gld[m] : 0.f;
      if (s.additive) {
        L.GA[p * L.GS + j] = g2;
        L.GY[p * L.XS + Ch + j] = g2;
      } else {
        const float shf = L.O[p * L.OS + 2 * j], sr = L.O[p * L.OS + 2 * j + 1];
        const float sc = sigmoidf(sr + 2.f), z2 = L.Y[p * L.XS + Ch + j];
        const float gshift = g2 * sc;
        L.GA[p * L.GS + 2 * j] = gshift;
        L.GA[p * L.GS + 2 * j + 1] = (m < s.B) ? g2 * (z2 + shf) * sc * (1.f - sc) + gl * (1.f - sc) : 0.f;
        L.GY[p * L.XS + Ch + j] = gshift;
      }
    }
    __syncthreads();
    const float* g = tab_mlp_backward(s, L);
    // g = grad wrt IN0 = (z1 | cond)   (MADE: (y | cond), added to the direct term of the tail)
    const int nf = tab_nfeat(s);
    for (int idx = threadIdx.x; idx < TR * nin0; idx += blockDim.x) {
      const int p = idx / nin0, c = idx - p * nin0, m = row0 + p;
      if (c < nf) {
        if (s.made) L.GY[p * L.XS + c] += g[p * L.GS + c];
        else L.GY[p * L.XS + c] = g[p * L.GS + c] + (m < s.B ? gout[(long)m * d + c] : 0.f);
      } else if (gcond && m < s.B) gcond[(long)m * s.ncond + (c - nf)] += g[p * L.GS + c];
    }
    __syncthreads();
    // ---- mix backward: gx = Wf^T gy ; partial (gWf, gcf)
    for (int idx = threadIdx.x; idx < NM; idx += blockDim.x) {
      float sacc = 0.f;
      if (idx < d * d) {
        const int o = idx / d, c = idx - o * d;
        for (int p = 0; p < TR; ++p) sacc += L.GY[p * L.XS + o] * L.X[p * L.XS + c];
      } else {
        const int o = idx - d * d;
        for (int p = 0; p < TR; ++p) sacc += L.GY[p * L.XS + o];
      }
      L.PM[idx] += sacc;
    }
    float* gn = (g == L.GA) ? L.GB : L.GA;   // the buffer the MLP backward is done with
    tab_dense(L.GY, L.XS, d, s.Wf, nullptr, gn, L.GS, d, 0, 1, L.Wl, TR);
    for (int idx = threadIdx.x; idx < TR * d; idx += blockDim.x) {
      const int p = idx / d, c = idx - p * d, m = row0 + p;
      if (m < s.B) gx[(long)m * d + c] = gn[p * L.GS + c];
    }
    __syncthreads();
  }
  __syncthreads();
  for (int i = threadIdx.x; i < NP; i += blockDim.x) part[(long)blockIdx.x * NP + i] = L.Pl[i];
  for (int i = threadIdx.x; i < NM; i += blockDim.x) part_mix[(long)blockIdx.x * NM + i] = L.PM[i];
}

// ---- resident backward (round 3).  As tab_bwd_kernel, with every weight matrix resident in LDS (tab_stage_resident), the
// transposed layers as row dot products on the same copies (tab_dense_res_t: no transposed staging), and the per-workgroup
// weight-gradient sums in LDS as before (thread t owns elements t, t + 256, ...: no barrier around them).  The layer loops are
// unrolled so that the LDS layout (TabLds: arrays of pointers) stays in registers: indexed with a run-time layer number it
// lived in scratch memory, and every use of a pointer was a round trip to it -- most of a tile's time.
__global__ void __launch_bounds__(256) tab_bwd_res_kernel(TabStep s, const float* __restrict__ x, const float* __restrict__ cond,
                                                          const float* __restrict__ gout, const float* __restrict__ gld,
                                                          float* __restrict__ gx, float* __restrict__ gcond,
                                                          float* __restrict__ part, float* __restrict__ part_mix, int tr) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  TabLds L;
  tab_layout(s, true, L, smem, true, tr);
  const int d = s.d, Ch = d / 2, C2 = d - Ch, TR = L.TR, nin0 = tab_nin0(s), YS = L.GS;   // (GY rows: the aligned stride)
  const int NP = tab_param_floats(s), NM = d * d + d;
  tab_f32x4 wacc[7][TAB_WG_TILES];   // weight-gradient sums: layers 0..5, then the channel mix
  float bacc[7];
#pragma unroll
  for (int l = 0; l < 7; ++l) {
    bacc[l] = 0.f;
#pragma unroll
    for (int j = 0; j < TAB_WG_TILES; ++j) wacc[l][j] = tab_f32x4{0.f, 0.f, 0.f, 0.f};
  }
  for (int i = threadIdx.x; i < TR * L.GS; i += blockDim.x) { L.GA[i] = 0.f; L.GB[i] = 0.f; L.GY[i] = 0.f; }   // pad columns finite
  tab_stage_resident(s, L);
  int poff[6];   // the layers' blocks of the partial sums
#pragma unroll
  for (int l = 0; l < 6; ++l) poff[l] = tab_param_off(s, l);
  const int ntiles = (s.B + TR - 1) / TR;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int row0 = tile * TR;
    tab_forward_tile(s, L, x, cond, row0);
    // ---- tail backward: grads wrt O (into GA) and wrt y2 (into GY second half)
    if (s.made)
      for (int idx = threadIdx.x; idx < TR * d; idx += blockDim.x) {
        const int p = idx / d, j = idx - p * d, m = row0 + p;
        const float gu = m < s.B ? gout[(long)m * d + j] : 0.f;
        const float gl = (m < s.B && gld) ? gld[m] : 0.f;
        const float mu = L.O[p * L.OS + 2 * j], al = L.O[p * L.OS + 2 * j + 1], e = expf(-al);
        const float u = (L.Y[p * L.XS + j] - mu) * e;
        L.GA[p * L.GS + 2 * j] = -gu * e;
        L.GA[p * L.GS + 2 * j + 1] = m < s.B ? -gu * u - gl : 0.f;
        L.GY[p * YS + j] = gu * e;
      }
    else
      for (int idx = threadIdx.x; idx < TR * C2; idx += blockDim.x) {
        const int p = idx / C2, j = idx - p * C2, m = row0 + p;
        const float g2 = m < s.B ? gout[(long)m * d + Ch + j] : 0.f;
        const float gl = (m < s.B && gld) ? gld[m] : 0.f;
        if (s.additive) {
          L.GA[p * L.GS + j] = g2;
          L.GY[p * YS + Ch + j] = g2;
        } else {
          const float shf = L.O[p * L.OS + 2 * j], sr = L.O[p * L.OS + 2 * j + 1];
          const float sc = sigmoidf(sr + 2.f), z2 = L.Y[p * L.XS + Ch + j];
          const float gshift = g2 * sc;
          L.GA[p * L.GS + 2 * j] = gshift;
          L.GA[p * L.GS + 2 * j + 1] = (m < s.B) ? g2 * (z2 + shf) * sc * (1.f - sc) + gl * (1.f - sc) : 0.f;
          L.GY[p * YS + Ch + j] = gshift;
        }
      }
    __syncthreads();
    // ---- MLP backward, layers 5..0
    float* g = L.GA;
    float* gn = L.GB;
#pragma unroll
    for (int l = 5; l >= 0; --l) {
      const int nin = tab_layer_in(s, l), no = tab_layer_out(s, l);
      const float* in = l == 0 ? L.IN0 : L.H(l > 0 ? l - 1 : 0);
      const int ldi = l == 0 ? L.IS : L.HS;
      tab_wgrad_mfma(g, L.GS, in, ldi, nin, no, wacc[l], bacc[l], TR);
      tab_dense_t_mfma(g, L.GS, no, tab_wr(s, L, l + 1), gn, L.GS, nin, TR);
      if (l > 0) {
        const int act = (l - 1) == 4 ? 2 : 1;
        for (int idx = threadIdx.x; idx < TR * nin; idx += blockDim.x) {
          const int p = idx / nin, c = idx - p * nin;
          const float h = L.H(l > 0 ? l - 1 : 0)[p * L.HS + c];
          gn[p * L.GS + c] *= act == 1 ? (h > 0.f ? 1.f : 0.f) : (1.f - h * h);
        }
        __syncthreads();
      }
      float* t = g; g = gn; gn = t;
    }
    // g = grad wrt IN0 = (z1 | cond)   (MADE: (y | cond), added to the direct term of the tail)
    const int nf = tab_nfeat(s);
    for (int idx = threadIdx.x; idx < TR * nin0; idx += blockDim.x) {
      const int p = idx / nin0, c = idx - p * nin0, m = row0 + p;
      if (c < nf) {
        if (s.made) L.GY[p * YS + c] += g[p * L.GS + c];
        else L.GY[p * YS + c] = g[p * L.GS + c] + (m < s.B ? gout[(long)m * d + c] : 0.f);
      } else if (gcond && m < s.B) gcond[(long)m * s.ncond + (c - nf)] += g[p * L.GS + c];
    }
    __syncthreads();
    // ---- mix backward: partial (gWf, gcf); gx = Wf^T gy
    tab_wgrad_mfma(L.GY, YS, L.X, L.XS, d, d, wacc[6], bacc[6], TR);
    tab_dense_t_mfma(L.GY, YS, d, tab_wr(s, L, 0), gn, L.GS, d, TR);
    for (int idx = threadIdx.x; idx < TR * d; idx += blockDim.x) {
      const int p = idx / d, c = idx - p * d, m = row0 + p;
      if (m < s.B) gx[(long)m * d + c] = gn[p * L.GS + c];
    }
    __syncthreads();
  }
  // the workgroup's slab of partial sums, in the layout tab_reduce_kernel / nfk_prep_bwd read
  float* pb = part + (long)blockIdx.x * NP;
#pragma unroll
  for (int l = 0; l < 6; ++l) {
    const int nin = tab_layer_in(s, l);
    tab_wgrad_store(nin, tab_layer_out(s, l), wacc[l], bacc[l], pb + poff[l], nin + 1, pb + poff[l] + nin, nin + 1);
  }
  float* pm = part_mix + (long)blockIdx.x * NM;
  tab_wgrad_store(d, d, wacc[6], bacc[6], pm, d, pm + d * d, 1);
}

// Backward of tab_inv_kernel (the reverse / sampling pass with gradients: the perceptual branch of the tabular
// distillation loss, pl_module.py:229-243).  Given g_x = dL/dx of the step's reverse OUTPUT x = Wi z + ci:
//   g_z = Wi^T g_x, partial (g_Wi, g_ci)            -> part_mix (same [d*d + d] layout as the forward's (gWf, gcf))
//   z2 = y2 / s - shift, s = sigmoid(sr + 2)        -> g_y2 = g_z2 / s, g_shift = -g_z2, g_sr = -g_z2 y2 (1 - s) / s
//   additive: z2 = y2 - h                           -> g_y2 = g_z2, g_h = -g_z2
//   logdet_out = logdet_in - dld - sum log s        -> g_sr -= g_ld (1 - s)
//   MLP backward                                    -> part, g_z1 += ..., g_cond
// The tile's reverse forward is recomputed in LDS first (y -> MLP -> z).
__global__ void __launch_bounds__(256) tab_inv_bwd_kernel(TabStep s, const float* __restrict__ y,
                                                          const float* __restrict__ cond, const float* __restrict__ gxo,
                                                          const float* __restrict__ gld, float* __restrict__ gy,
                                                          float* __restrict__ gcond, float* __restrict__ part,
                                                          float* __restrict__ part_mix) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  TabLds L;
  tab_layout(s, true, L, smem);
  const int d = s.d, Ch = d / 2, TR = L.TR, nin0 = tab_nin0(s);
  const int NP = tab_param_floats(s), NM = d * d + d;
  for (int i = threadIdx.x; i < NP; i += blockDim.x) L.Pl[i] = 0.f;
  for (int i = threadIdx.x; i < NM; i += blockDim.x) L.PM[i] = 0.f;
  const int ntiles = (s.B + TR - 1) / TR;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int row0 = tile * TR;
    __syncthreads();
    // ---- recompute: Y = y tile, IN0 = (z1 | cond), MLP, X = z = (z1 | z2); GY = g_x tile
    for (int idx = threadIdx.x; idx < TR * d; idx += blockDim.x) {
      const int p = idx / d, c = idx - p * d, m = row0 + p;
      L.Y[p * L.XS + c] = m < s.B ? y[(long)m * d + c] : 0.f;
      L.GY[p * L.XS + c] = m < s.B ? gxo[(long)m * d + c] : 0.f;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < TR * nin0; idx += blockDim.x) {
      const int p = idx / nin0, c = idx - p * nin0, m = row0 + p;
      L.IN0[p * L.IS + c] = c < Ch ? L.Y[p * L.XS + c] : (m < s.B ? cond[(long)m * s.ncond + (c - Ch)] : 0.f);
    }
    const float* in = L.IN0;
    int ldi = L.IS;
    for (int l = 0; l < 5; ++l) {
      tab_dense(in, ldi, tab_layer_in(s, l), s.W[l], s.b[l], L.H(l), L.HS, s.hid, l == 4 ? 2 : 1, 0, L.Wl, TR);
      in = L.H(l);
      ldi = L.HS;
    }
    tab_dense(L.H(4), L.HS, s.hid, s.W[5], s.b[5], L.O, L.OS, tab_nout(s), 0, 0, L.Wl, TR);
    for (int idx = threadIdx.x; idx < TR * d; idx += blockDim.x) {
      const int p = idx / d, c = idx - p * d;
      float v = L.Y[p * L.XS + c];
      if (c >= Ch) {
        const int j = c - Ch;
        if (s.additive) v -= L.O[p * L.OS + j];
        else v = v * (1.f + expf(-(L.O[p * L.OS + 2 * j + 1] + 2.f))) - L.O[p * L.OS + 2 * j];
      }
      L.X[p * L.XS + c] = v;
    }
    __syncthreads();
    // ---- final mix backward: partial (g_Wi, g_ci), g_z = Wi^T g_x -> GB
    for (int idx = threadIdx.x; idx < NM; idx += blockDim.x) {
      float sacc = 0.f;
      if (idx < d * d) {
        const int o = idx / d, c = idx - o * d;
        for (int p = 0; p < TR; ++p) sacc += L.GY[p * L.XS + o] * L.X[p * L.XS + c];
      } else {
        const int o = idx - d * d;
        for (int p = 0; p < TR; ++p) sacc += L.GY[p * L.XS + o];
      }
      L.PM[idx] += sacc;
    }
    tab_dense(L.GY, L.XS, d, s.Wi, nullptr, L.GB, L.GS, d, 0, 1, L.Wl, TR);
    // ---- tail backward: grads wrt the MLP output -> GA; g_y second half and g_z first half -> GY (g_x no longer needed)
    for (int idx = threadIdx.x; idx < TR * d; idx += blockDim.x) {
      const int p = idx / d, c = idx - p * d, m = row0 + p;
      const float gz = L.GB[p * L.GS + c];
      if (c < Ch) { L.GY[p * L.XS + c] = gz; continue; }
      const int j = c - Ch;
      if (s.additive) {
        L.GA[p * L.GS + j] = -gz;
        L.GY[p * L.XS + c] = gz;
      } else {
        const float sc = sigmoidf(L.O[p * L.OS + 2 * j + 1] + 2.f), y2 = L.Y[p * L.XS + c];
        const float gl = (m < s.B && gld) ? gld[m] : 0.f;
        L.GA[p * L.GS + 2 * j] = -gz;
        L.GA[p * L.GS + 2 * j + 1] = m < s.B ? -gz * y2 * (1.f - sc) / sc - gl * (1.f - sc) : 0.f;
        L.GY[p * L.XS + c] = gz / sc;
      }
    }
    __syncthreads();
    const float* g = tab_mlp_backward(s, L);
    // g = grad wrt IN0 = (z1 | cond)
    for (int idx = threadIdx.x; idx < TR * nin0; idx += blockDim.x) {
      const int p = idx / nin0, c = idx - p * nin0, m = row0 + p;
      if (c < Ch) L.GY[p * L.XS + c] += g[p * L.GS + c];
      else if (gcond && m < s.B) gcond[(long)m * s.ncond + (c - Ch)] += g[p * L.GS + c];
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < TR * d; idx += blockDim.x) {
      const int p = idx / d, c = idx - p * d, m = row0 + p;
      if (m < s.B) gy[(long)m * d + c] = L.GY[p * L.XS + c];
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < NP; i += blockDim.x) part[(long)blockIdx.x * NP + i] = L.Pl[i];
  for (int i = threadIdx.x; i < NM; i += blockDim.x) part_mix[(long)blockIdx.x * NM + i] = L.PM[i];
}

// gW[l][o][c] += sum_blk part[blk][off_l + o*(nin+1) + c];  gb[l][o] += ... [+ nin]
__global__ void tab_reduce_kernel(TabStep s, nfk_tab_grads g, const float* __restrict__ part, int nblk) {
  const int NP = tab_param_floats(s);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < NP; i += gridDim.x * blockDim.x) {
    float t = 0.f;
    int b = 0;
    for (; b + 8 <= nblk; b += 8) {   // eight slabs' loads in flight (a plain loop is a chain of nblk round trips)
      float u[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) u[e] = part[(long)(b + e) * NP + i];
#pragma unroll
      for (int e = 0; e < 8; ++e) t += u[e];
    }
    for (; b < nblk; ++b) t += part[(long)b * NP + i];
    int o = 0;
    for (int l = 0; l < 6; ++l) {
      const int nin = tab_layer_in(s, l), no = tab_layer_out(s, l), n = no * (nin + 1);
      if (i < o + n) {
        const int r = (i - o) / (nin + 1), c = (i - o) - r * (nin + 1);
        if (c < nin) g.gW[l][r * nin + c] += t; else g.gb[l][r] += t;
        break;
      }
      o += n;
    }
  }
}

// backward: one workgroup per CU (its LDS holds the tile's activations AND the per-workgroup weight-gradient sums: ~125 KB at
// d = 63), every one with a slab of partial sums of its own.  (128 until round 3: half the chip idle at 65 536 rows.)
constexpr int TAB_MAX_BLOCKS = 256;

static int tab_blocks(const TabStep& s, bool bwd) {
  const int nt = ceil_div(s.B, tab_rows(s.hid, s.d));
  const int cap = bwd ? TAB_MAX_BLOCKS : 2048;
  return nt < cap ? nt : cap;
}
static int tab_check(const TabStep& s, bool bwd, size_t* lds) {
  if (s.d < 2 || s.hid < 1 || s.B < 0 || s.ncond < 0) return NFK_ERR_ARG;
  TabLds dummy;
  *lds = tab_layout(s, bwd, dummy, nullptr);
  return *lds > 160 * 1024 ? NFK_ERR_ARG : NFK_OK;
}
static int g_tab_tall_min_tiles = 512;   // taller forward tiles only when this many of them exist
static int g_tab_resident = 1;   // experiment knob (NFK_TUNE_TAB_RESIDENT): 0 = the per-tile staging of rounds 1-2
// resident mode: every weight matrix in LDS for the life of the workgroup -- if the layout fits, the tile has a multiple of
// four rows (the transposed layers take four gradient rows per thread) and, backward, the private sums fit their slots
static bool tab_resident_ok(const TabStep& s, bool bwd, size_t* lds) {
  if (!g_tab_resident) return false;
  const int TR = tab_rows(s.hid, s.d);
  if (TR < 16 || (TR & 15)) return false;   // (the MFMA tiles: 16 rows)
  if (bwd) {   // the weight-gradient sums of a layer: at most TAB_WG_TILES 16 x 16 tiles per wave, in registers
    if (((s.d + 15) / 16) * ((s.d + 15) / 16) > 4 * TAB_WG_TILES || s.d > 256) return false;
    for (int l = 0; l < 6; ++l)
      if (((tab_layer_out(s, l) + 15) / 16) * ((tab_layer_in(s, l) + 15) / 16) > 4 * TAB_WG_TILES || tab_layer_out(s, l) > 256)
        return false;
  }
  TabLds dummy;
  const size_t n = tab_layout(s, bwd, dummy, nullptr, true);
  if (n > 160 * 1024 - 1024) return false;
  *lds = n;
  return true;
}
void tab_set_resident(int on) {   // bit 0: resident mode; bits 8..: (experiment) least number of tall tiles, 0 = default
  g_tab_resident = on & 1;
  g_tab_tall_min_tiles = (on >> 8) > 0 ? (on >> 8) : 512;
}

}  // namespace nfk

using namespace nfk;
extern "C" {

int nfk_tab_bwd_blocks(const nfk_tab_step* s) { return tab_blocks(*s, true); }
int nfk_tab_param_floats(const nfk_tab_step* s) { return tab_param_floats(*s); }

int nfk_tab_fwd(const nfk_tab_step* s, const float* x, const float* cond, const float* ld_in, float* zout,
                float* ld_out, void* stream) {
  size_t lds;
  if (tab_check(*s, false, &lds) != NFK_OK) return NFK_ERR_ARG;
  if (s->B == 0) return NFK_OK;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(tab_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024);
  size_t lds_r;
  int res = tab_resident_ok(*s, false, &lds_r) ? tab_rows(s->hid, s->d) : 0;
  int nb = tab_blocks(*s, false);
  if (res) {   // a workgroup stages the weights once and then walks its tiles: as many as fit the chip, no more
    // taller tiles (a tile's phases are chains of short dependent steps: twice the rows cost little more) while there are
    // enough tiles left for every CU
    for (int tr = 64; tr > res; tr >>= 1) {
      TabLds dummy;
      const size_t n = tab_layout(*s, false, dummy, nullptr, true, tr);
      // ... and only where they do not cost a co-resident workgroup (two per CU overlap each other's waits)
      const int per_base = (int)((160 * 1024) / (lds_r + 1024)), per_tr = (int)((160 * 1024) / (n + 1024));
      if (n <= 160 * 1024 - 1024 && ceil_div(s->B, tr) >= g_tab_tall_min_tiles && per_tr >= (per_base < 2 ? per_base : 2)) {
        res = tr; lds_r = n; break;
      }
    }
    lds = lds_r;
    nb = ceil_div(s->B, res);
    const int per_cu = (int)((160 * 1024) / (lds_r + 1024));
    const int cap = 256 * (per_cu < 1 ? 1 : (per_cu > 4 ? 4 : per_cu));
    if (nb > cap) nb = cap;
  }
  hipLaunchKernelGGL(tab_fwd_kernel, dim3(nb), dim3(256), lds, reinterpret_cast<hipStream_t>(stream),
                     *s, x, cond, ld_in, zout, ld_out, res);
  NFK_LAUNCH_CHECK();
  return NFK_OK;
}
int nfk_tab_inv(const nfk_tab_step* s, const float* y, const float* cond, const float* ld_in, float* xout,
                float* ld_out, void* stream) {
  size_t lds;
  if (tab_check(*s, false, &lds) != NFK_OK || !s->Wi) return NFK_ERR_ARG;
  if (s->B == 0) return NFK_OK;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(tab_inv_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024);
  hipLaunchKernelGGL(tab_inv_kernel, dim3(tab_blocks(*s, false)), dim3(256), lds, reinterpret_cast<hipStream_t>(stream),
                     *s, y, cond, ld_in, xout, ld_out);
  NFK_LAUNCH_CHECK();
  return NFK_OK;
}
int nfk_tab_bwd(const nfk_tab_step* s, const nfk_tab_grads* g, const float* x, const float* cond,
                const float* g_out, const float* g_ld, float* g_x, float* g_cond, float* part, float* part_mix,
                void* stream) {
  size_t lds;
  if (tab_check(*s, true, &lds) != NFK_OK) return NFK_ERR_ARG;
  if (s->B == 0) return NFK_OK;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(tab_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024);
  const int nb = tab_blocks(*s, true);
  size_t lds_r;
  if (tab_resident_ok(*s, true, &lds_r)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(tab_bwd_res_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024);
    // taller tiles while every workgroup of the (unchanged) grid still gets one: the slab count nb is the caller's
    int tr = tab_rows(s->hid, s->d);
    for (int t = 64; t > tr; t >>= 1) {
      TabLds dummy;
      const size_t n = tab_layout(*s, true, dummy, nullptr, true, t);
      if (n <= 160 * 1024 - 1024 && ceil_div(s->B, t) >= nb && ceil_div(s->B, t) >= g_tab_tall_min_tiles / 2) { tr = t; lds_r = n; break; }
    }
    hipLaunchKernelGGL(tab_bwd_res_kernel, dim3(nb), dim3(256), lds_r, st, *s, x, cond, g_out, g_ld, g_x, g_cond, part, part_mix,
                       tr);
  } else {
    hipLaunchKernelGGL(tab_bwd_kernel, dim3(nb), dim3(256), lds, st, *s, x, cond, g_out, g_ld, g_x, g_cond, part, part_mix);
  }
  NFK_LAUNCH_CHECK();
  hipLaunchKernelGGL(tab_reduce_kernel, dim3(ceil_div(tab_param_floats(*s), 256)), dim3(256), 0, st, *s, *g, part, nb);
  NFK_LAUNCH_CHECK();
  return NFK_OK;
}
int nfk_tab_inv_bwd(const nfk_tab_step* s, const nfk_tab_grads* g, const float* y, const float* cond,
                    const float* g_x, const float* g_ld, float* g_y, float* g_cond, float* part, float* part_mix,
                    void* stream) {
  size_t lds;
  if (tab_check(*s, true, &lds) != NFK_OK || !s->Wi || s->made) return NFK_ERR_ARG;   // (no reverse pass with gradients for MADE)
  if (s->B == 0) return NFK_OK;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(tab_inv_bwd_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const int nb = tab_blocks(*s, true);
  hipLaunchKernelGGL(tab_inv_bwd_kernel, dim3(nb), dim3(256), lds, st, *s, y, cond, g_x, g_ld, g_y, g_cond, part,
                     part_mix);
  NFK_LAUNCH_CHECK();
  hipLaunchKernelGGL(tab_reduce_kernel, dim3(ceil_div(tab_param_floats(*s), 256)), dim3(256), 0, st, *s, *g, part, nb);
  NFK_LAUNCH_CHECK();
  return NFK_OK;
}
}
