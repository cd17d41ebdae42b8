// nfk flow kernels: the HBM-bound part of a Glow FlowStep on NHWC activations.
//
//   prep        : per-step parameter products, once per forward for ALL steps in one launch:
//                 W = P(L+I)(U+diag(sign_s e^{log_s}))           models/layers.py:380-397
//                 Wf = W diag(e^{logs}), cf = Wf bias  (ActNorm folded into the 1x1 conv,
//                 models/layers.py:101-142 + :404-421), W^-1 by pivoted Gauss-Jordan instead of
//                 three torch.inverse calls (:390-395), dlogdet = pixels (sum logs + sum log_s),
//                 e^{logs} of the coupling net's ActNorms and e^{3 logs} of Conv2dZeros (:260).
//   chanmix     : z = Wf x + cf per pixel  (ActNorm2d/1d + InvertibleConv1x1 / Permute2d)
//   coupling    : affine/additive coupling tail + per-sample log-det (models/flows.py:157-169)
//   squeeze     : space-to-depth (models/layers.py:32-61), NCHW boundary folded in
//   split prior : Split2d Gaussian log-density (models/layers.py:293-313)
//   top prior   : gaussian_likelihood + bits/dim (models/kd_flows.py:136-150)
// plus the inverse and backward of each.  Log-det sums use one workgroup per sample with a
// fixed-order wave/LDS tree (no atomics), so results are bitwise reproducible run to run.
#include "nfk_internal.h"
#include "nfk_tail.h"
#include "nfk_repack.h"

namespace nfk {

// ==========================================================================================
// small dense helpers on global scratch, executed by ONE workgroup
// ==========================================================================================
constexpr size_t PREP_INV_BWD_LDS = 152 * 1024;   // seven C x C matrices: C <= 74
constexpr size_t PREP_LDS = 60 * 1024, PREP_BWD_LDS = 60 * 1024;   // dynamic LDS of prep_kernel / prep_bwd_kernel
// C[n][n] = A * B   (opA/opB: 0 = as is, 1 = transposed)
// Block-wide batch of dot products over k = 0..n-1, at most EP per thread, ALL of a thread's outputs advancing together:
// one pass over k with 2 EP LDS reads in flight per step.  (One output after the other, every output was a chain of n
// dependent read -> fma steps at full LDS latency: 37 us for the LU chain of a 48-channel step, measured with clock stamps.)
// addr(idx, pa, sa, pb, sb): the two operand rows of output idx (pointer, stride in floats); out(idx, v) stores it.
template <int EP, typename FA, typename FO>
__device__ __forceinline__ void blk_dots(int total, int n, FA addr, FO out) {
  const float* pa[EP];
  const float* pb[EP];
  int sa[EP], sb[EP];
  float s[EP];
#pragma unroll
  for (int e = 0; e < EP; ++e) {
    const int idx = threadIdx.x + e * blockDim.x;
    addr(idx < total ? idx : 0, pa[e], sa[e], pb[e], sb[e]);
    s[e] = 0.f;
  }
#pragma unroll 2
  for (int k = 0; k < n; ++k) {
#pragma unroll
    for (int e = 0; e < EP; ++e) {
      s[e] += pa[e][0] * pb[e][0];
      pa[e] += sa[e];
      pb[e] += sb[e];
    }
  }
#pragma unroll
  for (int e = 0; e < EP; ++e) {
    const int idx = threadIdx.x + e * blockDim.x;
    if (idx < total) out(idx, e, s[e]);
  }
}
constexpr int DOT_EP = 10;   // outputs per thread of the batched form: n * n <= 10 * 256 (n <= 50: every image level up to C = 48)

__device__ __forceinline__ void blk_matmul(const float* A, int ta, const float* B, int tb, float* C, int n) {
  const int ak = ta ? n : 1, ai = ta ? 1 : n, bk = tb ? 1 : n, bj = tb ? n : 1;
  if (n * n <= DOT_EP * (int)blockDim.x) {
    blk_dots<DOT_EP>(n * n, n,
                     [&](int idx, const float*& pa, int& sa, const float*& pb, int& sb) {
                       pa = A + (idx / n) * ai; sa = ak; pb = B + (idx % n) * bj; sb = bk;
                     },
                     [&](int idx, int, float v) { C[idx] = v; });
  } else {
    for (int idx = threadIdx.x; idx < n * n; idx += blockDim.x) {
      const int i = idx / n, j = idx % n;
      const float* a = A + i * ai;
      const float* b = B + j * bj;
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
      int k = 0;
      for (; k + 4 <= n; k += 4) {
        const float a0 = a[k * ak], a1 = a[(k + 1) * ak], a2 = a[(k + 2) * ak], a3 = a[(k + 3) * ak];
        const float b0 = b[k * bk], b1 = b[(k + 1) * bk], b2 = b[(k + 2) * bk], b3 = b[(k + 3) * bk];
        s0 += a0 * b0; s1 += a1 * b1; s2 += a2 * b2; s3 += a3 * b3;
      }
      for (; k < n; ++k) s0 += a[k * ak] * b[k * bk];
      C[idx] = (s0 + s1) + (s2 + s3);
    }
  }
  __syncthreads();
}

// In-place Gauss-Jordan inverse with partial pivoting of A (n x n) using aug[n][2n] scratch.
// Returns log|det A| (valid in all threads).  sh must hold >= 4 floats/ints.
__device__ __forceinline__ float blk_inverse(const float* A, float* inv, float* aug, int n, float* sh) {
  const int n2 = 2 * n;
  for (int idx = threadIdx.x; idx < n * n2; idx += blockDim.x) {
    const int i = idx / n2, j = idx % n2;
    aug[idx] = j < n ? A[i * n + j] : (j - n == i ? 1.f : 0.f);
  }
  __syncthreads();
  float logdet = 0.f;
  for (int k = 0; k < n; ++k) {
    if (threadIdx.x == 0) {
      int best = k;
      float bv = fabsf(aug[k * n2 + k]);
      for (int r = k + 1; r < n; ++r) {
        const float v = fabsf(aug[r * n2 + k]);
        if (v > bv) { bv = v; best = r; }
      }
      reinterpret_cast<int*>(sh)[0] = best;
    }
    __syncthreads();
    const int piv = reinterpret_cast<int*>(sh)[0];
    if (piv != k) {
      for (int j = threadIdx.x; j < n2; j += blockDim.x) {
        const float t = aug[k * n2 + j];
        aug[k * n2 + j] = aug[piv * n2 + j];
        aug[piv * n2 + j] = t;
      }
    }
    __syncthreads();
    const float d = aug[k * n2 + k];
    logdet += logf(fabsf(d));
    __syncthreads();
    const float rd = 1.f / d;
    for (int j = threadIdx.x; j < n2; j += blockDim.x) aug[k * n2 + j] *= rd;
    __syncthreads();
    for (int idx = threadIdx.x; idx < n * n2; idx += blockDim.x) {
      const int i = idx / n2, j = idx % n2;
      if (i == k) continue;
      // column k of row i is needed by every j of that row: read before any thread overwrites it
      // (it is overwritten only by j == k, which is handled after the barrier below)
      if (j != k) aug[idx] -= aug[i * n2 + k] * aug[k * n2 + j];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += blockDim.x)
      if (i != k) aug[i * n2 + k] = 0.f;
    __syncthreads();
  }
  for (int idx = threadIdx.x; idx < n * n; idx += blockDim.x) inv[idx] = aug[(idx / n) * n2 + n + idx % n];
  __syncthreads();
  return logdet;
}

// scratch layout inside PrepDesc::scratch (floats): T0,T1,T2 [C*C] each, AUG [2*C*C]
__device__ __forceinline__ void build_W(const PrepDesc& d, float* W, float* T0, float* T1, float* sh) {
  const int C = d.C;
  if (d.perm_kind == PERM_LU) {
    for (int idx = threadIdx.x; idx < C * C; idx += blockDim.x) {
      const int i = idx / C, j = idx % C;
      T0[idx] = i > j ? d.lower[idx] : (i == j ? 1.f : 0.f);
      T1[idx] = i < j ? d.upper[idx] : (i == j ? d.sign_s[i] * expf(d.log_s[i]) : 0.f);
    }
    __syncthreads();
    blk_matmul(T0, 0, T1, 0, W, C);        // W <- L U   (temporarily)
    for (int idx = threadIdx.x; idx < C * C; idx += blockDim.x) T0[idx] = W[idx];
    __syncthreads();
    blk_matmul(d.p, 0, T0, 0, W, C);       // W <- P (L U)
  } else if (d.perm_kind == PERM_DENSE) {
    for (int idx = threadIdx.x; idx < C * C; idx += blockDim.x) W[idx] = d.weight[idx];
    __syncthreads();
  } else {  // channel gather: z[:, o] = y[:, idx[o]]
    for (int idx = threadIdx.x; idx < C * C; idx += blockDim.x)
      W[idx] = (idx % C == d.perm_idx[idx / C]) ? 1.f : 0.f;
    __syncthreads();
  }
}

// `base`: 6 C x C work matrices, either the kernel's dynamic LDS (copy_p: P is staged there too) or the step's global
// scratch -- two inlined instances, so that the LDS one uses ds_ instructions (see prep_bwd_body).
__device__ __forceinline__ void prep_body(PrepDesc d, int want_inverse, float* base, const bool copy_p, float* sh) {
  const int C = d.C, CC = C * C;
  float* W = base;
  float* T0 = W + CC;
  float* T1 = T0 + CC;
  float* AUG = T1 + CC;
  // coupling-net scales first: they depend on nothing, and their loads travel under everything below
  for (int i = threadIdx.x; i < d.n1; i += blockDim.x) d.es1[i] = expf(d.l1[i]);
  for (int i = threadIdx.x; i < d.n2; i += blockDim.x) d.es2[i] = expf(d.l2[i]);
  for (int i = threadIdx.x; i < d.n3; i += blockDim.x) d.es3[i] = expf(3.f * d.l3[i]);
  if (C > 0) {
  if (copy_p && d.perm_kind == PERM_LU) {
    float* Pl = W + 5 * CC;
    for (int idx = threadIdx.x; idx < CC; idx += blockDim.x) Pl[idx] = d.p[idx];
    d.p = Pl;   // build_W's first barrier comes before its first use
  }
  build_W(d, W, T0, T1, sh);

  // sums of logs / log_s
  float s = 0.f;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    s += d.an_logs[c];
    if (d.perm_kind == PERM_LU) s += d.log_s[c];
  }
  s = block_sum(s, sh);
  float logabsdet = 0.f;
  if (d.perm_kind == PERM_DENSE || want_inverse) {
    logabsdet = blk_inverse(W, T0, AUG, C, sh);  // T0 <- W^-1
    if (d.perm_kind == PERM_DENSE) s += logabsdet;
  }
  if (threadIdx.x == 0) d.dld[0] = s * (float)d.pixels;

  // forward products.  2-D: z = W y ; 1-D: z = y @ W  (y = (x + b) e^{logs}).  Wf also stays in the work area (T1 is
  // free by now) so that cf does not wait for a read-back of what this workgroup has just written to memory
  for (int idx = threadIdx.x; idx < CC; idx += blockDim.x) {
    const int o = idx / C, c = idx % C;
    const float w = d.is_1d ? W[c * C + o] : W[idx];
    const float v = w * expf(d.an_logs[c]);
    d.Wf[idx] = v;
    T1[idx] = v;
  }
  __syncthreads();
  for (int o = threadIdx.x; o < C; o += blockDim.x) {
    float t = 0.f;
    for (int c = 0; c < C; ++c) t += T1[o * C + c] * d.an_bias[c];
    d.cf[o] = t;
  }
  if (want_inverse && d.Wi) {
    // x_c = e^{-logs_c} sum_o Winv[c][o] z_o - b_c   (1-D: Winv[o][c])
    for (int idx = threadIdx.x; idx < CC; idx += blockDim.x) {
      const int c = idx / C, o = idx % C;
      const float w = d.is_1d ? T0[o * C + c] : T0[idx];
      d.Wi[idx] = w * expf(-d.an_logs[c]);
    }
    for (int c = threadIdx.x; c < C; c += blockDim.x) d.ci[c] = -d.an_bias[c];
  }
  }  // C > 0
}

__global__ void __launch_bounds__(256) prep_kernel(const PrepDesc* __restrict__ descs, int want_inverse) {
  __shared__ float sh[32];
  const PrepDesc d = descs[blockIdx.x];
  // work matrices (and P) in LDS when six C x C matrices fit the 60 KiB of the launch, else in the step's scratch
  extern __shared__ __attribute__((aligned(16))) float pf_lds[];
  if ((size_t)6 * d.C * d.C * sizeof(float) <= PREP_LDS) prep_body(d, want_inverse, pf_lds, true, sh);
  else prep_body(d, want_inverse, d.scratch, false, sh);
}

// Everything a forward needs from the parameters in ONE launch: the steps' prep (first n_prep workgroups: one latency
// chain of ~40 us each) and, beside them, the conv weight repacks (64 workgroups per weight; f32 descriptors first).
// As separate launches on one stream the three ran one after the other at the head of every training step.
constexpr int PR_THREADS = 512;   // 60 KiB of LDS per workgroup leave room for two per CU: wider workgroups keep the CU's
                                  // memory pipeline busy during the repack
__global__ void __launch_bounds__(PR_THREADS) prep_repack_kernel(const PrepDesc* __restrict__ descs, int n_prep, int want_inverse,
                                                          const nfk_repack_desc* __restrict__ rf32, int n_f32,
                                                          const nfk_repack_desc* __restrict__ rbf16) {
  __shared__ float sh[32];
  extern __shared__ __attribute__((aligned(16))) float pf_lds[];
  const int b = blockIdx.x;
  if (b < n_prep) {
    const PrepDesc d = descs[b];
    if ((size_t)6 * d.C * d.C * sizeof(float) <= PREP_LDS) prep_body(d, want_inverse, pf_lds, true, sh);
    else prep_body(d, want_inverse, d.scratch, false, sh);
    return;
  }
  // a repack workgroup = two 256-thread workers on two tiles of the same weight (half the LDS each): with 60 KiB per
  // workgroup only two fit a CU, and a tile is a chain of three memory round trips -- this halves the number of chains
  const int r = b - n_prep, w = r >> 5, half = threadIdx.x >> 8, bx = (r & 31) * 2 + half, tid = threadIdx.x & 255;
  if (w < n_f32) repack_f32_body(rf32[w], bx, 64, tid, 256);
  else repack_bf16_body(rbf16[w - n_f32], bx, 64, pf_lds + half * RP_LDS_FLOATS, tid, 256);
}

// In-place reduction of the per-workgroup partial (gWf, gcf) slabs of every step into slab 0:
// grid = (chunks of 256 elements, steps); element e of all slabs is touched by one thread only.
__global__ void __launch_bounds__(256) part_reduce_kernel(const PrepDesc* __restrict__ descs,
                                                          const PrepGrad* __restrict__ grads) {
  const PrepDesc d = descs[blockIdx.y];
  const PrepGrad g = grads[blockIdx.y];
  const int E = d.C * d.C + d.C;
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (d.C == 0 || e >= E || g.nparts <= 1) return;
  float* part = const_cast<float*>(g.part);
  part[e] = slab_sum(part, g.nparts, E, e);
}


// W = P L U: chain a gradient GW (indexed like W) to (lower, upper, log_s); Gs is added to every g_log_s (the log-det
// term).  T0, T1, T2, Wk: C*C scratch matrices (T0 / T1 are overwritten with L / U).  Block-wide; accumulates.
// pre (optional): this thread's elements idx = threadIdx.x + e * blockDim.x of `lower`, `upper` and of the diagonal
// s = sign_s * exp(log_s), loaded by the caller at the start of the kernel (LU_PF per array; only when C*C fits)
constexpr int LU_PF = 10;   // C * C <= 10 * 256: every image level up to C = 48 (and 1-D flows up to d = 50)
struct LuPre { float lo[LU_PF], up[LU_PF], sd[LU_PF], old[LU_PF]; };
// (old: the element of g_lower / g_upper / g_log_s that output idx of the chain is added to)
__device__ __forceinline__ float* lu_target(const PrepGrad& g, int C, int idx) {
  const int i = idx / C, j = idx % C;
  return i > j ? g.g_lower + idx : (i < j ? g.g_upper + idx : g.g_log_s + i);
}
__device__ __forceinline__ void lu_preload(const PrepDesc& d, const PrepGrad& g, int C, LuPre& pre) {
#pragma unroll
  for (int e = 0; e < LU_PF; ++e) {
    const int idx = min((int)(threadIdx.x + e * blockDim.x), C * C - 1), i = idx / C;
    pre.lo[e] = d.lower[idx];
    pre.up[e] = d.upper[idx];
    pre.sd[e] = d.sign_s[i] * expf(d.log_s[i]);
    pre.old[e] = *lu_target(g, C, idx);
  }
}
__device__ __forceinline__ void prep_lu_chain(const PrepDesc& d, const PrepGrad& g, int C, const float* P, const float* GW, float* Wk,
                              float* T0, float* T1, float* T2, float Gs, const LuPre* pre = nullptr) {
  const int CC = C * C;
  // T0 = L (unit lower), T1 = U^T (U = strict upper + diag s): transposed, so that both dot products below read their
  // second operand along consecutive columns (U itself: a 16-way bank conflict per read at C = 48)
  if (pre) {
#pragma unroll
    for (int e = 0; e < LU_PF; ++e) {
      const int idx = threadIdx.x + e * blockDim.x;
      if (idx < CC) {
        const int i = idx / C, j = idx % C;
        T0[idx] = i > j ? pre->lo[e] : (i == j ? 1.f : 0.f);
        T1[j * C + i] = i < j ? pre->up[e] : (i == j ? pre->sd[e] : 0.f);
      }
    }
  } else {
    for (int idx = threadIdx.x; idx < CC; idx += blockDim.x) {
      const int i = idx / C, j = idx % C;
      T0[idx] = i > j ? d.lower[idx] : (i == j ? 1.f : 0.f);
      T1[j * C + i] = i < j ? d.upper[idx] : (i == j ? d.sign_s[i] * expf(d.log_s[i]) : 0.f);
    }
  }
  __syncthreads();
  blk_matmul(P, 1, GW, 0, T2, C);     // T2 = P^T gW
  // g_L = strict lower part of T2 U^T, g_U / g_s = upper part + diagonal of L^T T2: every (i, j) needs ONE of the two dot
  // products.  The triangular factors sit in T0 / T1 as full matrices with their zeros: every output runs over all k, so
  // that the outputs of a thread advance together (blk_dots).  The gradient elements the results are added to come in
  // `pre`, requested at the start of the kernel (as a read-modify-write per output each was a memory round trip).
  auto pick = [&](int idx, const float*& pa, int& sa, const float*& pb, int& sb) {
    const int i = idx / C, j = idx % C;
    if (i > j) { pa = T2 + i * C; sa = 1; pb = T1 + j; sb = C; }    // sum_k T2[i][k] U[j][k]
    else { pa = T0 + i; sa = C; pb = T2 + j; sb = C; }               // sum_k L[k][i] T2[k][j]
  };
  if (pre) {
    float res[LU_PF];
    blk_dots<LU_PF>(CC, C, pick, [&](int, int e, float v) { res[e] = v; });
#pragma unroll
    for (int e = 0; e < LU_PF; ++e) {
      const int idx = threadIdx.x + e * blockDim.x;
      if (idx < CC) *lu_target(g, C, idx) = pre->old[e] + (idx / C == idx % C ? res[e] * pre->sd[e] + Gs : res[e]);
    }
  } else {
    for (int idx = threadIdx.x; idx < CC; idx += blockDim.x) {
      const float* pa; const float* pb; int sa, sb;
      pick(idx, pa, sa, pb, sb);
      float v = 0.f;
      for (int k = 0; k < C; ++k) v += pa[k * sa] * pb[k * sb];
      const int i = idx / C;
      *lu_target(g, C, idx) += (i == idx % C) ? v * d.sign_s[i] * expf(d.log_s[i]) + Gs : v;
    }
  }
}

// Backward of prep: chain (gWf, gcf, G = sum_b g_logdet[b]) to the raw parameters; accumulates.
//   part: [nparts][C*C + C] partial sums of  gWf[o][c] = sum_m gz[m][o] x[m][c],  gcf[o] = sum_m gz[m][o]
// `base`: 6 C x C work matrices.  Called once with the kernel's dynamic LDS and once with the step's global scratch so
// that, after inlining, the LDS instance addresses LDS with ds_ instructions (a pointer selected at run time between the
// two would make every access a flat load: same latency as global memory).
__device__ __forceinline__ void prep_bwd_body(const PrepDesc& d, const PrepGrad& g, const float* __restrict__ gld, int B,
                                              float* base, const bool copy_p, float* sh, float* vec) {
  const int C = d.C, CC = C * C;
  float* W = base;
  float* T0 = W + CC;
  float* T1 = T0 + CC;
  float* AUG = T1 + CC;   // 2*CC floats: use as two more matrices
  float* GW = AUG;
  float* T2 = AUG + CC;
  const float* P = copy_p ? W + 5 * CC : d.p;   // only read for the LU parametrisation
  if (copy_p && d.perm_kind == PERM_LU) {
    float* Pl = W + 5 * CC;
    for (int idx = threadIdx.x; idx < CC; idx += blockDim.x) Pl[idx] = d.p[idx];   // visible after the first barrier below
  }

  // everything the chain below reads from memory is requested here, before the first wait: the L / U / s elements into
  // registers, Wf into the W work matrix (free until the LU chain writes it)
  LuPre pre;
  const bool use_pre = d.perm_kind == PERM_LU && CC <= LU_PF * (int)blockDim.x;
  if (use_pre) lu_preload(d, g, C, pre);
  // Wf -> W and the reduced partials -> T0 / T1: all of a thread's loads go out before the first one is waited for
  constexpr int EP = 10;
  if (CC + C <= EP * (int)blockDim.x) {
    float wv[EP], pv[EP];
#pragma unroll
    for (int e = 0; e < EP; ++e) {
      const int idx = threadIdx.x + e * blockDim.x;
      wv[e] = idx < CC ? d.Wf[idx] : 0.f;
      pv[e] = idx < CC + C ? g.part[idx] : 0.f;   // slab 0 holds the total (part_reduce_kernel)
    }
#pragma unroll
    for (int e = 0; e < EP; ++e) {
      const int idx = threadIdx.x + e * blockDim.x;
      if (idx < CC) { W[idx] = wv[e]; T0[idx] = pv[e]; }
      else if (idx < CC + C) T1[idx - CC] = pv[e];
    }
  } else {
    for (int idx = threadIdx.x; idx < CC; idx += blockDim.x) W[idx] = d.Wf[idx];
    for (int idx = threadIdx.x; idx < CC + C; idx += blockDim.x) {
      const float s = g.part[idx];
      if (idx < CC) T0[idx] = s; else T1[idx - CC] = s;
    }
  }
  // ActNorm's bias and e^{logs} once into LDS (they were re-read from memory per matrix element below), and the two
  // per-channel gradients this body accumulates into, requested now and written at the end
  float* vb = vec;
  float* ve = vec + C;
  float old_b = 0.f, old_l = 0.f;
  for (int c = threadIdx.x; c < C; c += blockDim.x) { vb[c] = d.an_bias[c]; ve[c] = expf(d.an_logs[c]); }
  if ((int)threadIdx.x < C) { old_b = g.g_an_bias[threadIdx.x]; old_l = g.g_an_logs[threadIdx.x]; }

  float G = 0.f;
  for (int b = threadIdx.x; b < B; b += blockDim.x) G += gld[b];
  G = block_sum(G, sh) * (float)d.pixels;
  __syncthreads();
  // g_bias[c] = sum_o Wf[o][c] gcf[o];  gWf_tot = gWf + gcf b^T
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float t0 = 0.f, t1 = 0.f;
    int o = 0;
    for (; o + 2 <= C; o += 2) { t0 += W[o * C + c] * T1[o]; t1 += W[(o + 1) * C + c] * T1[o + 1]; }
    if (o < C) t0 += W[o * C + c] * T1[o];
    if (c == (int)threadIdx.x) g.g_an_bias[c] = old_b + (t0 + t1); else g.g_an_bias[c] += t0 + t1;
  }
  for (int idx = threadIdx.x; idx < CC; idx += blockDim.x) T0[idx] += T1[idx / C] * vb[idx % C];
  __syncthreads();
  // g_logs[c] = sum_o gWf_tot[o][c] Wf[o][c] + G ;  gW (in W's own index convention) -> GW
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float t0 = 0.f, t1 = 0.f;
    int o = 0;
    for (; o + 2 <= C; o += 2) { t0 += T0[o * C + c] * W[o * C + c]; t1 += T0[(o + 1) * C + c] * W[(o + 1) * C + c]; }
    if (o < C) t0 += T0[o * C + c] * W[o * C + c];
    if (c == (int)threadIdx.x) g.g_an_logs[c] = old_l + (t0 + t1) + G; else g.g_an_logs[c] += (t0 + t1) + G;
  }
  for (int idx = threadIdx.x; idx < CC; idx += blockDim.x) {
    const int o = idx / C, c = idx % C;
    const float v = T0[idx] * ve[c];
    if (d.is_1d) GW[c * C + o] = v; else GW[idx] = v;
  }
  __syncthreads();
  if (d.perm_kind == PERM_LU) prep_lu_chain(d, g, C, P, GW, W, T0, T1, T2, G, use_pre ? &pre : nullptr);
}

__global__ void __launch_bounds__(256) prep_bwd_kernel(const PrepDesc* __restrict__ descs,
                                                       const PrepGrad* __restrict__ grads,
                                                       const float* __restrict__ gld, int B) {
  __shared__ float sh[32];
  __shared__ float vec[2 * 256];   // ActNorm bias | e^{logs} of the step (C <= 256)
  const PrepDesc d = descs[blockIdx.x];
  const PrepGrad g = grads[blockIdx.x];
  if (d.C == 0 || d.C > 256 || !g.part) return;   // (no partial sums: a step whose chain a level call has run already)
  // the C x C work matrices (and P) live in LDS when six of them fit the 60 KiB the launch provides (C <= 50);
  // wider steps fall back to the step's global scratch
  extern __shared__ __attribute__((aligned(16))) float pb_lds[];
  if ((size_t)6 * d.C * d.C * sizeof(float) <= PREP_BWD_LDS) prep_bwd_body(d, g, gld, B, pb_lds, true, sh, vec);
  else prep_bwd_body(d, g, gld, B, d.scratch, false, sh, vec);
}

// Dense-weight variant needs GW after the inverse; done in a second tiny kernel to keep scratch small.
__global__ void __launch_bounds__(256) prep_bwd_dense_kernel(const PrepDesc* __restrict__ descs,
                                                             const PrepGrad* __restrict__ grads,
                                                             const float* __restrict__ gld, int B) {
  __shared__ float sh[32];
  const PrepDesc d = descs[blockIdx.x];
  if (d.perm_kind != PERM_DENSE) return;
  const PrepGrad g = grads[blockIdx.x];
  if (!g.part) return;
  const int C = d.C, CC = C * C;
  float* W = d.scratch;
  float* T0 = W + CC;
  float* T1 = T0 + CC;
  float* AUG = T1 + CC;
  float G = 0.f;
  for (int b = threadIdx.x; b < B; b += blockDim.x) G += gld[b];
  G = block_sum(G, sh) * (float)d.pixels;
  for (int idx = threadIdx.x; idx < CC; idx += blockDim.x) W[idx] = d.weight[idx];
  __syncthreads();
  blk_inverse(W, T0, AUG, C, sh);  // T0 = W^-1
  // recompute gWf_tot from partials
  for (int idx = threadIdx.x; idx < CC + C; idx += blockDim.x) {
    const float s = g.part[idx];
    if (idx < CC) AUG[idx] = s; else T1[idx - CC] = s;
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < CC; idx += blockDim.x) {
    const int o = idx / C, c = idx % C;
    const float v = (AUG[idx] + T1[o] * d.an_bias[c]) * expf(d.an_logs[c]);
    // d log|det W| / dW = W^-T
    if (d.is_1d) g.g_weight[c * C + o] += v + G * T0[o * C + c];
    else g.g_weight[idx] += v + G * T0[c * C + o];
  }
}

// Backward of prep's INVERSE products (nfk_prep with want_inverse: Wi, ci used by the reverse pass), for reverse passes
// that carry gradients.  part (slab 0 after part_reduce_kernel) holds g_Wi [C][C] and g_ci [C]; gld = dL/d(logdet) of the
// reverse pass, whose logdet is logdet_in - dld (so every log-det parameter gets -G).
//   ci = -bias                                   -> g_bias -= g_ci
//   Wi[c][o] = Y(c,o) e^{-logs[c]},  Y(c,o) = Winv[o][c] (1-D) or Winv[c][o] (2-D)
//                                                -> g_logs[c] -= sum_o g_Wi[c][o] Wi[c][o] + G ;  g_Y = g_Wi e^{-logs[c]}
//   Winv = W^-1                                  -> g_W = -Winv^T g_Winv Winv^T, then the P L U chain (or g_weight)
__global__ void __launch_bounds__(256) prep_inv_bwd_kernel(const PrepDesc* __restrict__ descs,
                                                           const PrepGrad* __restrict__ grads,
                                                           const float* __restrict__ gld, int B) {
  __shared__ float sh[32];
  PrepDesc d = descs[blockIdx.x];
  const PrepGrad g = grads[blockIdx.x];
  const int C = d.C, CC = C * C;
  if (C == 0 || g.nparts <= 0) return;
  extern __shared__ __attribute__((aligned(16))) float pi_lds[];
  const bool in_lds = (size_t)7 * CC * sizeof(float) <= PREP_INV_BWD_LDS;   // C <= 74 (checked by the caller)
  float* W = in_lds ? pi_lds : d.scratch;   // the global scratch holds 5 matrices: 6 and 7 only exist in LDS
  float* T0 = W + CC;
  float* T1 = T0 + CC;
  float* AUG = T1 + CC;                      // 2 CC
  float* GY = in_lds ? AUG + 2 * CC : nullptr;
  float* T2 = in_lds ? GY + CC : nullptr;
  if (!in_lds) return;
  float G = 0.f;
  if (gld) for (int b = threadIdx.x; b < B; b += blockDim.x) G += gld[b];
  G = block_sum(G, sh) * (float)d.pixels;
  build_W(d, W, T0, T1, sh);                 // W (uses T0/T1 as scratch)
  __syncthreads();
  for (int idx = threadIdx.x; idx < CC; idx += blockDim.x) T2[idx] = W[idx];
  __syncthreads();
  blk_inverse(T2, T0, AUG, C, sh);           // T0 = W^-1
  // g_bias, g_logs, g_Winv (indexed like Winv) -> GY
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    g.g_an_bias[c] -= g.part[CC + c];
    const float e = expf(-d.an_logs[c]);
    float t = 0.f;
    for (int o = 0; o < C; ++o) {
      const float y = d.is_1d ? T0[o * C + c] : T0[c * C + o];
      t += g.part[c * C + o] * y * e;
    }
    g.g_an_logs[c] += -t - G;
  }
  for (int idx = threadIdx.x; idx < CC; idx += blockDim.x) {
    const int c = idx / C, o = idx - c * C;
    const float v = g.part[idx] * expf(-d.an_logs[c]);
    if (d.is_1d) GY[o * C + c] = v; else GY[idx] = v;
  }
  __syncthreads();
  blk_matmul(T0, 1, GY, 0, T1, C);           // T1 = Winv^T gWinv
  blk_matmul(T1, 0, T0, 1, AUG, C);          // AUG = Winv^T gWinv Winv^T
  for (int idx = threadIdx.x; idx < CC; idx += blockDim.x) AUG[idx] = -AUG[idx];   // g_W
  __syncthreads();
  if (d.perm_kind == PERM_LU) {
    float* Pl = AUG + CC;                    // second half of AUG
    for (int idx = threadIdx.x; idx < CC; idx += blockDim.x) Pl[idx] = d.p[idx];
    __syncthreads();
    prep_lu_chain(d, g, C, Pl, AUG, W, T0, T1, T2, -G);
  } else if (d.perm_kind == PERM_DENSE) {
    // d(-log|det W|)/dW = -W^-T ; T0 still holds W^-1
    for (int idx = threadIdx.x; idx < CC; idx += blockDim.x) {
      const int i = idx / C, j = idx - i * C;
      g.g_weight[idx] += AUG[idx] - G * T0[j * C + i];
    }
  }
}

__global__ void __launch_bounds__(256) chanmix_kernel(ChanMixArgs a) {
  const CouplingBwdArgs none{};
  chanmix_body<false>(a, blockIdx.x, gridDim.x, none);
}

// ==========================================================================================
// coupling tail (affine / additive), one workgroup per sample
// ==========================================================================================


__global__ void __launch_bounds__(1024) coupling_fwd_kernel(CouplingArgs a) {
  __shared__ float sh[32];
  const int b = blockIdx.x, Ch = a.C / 2, C2 = a.C - Ch;
  const long mb = (long)b * a.HW;
  float ls = 0.f;
  for (int idx = threadIdx.x; idx < a.HW * C2; idx += blockDim.x) {
    const int p = idx / C2, j = idx % C2;
    const long m = mb + p;
    const float z2 = a.z[m * a.ldz + Ch + j];
    float out;
    if (a.additive) {
      const float h = (slab_sum(a.a3, a.S, a.slab_stride, m * a.Cout + j) + a.bias3[j]) * a.es3[j];
      if (a.hsave) a.hsave[m * a.Cout + j] = h;
      out = z2 + h;
    } else {
      const float sh_ = (slab_sum(a.a3, a.S, a.slab_stride, m * a.Cout + 2 * j) + a.bias3[2 * j]) * a.es3[2 * j];
      const float sr = (slab_sum(a.a3, a.S, a.slab_stride, m * a.Cout + 2 * j + 1) + a.bias3[2 * j + 1]) * a.es3[2 * j + 1];
      if (a.hsave) { a.hsave[m * a.Cout + 2 * j] = sh_; a.hsave[m * a.Cout + 2 * j + 1] = sr; }
      const float t = sr + 2.f;
      out = (z2 + sh_) * sigmoidf(t);
      ls += log_sigmoidf(t);
    }
    a.zout[m * a.ldo + Ch + j] = out;
  }
  for (int idx = threadIdx.x; idx < a.HW * Ch; idx += blockDim.x) {
    const int p = idx / Ch, j = idx % Ch;
    a.zout[(mb + p) * a.ldo + j] = a.z[(mb + p) * a.ldz + j];
  }
  if (a.has_ld) {
    ls = block_sum(ls, sh);
    if (threadIdx.x == 0) a.ld_out[b] = (a.ld_in ? a.ld_in[b] : 0.f) + a.dld[0] + ls;
  }
}

// inverse: given the step OUTPUT y = (z1 | z2'), recover the mixed z = (z1 | z2)
__global__ void __launch_bounds__(1024) coupling_inv_kernel(CouplingArgs a) {
  __shared__ float sh[32];
  const int b = blockIdx.x, Ch = a.C / 2, C2 = a.C - Ch;
  const long mb = (long)b * a.HW;
  float ls = 0.f;
  for (int idx = threadIdx.x; idx < a.HW * C2; idx += blockDim.x) {
    const int p = idx / C2, j = idx % C2;
    const long m = mb + p;
    const float y2 = a.z[m * a.ldz + Ch + j];
    float out;
    if (a.additive) {
      const float h = (slab_sum(a.a3, a.S, a.slab_stride, m * a.Cout + j) + a.bias3[j]) * a.es3[j];
      if (a.hsave) a.hsave[m * a.Cout + j] = h;
      out = y2 - h;
    } else {
      const float sh_ = (slab_sum(a.a3, a.S, a.slab_stride, m * a.Cout + 2 * j) + a.bias3[2 * j]) * a.es3[2 * j];
      const float sr = (slab_sum(a.a3, a.S, a.slab_stride, m * a.Cout + 2 * j + 1) + a.bias3[2 * j + 1]) * a.es3[2 * j + 1];
      if (a.hsave) { a.hsave[m * a.Cout + 2 * j] = sh_; a.hsave[m * a.Cout + 2 * j + 1] = sr; }
      const float t = sr + 2.f;
      out = y2 * (1.f + expf(-t)) - sh_;   // y2 / sigmoid(t) - shift
      ls += log_sigmoidf(t);
    }
    a.zout[m * a.ldo + Ch + j] = out;
  }
  for (int idx = threadIdx.x; idx < a.HW * Ch; idx += blockDim.x) {
    const int p = idx / Ch, j = idx % Ch;
    a.zout[(mb + p) * a.ldo + j] = a.z[(mb + p) * a.ldz + j];
  }
  if (a.has_ld) {
    ls = block_sum(ls, sh);
    if (threadIdx.x == 0) a.ld_out[b] = a.ld_in[b] - a.dld[0] - ls;
  }
}

// backward of the coupling tail

__global__ void __launch_bounds__(256) coupling_bwd_kernel(CouplingBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float acc[];  // [G][4][C2] per-group column sums
  // a.tiles workgroups per sample (pixel ranges of HW / tiles): one partial row each
  const int T = a.tiles > 0 ? a.tiles : 1, b = blockIdx.x / T, tile = blockIdx.x - b * T, PT = a.HW / T;
  const int Ch = a.C / 2, C2 = a.C - Ch;
  const long mb = (long)b * a.HW + (long)tile * PT;
  const float gl = a.gld ? a.gld[b] : 0.f;
  // thread (g, j): column j, pixels g, g+G, ...; column sums reduced over g in fixed order via LDS
  const int G = blockDim.x / C2 > 0 ? blockDim.x / C2 : 1;
  const int j = threadIdx.x % C2, g = threadIdx.x / C2;
  float sb0 = 0.f, sl0 = 0.f, sb1 = 0.f, sl1 = 0.f;
  if (g < G && (int)threadIdx.x < G * C2) {
    for (int p = g; p < PT; p += G) {
      const long m = mb + p;
      float g2 = a.gout[m * a.ldg + Ch + j];
      if (a.gext) g2 += a.gext[m * a.ldge + Ch + j];
      if (a.additive) {
        const float h = a.h[m * a.Cout + j];
        a.gz[m * a.ldgz + Ch + j] = g2;
        const float ga = g2 * a.es3[j];
        a.ga3[m * a.Cout + j] = ga;
        sb0 += ga; sl0 += g2 * h;
      } else {
        const float shf = a.h[m * a.Cout + 2 * j], sr = a.h[m * a.Cout + 2 * j + 1];
        const float sc = sigmoidf(sr + 2.f);
        const float z2 = a.z[m * a.ldz + Ch + j];
        const float gshift = g2 * sc;
        // d/dsr [ (z2+shift) sc ] = (z2+shift) sc (1-sc);  d/dsr log sc = 1 - sc
        const float gsr = g2 * (z2 + shf) * sc * (1.f - sc) + gl * (1.f - sc);
        a.gz[m * a.ldgz + Ch + j] = gshift;
        const float ga0 = gshift * a.es3[2 * j], ga1 = gsr * a.es3[2 * j + 1];
        a.ga3[m * a.Cout + 2 * j] = ga0;
        a.ga3[m * a.Cout + 2 * j + 1] = ga1;
        sb0 += ga0; sl0 += gshift * shf; sb1 += ga1; sl1 += gsr * sr;
      }
    }
    float* row = acc + (long)g * 4 * C2;
    row[j] = sb0; row[C2 + j] = sl0; row[2 * C2 + j] = sb1; row[3 * C2 + j] = sl1;
  }
  __syncthreads();
  for (int jj = threadIdx.x; jj < C2; jj += blockDim.x) {
    float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
    for (int gg = 0; gg < G; ++gg) {
      const float* row = acc + (long)gg * 4 * C2;
      t0 += row[jj]; t1 += row[C2 + jj]; t2 += row[2 * C2 + jj]; t3 += row[3 * C2 + jj];
    }
    float* part = a.part + (long)blockIdx.x * 2 * a.Cout;
    if (a.additive) { part[jj] = t0; part[a.Cout + jj] = 3.f * t1; }
    else {
      part[2 * jj] = t0; part[2 * jj + 1] = t2;
      part[a.Cout + 2 * jj] = 3.f * t1; part[a.Cout + 2 * jj + 1] = 3.f * t3;
    }
  }
}

// backward of the INVERSE coupling tail (reverse pass with gradients, pl_module.py:229-243 on image flows):
//   z2 = y2 / sigmoid(sr + 2) - shift   (additive: z2 = y2 - h),   logdet_out = logdet_in - dld - sum log sigmoid(sr + 2)
// a.gout = dL/dz (the mixed tensor the inverse channel mix reads), a.z = the step's saved input y, a.h = (shift | sr) kept
// by coupling_inv_kernel; writes dL/dy2 into a.gz's second half, the conv-net output gradient g_a3 and the same partial
// rows as coupling_bwd_kernel.  (The first half of dL/dy is dL/dz1 + the coupling net's input gradient: added by the caller.)
__global__ void __launch_bounds__(256) coupling_inv_bwd_kernel(CouplingBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float acc[];
  const int b = blockIdx.x, Ch = a.C / 2, C2 = a.C - Ch;
  const long mb = (long)b * a.HW;
  const float gl = a.gld ? a.gld[b] : 0.f;
  const int G = blockDim.x / C2 > 0 ? blockDim.x / C2 : 1;
  const int j = threadIdx.x % C2, g = threadIdx.x / C2;
  float sb0 = 0.f, sl0 = 0.f, sb1 = 0.f, sl1 = 0.f;
  if (g < G && (int)threadIdx.x < G * C2) {
    for (int p = g; p < a.HW; p += G) {
      const long m = mb + p;
      const float g2 = a.gout[m * a.ldg + Ch + j];
      if (a.additive) {
        const float h = a.h[m * a.Cout + j];
        a.gz[m * a.ldgz + Ch + j] = g2;
        const float ga = -g2 * a.es3[j];
        a.ga3[m * a.Cout + j] = ga;
        sb0 += ga; sl0 += -g2 * h;
      } else {
        const float shf = a.h[m * a.Cout + 2 * j], sr = a.h[m * a.Cout + 2 * j + 1];
        const float isc = 1.f + expf(-(sr + 2.f)), sc = 1.f / isc;   // 1 / sigmoid, sigmoid
        const float y2 = a.z[m * a.ldz + Ch + j];
        const float gshift = -g2;
        // d/dsr [ y2 / sc ] = -y2 (1 - sc) / sc ;  d/dsr [ -log sc ] = -(1 - sc)
        const float gsr = -(g2 * y2 * isc + gl) * (1.f - sc);
        a.gz[m * a.ldgz + Ch + j] = g2 * isc;
        const float ga0 = gshift * a.es3[2 * j], ga1 = gsr * a.es3[2 * j + 1];
        a.ga3[m * a.Cout + 2 * j] = ga0;
        a.ga3[m * a.Cout + 2 * j + 1] = ga1;
        sb0 += ga0; sl0 += gshift * shf; sb1 += ga1; sl1 += gsr * sr;
      }
    }
    float* row = acc + (long)g * 4 * C2;
    row[j] = sb0; row[C2 + j] = sl0; row[2 * C2 + j] = sb1; row[3 * C2 + j] = sl1;
  }
  __syncthreads();
  for (int jj = threadIdx.x; jj < C2; jj += blockDim.x) {
    float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
    for (int gg = 0; gg < G; ++gg) {
      const float* row = acc + (long)gg * 4 * C2;
      t0 += row[jj]; t1 += row[C2 + jj]; t2 += row[2 * C2 + jj]; t3 += row[3 * C2 + jj];
    }
    float* part = a.part + (long)b * 2 * a.Cout;
    if (a.additive) { part[jj] = t0; part[a.Cout + jj] = 3.f * t1; }
    else {
      part[2 * jj] = t0; part[2 * jj + 1] = t2;
      part[a.Cout + 2 * jj] = 3.f * t1; part[a.Cout + 2 * jj + 1] = 3.f * t3;
    }
  }
}

// ==========================================================================================
// squeeze / unsqueeze (space-to-depth by 2)
// ==========================================================================================
// NCHW input (model boundary) -> NHWC squeezed; optional in-place dequantisation noise.
__global__ void squeeze_nchw_kernel(float* __restrict__ x, const float* __restrict__ noise, float* __restrict__ out,
                                    int B, int C, int H, int W) {
  const long total = (long)B * C * H * W;
  const int Ho = H / 2, Wo = W / 2;
  for (long t = blockIdx.x * (long)blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const int w = t % W; long r = t / W;
    const int h = r % H; r /= H;
    const int c = r % C; const int b = r / C;
    float v = x[t];
    if (noise) { v += noise[t]; x[t] = v; }
    const int co = c * 4 + (h & 1) * 2 + (w & 1);
    out[(((long)b * Ho + (h >> 1)) * Wo + (w >> 1)) * (4 * C) + co] = v;
  }
}
// NHWC -> NHWC squeezed
__global__ void squeeze_nhwc_kernel(const float* __restrict__ x, int ldx, float* __restrict__ out, int B, int C,
                                    int H, int W) {
  const int Ho = H / 2, Wo = W / 2, Co = 4 * C;
  const long total = (long)B * Ho * Wo * Co;
  for (long t = blockIdx.x * (long)blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const int co = t % Co; long r = t / Co;
    const int wo = r % Wo; r /= Wo;
    const int ho = r % Ho; const int b = r / Ho;
    const int c = co >> 2, fh = (co >> 1) & 1, fw = co & 1;
    out[t] = x[(((long)b * H + 2 * ho + fh) * W + 2 * wo + fw) * ldx + c];
  }
}
// NHWC squeezed -> NHWC unsqueezed (also the backward of squeeze_nhwc)
__global__ void unsqueeze_nhwc_kernel(const float* __restrict__ y, const float* __restrict__ y2, float* __restrict__ out, int B,
                                      int C, int H, int W) {  // (C,H,W) = UNsqueezed dims; y2: optional second addend
  const int Ho = H / 2, Wo = W / 2, Co = 4 * C;
  const long total = (long)B * H * W * C;
  for (long t = blockIdx.x * (long)blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const int c = t % C; long r = t / C;
    const int w = r % W; r /= W;
    const int h = r % H; const int b = r / H;
    const long src = (((long)b * Ho + (h >> 1)) * Wo + (w >> 1)) * Co + c * 4 + (h & 1) * 2 + (w & 1);
    out[t] = y2 ? y[src] + y2[src] : y[src];
  }
}
// NHWC squeezed -> NCHW unsqueezed (model boundary of the reverse pass / grad wrt input image)
__global__ void unsqueeze_to_nchw_kernel(const float* __restrict__ y, const float* __restrict__ y2, float* __restrict__ out,
                                         int B, int C, int H, int W) {
  const int Ho = H / 2, Wo = W / 2, Co = 4 * C;
  const long total = (long)B * C * H * W;
  for (long t = blockIdx.x * (long)blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const int w = t % W; long r = t / W;
    const int h = r % H; r /= H;
    const int c = r % C; const int b = r / C;
    const long src = (((long)b * Ho + (h >> 1)) * Wo + (w >> 1)) * Co + c * 4 + (h & 1) * 2 + (w & 1);
    out[t] = y2 ? y[src] + y2[src] : y[src];
  }
}

// ==========================================================================================
// Split2d prior tail
// ==========================================================================================
constexpr float LN_2PI = 1.8378770664093453f;

__global__ void __launch_bounds__(256) split_fwd_kernel(SplitArgs a) {
  __shared__ float sh[32];
  const int b = blockIdx.x, Ch = a.C / 2;
  const long mb = (long)b * a.HW;
  float lp = 0.f;
  for (int idx = threadIdx.x; idx < a.HW * Ch; idx += blockDim.x) {
    const int p = idx / Ch, j = idx % Ch;
    const long m = mb + p;
    const float mean = (slab_sum(a.a, a.S, a.slab_stride, m * a.C + 2 * j) + a.bias[2 * j]) * a.es[2 * j];
    const float logs = (slab_sum(a.a, a.S, a.slab_stride, m * a.C + 2 * j + 1) + a.bias[2 * j + 1]) * a.es[2 * j + 1];
    if (a.hsave) { a.hsave[m * a.C + 2 * j] = mean; a.hsave[m * a.C + 2 * j + 1] = logs; }
    const float d = a.x[m * a.ldx + Ch + j] - mean;
    lp += -0.5f * (2.f * logs + d * d * expf(-2.f * logs) + LN_2PI);
    a.z1out[m * Ch + j] = a.x[m * a.ldx + j];
  }
  if (a.has_ld) {
    lp = block_sum(lp, sh);
    if (threadIdx.x == 0) a.ld_out[b] = a.ld_in[b] + lp;
  }
}
// inverse: x = z1 [M][Ch] (ldx = Ch); writes full [M][C] = (z1 | mean + e^{logs} T eps) into z1out
__global__ void __launch_bounds__(256) split_inv_kernel(SplitArgs a) {
  const int b = blockIdx.x, Ch = a.C / 2;
  const long mb = (long)b * a.HW;
  for (int idx = threadIdx.x; idx < a.HW * Ch; idx += blockDim.x) {
    const int p = idx / Ch, j = idx % Ch;
    const long m = mb + p;
    const float mean = (slab_sum(a.a, a.S, a.slab_stride, m * a.C + 2 * j) + a.bias[2 * j]) * a.es[2 * j];
    const float logs = (slab_sum(a.a, a.S, a.slab_stride, m * a.C + 2 * j + 1) + a.bias[2 * j + 1]) * a.es[2 * j + 1];
    const float e = a.eps ? a.eps[m * Ch + j] : 0.f;
    a.z1out[m * a.C + j] = a.x[m * a.ldx + j];
    a.z1out[m * a.C + Ch + j] = mean + expf(logs) * a.temperature * e;
  }
}

__global__ void __launch_bounds__(256) split_bwd_kernel(SplitBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float acc[];
  const int b = blockIdx.x, Ch = a.C / 2;
  const long mb = (long)b * a.HW;
  const float gl = a.gld[b];
  const int G = blockDim.x / Ch > 0 ? blockDim.x / Ch : 1;
  const int j = threadIdx.x % Ch, g = threadIdx.x / Ch;
  if (g < G && (int)threadIdx.x < G * Ch) {
    float sb0 = 0.f, sl0 = 0.f, sb1 = 0.f, sl1 = 0.f;
    for (int p = g; p < a.HW; p += G) {
      const long m = mb + p;
      const float mean = a.h[m * a.C + 2 * j], logs = a.h[m * a.C + 2 * j + 1];
      const float d = a.x[m * a.ldx + Ch + j] - mean, iv = expf(-2.f * logs);
      const float gmean = gl * d * iv, glogs = gl * (d * d * iv - 1.f);
      a.gx[m * a.ldgx + Ch + j] = -gmean;
      const float ga0 = gmean * a.es[2 * j], ga1 = glogs * a.es[2 * j + 1];
      a.ga[m * a.C + 2 * j] = ga0;
      a.ga[m * a.C + 2 * j + 1] = ga1;
      sb0 += ga0; sl0 += gmean * mean; sb1 += ga1; sl1 += glogs * logs;
    }
    float* row = acc + (long)g * 4 * Ch;
    row[j] = sb0; row[Ch + j] = sl0; row[2 * Ch + j] = sb1; row[3 * Ch + j] = sl1;
  }
  __syncthreads();
  for (int jj = threadIdx.x; jj < Ch; jj += blockDim.x) {
    float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
    for (int gg = 0; gg < G; ++gg) {
      const float* row = acc + (long)gg * 4 * Ch;
      t0 += row[jj]; t1 += row[Ch + jj]; t2 += row[2 * Ch + jj]; t3 += row[3 * Ch + jj];
    }
    float* part = a.part + (long)b * 2 * a.C;
    part[2 * jj] = t0; part[2 * jj + 1] = t2;
    part[a.C + 2 * jj] = 3.f * t1; part[a.C + 2 * jj + 1] = 3.f * t3;
  }
}

// out[m][c] = g[m][c] (+ gext[m][c]) + sum_s slab[s][m][c]   for c < Ch  (first-half grads)
__global__ void add_slabs_kernel(const float* __restrict__ g, int ldg, const float* __restrict__ gext, int ldge,
                                 const float* __restrict__ slab, int S, long slab_stride, float* __restrict__ out,
                                 int ldo, long M, int Ch) {
  const long total = M * Ch;
  for (long t = blockIdx.x * (long)blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const long m = t / Ch; const int c = t % Ch;
    float v = g ? g[m * ldg + c] : 0.f;
    if (gext) v += gext[m * ldge + c];
    v += slab_sum(slab, S, slab_stride, t);
    out[m * ldo + c] = v;
  }
}

// ==========================================================================================
// top prior: objective = logdet + sum log N(z; mean, e^{logs}); bits/dim
// ==========================================================================================
__global__ void __launch_bounds__(256) prior_fwd_kernel(PriorArgs a) {
  __shared__ float sh[32];
  const int b = blockIdx.x;
  const float* ph = a.ph + b * a.ph_bstride;
  float lp = 0.f;
  for (int idx = threadIdx.x; idx < a.HW * a.C; idx += blockDim.x) {
    const int p = idx / a.C, c = idx % a.C;
    const float mean = ph[(long)c * a.HW + p], logs = ph[(long)(a.C + c) * a.HW + p];
    const float d = a.z[((long)b * a.HW + p) * a.ldz + c] - mean;
    lp += -0.5f * (2.f * logs + d * d * expf(-2.f * logs) + LN_2PI);
  }
  lp = block_sum(lp, sh);
  if (threadIdx.x == 0) {
    const float obj = a.ld[b] + lp;
    a.objective[b] = obj;
    a.bpd[b] = -obj * a.bpd_scale;
  }
}
__global__ void __launch_bounds__(256) prior_bwd_kernel(PriorBwdArgs a) {
  const int b = blockIdx.x;
  const float* ph = a.ph + b * a.ph_bstride;
  const float gobj = -a.gbpd[b] * a.bpd_scale;
  if (threadIdx.x == 0) a.gld[b] = gobj;
  for (int idx = threadIdx.x; idx < a.HW * a.C; idx += blockDim.x) {
    const int p = idx / a.C, c = idx % a.C;
    const float mean = ph[(long)c * a.HW + p], logs = ph[(long)(a.C + c) * a.HW + p];
    const float iv = expf(-2.f * logs), d = a.z[((long)b * a.HW + p) * a.ldz + c] - mean;
    a.gz[((long)b * a.HW + p) * a.ldgz + c] = -gobj * d * iv;
    if (a.gph) {
      a.gph[((long)b * 2 * a.C + c) * a.HW + p] = gobj * d * iv;
      a.gph[((long)b * 2 * a.C + a.C + c) * a.HW + p] = gobj * (d * d * iv - 1.f);
    }
  }
}

// Finalise the small per-channel parameter gradients of one coupling net (accumulates):
//   colsum layout from conv_gemm EPI_DGRAD: [mblocks][2][N] = (sum acc*h , sum masked*scale)
//   tail layout from coupling_bwd / split_bwd: [B][2][Cout] = (sum g_a3, 3 sum g_h h)
// One block = 32 columns x 8 row-groups; rows (partials) are summed per row-group, then over the
// 8 groups in fixed order through LDS.  blockIdx.y selects the segment (layer-1, layer-2, tail).
__global__ void __launch_bounds__(256) small_grads_kernel(SmallGradArgs a) {
  __shared__ float red[2][8][32];
  const int seg = blockIdx.y;
  const float* src; int rows, n; float *g0, *g1;
  if (seg == 0) { src = a.col1; rows = a.mb1; n = a.n1; g0 = a.g_l1; g1 = a.g_b1; }
  else if (seg == 1) { src = a.col2; rows = a.mb2; n = a.n2; g0 = a.g_l2; g1 = a.g_b2; }
  else { src = a.tail; rows = a.B; n = a.n3; g0 = a.g_bias3; g1 = a.g_l3; }
  const int col = blockIdx.x * 32 + (threadIdx.x & 31), rg = threadIdx.x >> 5;
  if (blockIdx.x * 32 >= n) return;
  float s0 = 0.f, s1 = 0.f;
  if (col < n)
    partial_rows_sum(src, rows, n, col, rg, s0, s1);
  red[0][rg][threadIdx.x & 31] = s0;
  red[1][rg][threadIdx.x & 31] = s1;
  __syncthreads();
  if (rg < 2 && col < n) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += red[rg][k][threadIdx.x & 31];
    (rg == 0 ? g0 : g1)[col] += t;
  }
}

__global__ void __launch_bounds__(256) step_tail_kernel(SmallGradArgs a, StepTailArgs f) { step_tail_body(a, f, blockIdx.x); }
// The step tail and the step's ActNorm / invconv backward (chanmix) both only wait for the last GEMM launch and do not
// depend on each other: one launch, the first n_tail workgroups run the tail, the rest the channel mix.
// The channel-mix workgroups come first in the grid: they are few and each is a chain of dependent loads, so they should
// not queue behind the (many, short) slab-reduction workgroups.
__global__ void __launch_bounds__(256) tail_chanmix_kernel(SmallGradArgs a, StepTailArgs f, int n_cm, ChanMixArgs m,
                                                           CouplingBwdArgs nx) {
  if ((int)blockIdx.x < n_cm) {
    if (nx.ga3) chanmix_body<true>(m, blockIdx.x, n_cm, nx);
    else chanmix_body<false>(m, blockIdx.x, n_cm, nx);
  } else step_tail_body(a, f, blockIdx.x - n_cm);
}

// ==========================================================================================
// host launchers
// ==========================================================================================
static inline int ew_blocks(long total) {
  long b = (total + 255) / 256;
  return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

// threads for a one-workgroup-per-sample kernel over n elements
static inline int sample_threads(int n) { return n >= 1024 ? 1024 : (n > 256 ? ((n + 63) / 64) * 64 : 256); }

int launch_prep_repack(const PrepDesc* descs_dev, int nsteps, int want_inverse, const nfk_repack_desc* rf32, int n_f32,
                       const nfk_repack_desc* rbf16, int n_bf16, hipStream_t st) {
  static_assert(PREP_LDS >= 2 * RP_LDS_FLOATS * sizeof(float), "two repack tiles live in the prep launch's LDS");
  static_assert(PR_THREADS == 512, "a repack workgroup is two 256-thread workers");
  const int grid = nsteps + 32 * (n_f32 + n_bf16);
  if (grid <= 0) return NFK_OK;
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(prep_repack_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)PREP_LDS) != hipSuccess)
      return NFK_ERR_LAUNCH;
    attr = true;
  }
  hipLaunchKernelGGL(prep_repack_kernel, dim3(grid), dim3(PR_THREADS), PREP_LDS, st, descs_dev, nsteps, want_inverse, rf32, n_f32,
                     rbf16);
  NFK_LAUNCH_CHECK();
  return NFK_OK;
}
int launch_prep(const PrepDesc* descs_dev, int nsteps, int want_inverse, hipStream_t st) {
  if (nsteps <= 0) return NFK_OK;
  hipLaunchKernelGGL(prep_kernel, dim3(nsteps), dim3(256), PREP_LDS, st, descs_dev, want_inverse);
  NFK_LAUNCH_CHECK();
  return NFK_OK;
}
int launch_prep_bwd(const PrepDesc* descs_dev, const PrepGrad* grads_dev, int nsteps, const float* gld, int B,
                    int any_dense, hipStream_t st) {
  if (nsteps <= 0) return NFK_OK;
  // C <= 192 covers every supported width: (192*192 + 192) / 256 = 145 chunks
  hipLaunchKernelGGL(part_reduce_kernel, dim3(145, nsteps), dim3(256), 0, st, descs_dev, grads_dev);
  NFK_LAUNCH_CHECK();
  hipLaunchKernelGGL(prep_bwd_kernel, dim3(nsteps), dim3(256), PREP_BWD_LDS, st, descs_dev, grads_dev, gld, B);
  NFK_LAUNCH_CHECK();
  if (any_dense) {
    hipLaunchKernelGGL(prep_bwd_dense_kernel, dim3(nsteps), dim3(256), 0, st, descs_dev, grads_dev, gld, B);
    NFK_LAUNCH_CHECK();
  }
  return NFK_OK;
}

int launch_prep_inv_bwd(const PrepDesc* descs_dev, const PrepGrad* grads_dev, int nsteps, const float* gld, int B,
                        hipStream_t st) {
  if (nsteps <= 0) return NFK_OK;
  hipLaunchKernelGGL(part_reduce_kernel, dim3(145, nsteps), dim3(256), 0, st, descs_dev, grads_dev);
  NFK_LAUNCH_CHECK();
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(prep_inv_bwd_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)PREP_INV_BWD_LDS) != hipSuccess)
      return NFK_ERR_LAUNCH;
    attr = true;
  }
  hipLaunchKernelGGL(prep_inv_bwd_kernel, dim3(nsteps), dim3(256), PREP_INV_BWD_LDS, st, descs_dev, grads_dev, gld, B);
  NFK_LAUNCH_CHECK();
  return NFK_OK;
}

int launch_chanmix(const ChanMixArgs& a, hipStream_t st) {
  if (a.M <= 0) return NFK_OK;
  const int C = a.C, Cp = (C + 3) & ~3, TP = cm_tile_pixels(C);
  const size_t lds = (size_t)(C * Cp + 2 * TP * (C + 1) + (a.part ? C * C + C : 0)) * sizeof(float);
  if (lds > 160 * 1024) return NFK_ERR_ARG;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(chanmix_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  int blocks = ceil_div(a.M, TP);
  if (a.part && blocks > CM_MAX_BLOCKS) blocks = CM_MAX_BLOCKS;  // == number of partial slabs written
  hipLaunchKernelGGL(chanmix_kernel, dim3(blocks), dim3(256), lds, st, a);
  NFK_LAUNCH_CHECK();
  return NFK_OK;
}
// number of partial slabs nfk_step2d_bwd writes for (M, C)
int chanmix_parts(int M, int C) {
  const int b = ceil_div(M, cm_tile_pixels(C));
  return b < CM_MAX_BLOCKS ? b : CM_MAX_BLOCKS;
}

int launch_coupling_fwd(const CouplingArgs& a, int B, hipStream_t st) {
  hipLaunchKernelGGL(coupling_fwd_kernel, dim3(B), dim3(sample_threads(a.HW * (a.C - a.C / 2))), 0, st, a);
  NFK_LAUNCH_CHECK();
  return NFK_OK;
}
int launch_coupling_inv(const CouplingArgs& a, int B, hipStream_t st) {
  hipLaunchKernelGGL(coupling_inv_kernel, dim3(B), dim3(sample_threads(a.HW * (a.C - a.C / 2))), 0, st, a);
  NFK_LAUNCH_CHECK();
  return NFK_OK;
}
// partial rows a stand-alone coupling backward writes: B * tiles (<= 256: the step's `tail` workspace)
int coupling_bwd_tiles(int B, int HW) {
  int T = 1;
  while (T < 4 && B * T * 2 <= 256 && HW % (T * 2) == 0 && HW / (T * 2) >= 16) T *= 2;
  return T;
}
int launch_coupling_bwd(const CouplingBwdArgs& a_in, int B, hipStream_t st) {
  {
    CouplingBwdArgs a = a_in;
    const int C2 = a.C - a.C / 2;
    if (C2 > 256) return NFK_ERR_ARG;
    if (a.tiles <= 0) a.tiles = 1;
    hipLaunchKernelGGL(coupling_bwd_kernel, dim3(B * a.tiles), dim3(256), (size_t)(256 / C2) * 4 * C2 * sizeof(float), st, a);
  }
  NFK_LAUNCH_CHECK();
  return NFK_OK;
}
int launch_coupling_inv_bwd(const CouplingBwdArgs& a, int B, hipStream_t st) {
  const int C2 = a.C - a.C / 2;
  if (C2 > 256) return NFK_ERR_ARG;
  hipLaunchKernelGGL(coupling_inv_bwd_kernel, dim3(B), dim3(256), (size_t)(256 / C2) * 4 * C2 * sizeof(float), st, a);
  NFK_LAUNCH_CHECK();
  return NFK_OK;
}
int launch_squeeze_nchw(float* x, const float* noise, float* out, int B, int C, int H, int W, hipStream_t st) {
  if ((H | W) & 1) return NFK_ERR_ARG;
  hipLaunchKernelGGL(squeeze_nchw_kernel, dim3(ew_blocks((long)B * C * H * W)), dim3(256), 0, st, x, noise, out,
                     B, C, H, W);
  NFK_LAUNCH_CHECK();
  return NFK_OK;
}
int launch_squeeze_nhwc(const float* x, int ldx, float* out, int B, int C, int H, int W, hipStream_t st) {
  if ((H | W) & 1) return NFK_ERR_ARG;
  hipLaunchKernelGGL(squeeze_nhwc_kernel, dim3(ew_blocks((long)B * C * H * W)), dim3(256), 0, st, x, ldx, out,
                     B, C, H, W);
  NFK_LAUNCH_CHECK();
  return NFK_OK;
}
int launch_unsqueeze(const float* y, const float* y2, float* out, int B, int C, int H, int W, int to_nchw, hipStream_t st) {
  if ((H | W) & 1) return NFK_ERR_ARG;
  if (to_nchw)
    hipLaunchKernelGGL(unsqueeze_to_nchw_kernel, dim3(ew_blocks((long)B * C * H * W)), dim3(256), 0, st, y, y2, out,
                       B, C, H, W);
  else
    hipLaunchKernelGGL(unsqueeze_nhwc_kernel, dim3(ew_blocks((long)B * C * H * W)), dim3(256), 0, st, y, y2, out, B,
                       C, H, W);
  NFK_LAUNCH_CHECK();
  return NFK_OK;
}
int launch_split_fwd(const SplitArgs& a, int B, hipStream_t st) {
  hipLaunchKernelGGL(split_fwd_kernel, dim3(B), dim3(256), 0, st, a);
  NFK_LAUNCH_CHECK();
  return NFK_OK;
}
int launch_split_inv(const SplitArgs& a, int B, hipStream_t st) {
  hipLaunchKernelGGL(split_inv_kernel, dim3(B), dim3(256), 0, st, a);
  NFK_LAUNCH_CHECK();
  return NFK_OK;
}
int launch_split_bwd(const SplitBwdArgs& a, int B, hipStream_t st) {
  {
    const int Ch = a.C / 2;
    if (Ch > 256) return NFK_ERR_ARG;
    hipLaunchKernelGGL(split_bwd_kernel, dim3(B), dim3(256), (size_t)(256 / Ch) * 4 * Ch * sizeof(float), st, a);
  }
  NFK_LAUNCH_CHECK();
  return NFK_OK;
}
int launch_add_slabs(const float* g, int ldg, const float* gext, int ldge, const float* slab, int S,
                     long slab_stride, float* out, int ldo, long M, int Ch, hipStream_t st) {
  hipLaunchKernelGGL(add_slabs_kernel, dim3(ew_blocks(M * Ch)), dim3(256), 0, st, g, ldg, gext, ldge, slab, S,
                     slab_stride, out, ldo, M, Ch);
  NFK_LAUNCH_CHECK();
  return NFK_OK;
}
int launch_prior_fwd(const PriorArgs& a, int B, hipStream_t st) {
  hipLaunchKernelGGL(prior_fwd_kernel, dim3(B), dim3(256), 0, st, a);
  NFK_LAUNCH_CHECK();
  return NFK_OK;
}
int launch_prior_bwd(const PriorBwdArgs& a, int B, hipStream_t st) {
  hipLaunchKernelGGL(prior_bwd_kernel, dim3(B), dim3(256), 0, st, a);
  NFK_LAUNCH_CHECK();
  return NFK_OK;
}
int launch_small_grads(const SmallGradArgs& a, hipStream_t st) {
  int n = a.n1 > a.n2 ? a.n1 : a.n2;
  if (a.n3 > n) n = a.n3;
  if (n <= 0) return NFK_OK;
  hipLaunchKernelGGL(small_grads_kernel, dim3(ceil_div(n, 32), 3), dim3(256), 0, st, a);
  NFK_LAUNCH_CHECK();
  return NFK_OK;
}

int launch_tail_chanmix(const SmallGradArgs& a, StepTailArgs f, const ChanMixArgs& m, const CouplingBwdArgs* next,
                        hipStream_t st) {
  const int C = m.C, Cp = (C + 3) & ~3, TP = cm_tile_pixels(C), C2 = C - C / 2;
  size_t lds = (size_t)(C * Cp + 2 * TP * (C + 1) + (m.part ? C * C + C : 0)) * sizeof(float);
  CouplingBwdArgs nx{};
  if (next) {
    if (C2 < 1 || C2 > 256) return NFK_ERR_ARG;
    nx = *next;
    lds += (size_t)(TP * (C + 1) + (256 / C2) * 4 * C2 + 4 * C2) * sizeof(float);
  }
  if (lds > 64 * 1024 || m.M <= 0) return NFK_ERR_ARG;
  int cb = ceil_div(m.M, TP);
  if (m.part && cb > CM_MAX_BLOCKS) cb = CM_MAX_BLOCKS;
  const int nt = tail_blocks(a, f);
  hipLaunchKernelGGL(tail_chanmix_kernel, dim3(cb + nt), dim3(256), lds, st, a, f, cb, m, nx);
  NFK_LAUNCH_CHECK();
  return NFK_OK;
}
int launch_step_tail(const SmallGradArgs& a, StepTailArgs f, hipStream_t st) {
  int n = a.n1 > a.n2 ? a.n1 : a.n2;
  if (a.n3 > n) n = a.n3;
  f.sg_blocks = ceil_div(n > 0 ? n : 1, 32);
  int total = f.sg_blocks * 3;
  for (int k = 0; k < 3; ++k) {
    const long e = (long)f.Cout[k] * f.Cin[k] * f.taps[k];
    int b = f.S[k] > 0 ? (int)((e + 255) / 256) : 0;
    if (b > 1024) b = 1024;
    f.blocks[k] = b;
    total += b;
  }
  hipLaunchKernelGGL(step_tail_kernel, dim3(total), dim3(256), 0, st, a, f);
  NFK_LAUNCH_CHECK();
  return NFK_OK;
}

}  // namespace nfk
