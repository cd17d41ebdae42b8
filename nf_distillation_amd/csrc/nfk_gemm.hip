// nfk GEMM engine: the coupling network's 3x3 / 1x1 convolutions as implicit GEMMs on the
// gfx950 matrix cores (exact-f32 MFMA, v_mfma_f32_16x16x4_f32), NHWC activations.
//
// Replaces, for the reference's `Conv2d` / `Conv2dZeros` (models/layers.py:190-260) inside
// `get_block_2d` (models/flows.py:25-34) and `Split2d.conv` (models/layers.py:296):
//   forward   : ZeroPad2d + F.conv2d (+ ActNorm2d + ReLU fused as an epilogue)
//   backward  : dgrad (a 3x3 SAME conv with flipped taps) and wgrad (reduction over pixels)
//
// One MFMA core serves all three: D[i][j] = sum_k A[i][k] * B[j][k] with both operand tiles
// staged in LDS k-contiguous ([rows][32+4 pad]) and read with ds_read_b128, 4 MFMAs per read.
//   conv/dgrad : A rows = output channels (repacked weights [N][Kp]), B rows = pixels (im2col
//                gathered on the fly from NHWC, zero padding folded into the address predicate)
//   wgrad      : A rows = channels of P, B rows = (tap,channel) of Q, k = pixel (transposed
//                while staging)
// The accumulator then holds 4 consecutive channels of one pixel per lane, so epilogues are
// float4-wide and per-channel parameters load as float4.
#include "nfk_internal.h"
#include "nfk_repack.h"
#include <vector>

namespace nfk {

constexpr int BK = 32;       // k-chunk per LDS stage
constexpr int LS = BK + 4;   // LDS row stride (floats): 144 B keeps ds_read_b128 rows 16-B aligned



// Buffer-resource loads (hardware bounds check: offset >= num_records reads as 0).
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
constexpr unsigned BUF_OOB = 0x80000000u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 buf_load4(__amdgpu_buffer_rsrc_t r, unsigned off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0));
}
__device__ __forceinline__ float buf_load1(__amdgpu_buffer_rsrc_t r, unsigned off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, 0));
}

// XCD-aware block remap (bijective): blocks that share an XCD (id % 8) get a contiguous range.
__device__ __forceinline__ int xcd_remap(int id, int n) {
  const int q = n >> 3, r = n & 7, x = id & 7, s = id >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + s;
}

// LDS tile addressing.  SWZ=false: rows padded to 36 floats (2-way conflicts on ds_read_b128).
// SWZ=true: 32-float rows, 16-byte slot index XORed with (row>>1)&7 -- conflict-free for the
// ds_read_b128 lane groups of gfx950 and 11 % less LDS.  Tile row bases are multiples of 16.
template <bool SWZ>
__device__ __forceinline__ int lds_off(int row, int slot) {
  return SWZ ? row * BK + ((slot ^ ((row >> 1) & 7)) << 2) : row * LS + (slot << 2);
}

template <int TN, int TM, bool SWZ>
__device__ __forceinline__ void mfma_tile_step(const float* __restrict__ As, const float* __restrict__ Bs,
                                               int arow0, int brow0, int lane, f32x4 (&acc)[TN][TM]) {
  const int r = lane & 15, q = lane >> 4;
#pragma unroll
  for (int kh = 0; kh < 2; ++kh) {
    f32x4 a[TN], b[TM];
#pragma unroll
    for (int t = 0; t < TN; ++t)
      a[t] = *reinterpret_cast<const f32x4*>(As + lds_off<SWZ>(arow0 + t * 16 + r, kh * 4 + q));
#pragma unroll
    for (int t = 0; t < TM; ++t)
      b[t] = *reinterpret_cast<const f32x4*>(Bs + lds_off<SWZ>(brow0 + t * 16 + r, kh * 4 + q));
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
          acc[tn][tm] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[tn][j], b[tm][j], acc[tn][tm], 0, 0, 0);
  }
}

// Same MFMA schedule for tiles stored k-major: T[k][rows + 4 pad] (k = pixel), read with ds_read_b32.
// MFMA jj of half kh consumes k = 16*kh + 4*(lane>>4) + jj on both operands.
template <int TN, int TM>
__device__ __forceinline__ void mfma_tile_step_kmajor(const float* __restrict__ As, int lda, const float* __restrict__ Bs,
                                                      int ldb, int arow0, int brow0, int lane, f32x4 (&acc)[TN][TM]) {
  const int r = lane & 15, kq = (lane >> 4) * 4;
#pragma unroll
  for (int kh = 0; kh < 2; ++kh) {
    f32x4 a[TN], b[TM];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = kh * 16 + kq + j;
#pragma unroll
      for (int t = 0; t < TN; ++t) a[t][j] = As[k * lda + arow0 + t * 16 + r];
#pragma unroll
      for (int t = 0; t < TM; ++t) b[t][j] = Bs[k * ldb + brow0 + t * 16 + r];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
          acc[tn][tm] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[tn][j], b[tm][j], acc[tn][tm], 0, 0, 0);
  }
}

// ------------------------------------------------------------------------------------------
// conv / dgrad kernel
// ------------------------------------------------------------------------------------------
template <int TN, int TM, int WN, int WM, bool SWZ>
__global__ void __launch_bounds__(256) conv_gemm_kernel(ConvGemmArgs a) {
  static_assert(WN * WM == 4, "256-thread workgroups");
  constexpr int BN = 16 * TN * WN, BM = 16 * TM * WM;
  constexpr int LSK = SWZ ? BK : LS;
  constexpr int WL = (BN * 8 + 255) / 256, XL = BM / 32;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wn = wid % WN, wm = wid / WN;
  const int nNB = (a.N + BN - 1) / BN, nMB = (a.M + BM - 1) / BM;
  const int bid = xcd_remap(blockIdx.x, nNB * nMB);
  const int n0 = (bid % nNB) * BN, m0 = (bid / nNB) * BM;
  const int HW = a.H * a.W;
  const bool vec = ((a.Cin | a.ldx) & 3) == 0;

  // split-K range (chunks of 32)
  const int S = gridDim.y, z = blockIdx.y;
  const int c_begin = (int)((long)a.nkchunks * z / S), c_end = (int)((long)a.nkchunks * (z + 1) / S);

  // Operands are fetched with buffer loads: an offset >= num_records returns 0, so SAME zero padding,
  // the k tail and the row/column tails of the last tiles need no branches (one v_cndmask per load).
  const __amdgpu_buffer_rsrc_t rX = make_rsrc(a.X, (unsigned)((long)a.M * a.ldx * 4));
  const __amdgpu_buffer_rsrc_t rW = make_rsrc(a.Wt, (unsigned)((long)a.N * a.Kp * 4));
  const int lq = tid & 7, lr = tid >> 3;
  unsigned rowoff[XL], woff[WL];
  int yx[XL];
#pragma unroll
  for (int i = 0; i < XL; ++i) {
    const int m = m0 + lr + 32 * i;
    const int rem = m % HW;
    rowoff[i] = m < a.M ? (unsigned)m * (unsigned)a.ldx * 4u : BUF_OOB;
    yx[i] = ((rem / a.W) << 16) | (rem % a.W);
  }
#pragma unroll
  for (int i = 0; i < WL; ++i) {
    const int r = (tid + 256 * i) >> 3, n = n0 + r;
    woff[i] = (r < BN && n < a.N) ? (unsigned)n * (unsigned)a.Kp * 4u : BUF_OOB;
  }
  const bool unif = vec && (a.Cin % BK) == 0;  // a k-chunk never straddles a tap: tap is wave-uniform

  f32x4 wreg[WL], xreg[XL];
  auto tap_shift = [&](int tap, int& dy, int& dx) {
    dy = (a.taps == 1) ? 0 : tap / 3 - 1;
    dx = (a.taps == 1) ? 0 : tap % 3 - 1;
  };
  auto gload = [&](int chunk) {
    const int k = chunk * BK + lq * 4;
#pragma unroll
    for (int i = 0; i < WL; ++i) wreg[i] = buf_load4(rW, woff[i] + (unsigned)k * 4u);
    if (unif) {
      const int kb = chunk * BK;
      const int tap = (a.taps == 1) ? 0 : kb / a.Cin;
      int dy, dx;
      tap_shift(tap, dy, dx);
      const unsigned add = (unsigned)((dy * a.W + dx) * a.ldx + (kb - tap * a.Cin) + lq * 4) * 4u;
#pragma unroll
      for (int i = 0; i < XL; ++i) {
        const int yy = (yx[i] >> 16) + dy, xx = (yx[i] & 0xffff) + dx;
        const bool ok = (unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W;
        xreg[i] = buf_load4(rX, ok ? rowoff[i] + add : BUF_OOB);
      }
    } else if (vec) {
      int tap = 0, ci = k;
      if (a.taps != 1) { tap = k / a.Cin; ci = k - tap * a.Cin; }
      int dy, dx;
      tap_shift(tap, dy, dx);
      const unsigned add = (unsigned)((dy * a.W + dx) * a.ldx + ci) * 4u;
      const bool kok = k < a.K;
#pragma unroll
      for (int i = 0; i < XL; ++i) {
        const int yy = (yx[i] >> 16) + dy, xx = (yx[i] & 0xffff) + dx;
        const bool ok = kok && (unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W;
        xreg[i] = buf_load4(rX, ok ? rowoff[i] + add : BUF_OOB);
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int kk = k + j;
        int tap = 0, ci = kk;
        if (a.taps != 1) { tap = kk / a.Cin; ci = kk - tap * a.Cin; }
        int dy, dx;
        tap_shift(tap, dy, dx);
        const unsigned add = (unsigned)((dy * a.W + dx) * a.ldx + ci) * 4u;
        const bool kok = kk < a.K;
#pragma unroll
        for (int i = 0; i < XL; ++i) {
          const int yy = (yx[i] >> 16) + dy, xx = (yx[i] & 0xffff) + dx;
          const bool ok = kok && (unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W;
          xreg[i][j] = buf_load1(rX, ok ? rowoff[i] + add : BUF_OOB);
        }
      }
    }
  };
  auto lstore = [&](int buf) {
    float* Ws = smem + buf * (BN + BM) * LSK;
    float* Xs = Ws + BN * LSK;
#pragma unroll
    for (int i = 0; i < WL; ++i) {
      const int idx = tid + 256 * i, r = idx >> 3;
      if (r < BN) *reinterpret_cast<f32x4*>(Ws + lds_off<SWZ>(r, lq)) = wreg[i];
    }
#pragma unroll
    for (int i = 0; i < XL; ++i)
      *reinterpret_cast<f32x4*>(Xs + lds_off<SWZ>(lr + 32 * i, lq)) = xreg[i];
  };

  f32x4 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (c_begin < c_end) {
    gload(c_begin);
    lstore(0);
  }
  __syncthreads();
  for (int c = c_begin; c < c_end; ++c) {
    const int buf = (c - c_begin) & 1;
    if (c + 1 < c_end) gload(c + 1);
    const float* Ws = smem + buf * (BN + BM) * LSK;
    mfma_tile_step<TN, TM, SWZ>(Ws, Ws + BN * LSK, wn * 16 * TN, wm * 16 * TM, lane, acc);
    if (c + 1 < c_end) lstore(buf ^ 1);
    __syncthreads();
  }

  // ---------------- epilogue ----------------
  const int nq = (lane >> 4) * 4, mi = lane & 15;
  const bool ovec = ((a.ldo | a.N) & 3) == 0;
  f32x4 cdl[TN], cdb[TN];
#pragma unroll
  for (int t = 0; t < TN; ++t) { cdl[t] = f32x4{0.f, 0.f, 0.f, 0.f}; cdb[t] = cdl[t]; }
  float* outp = a.out + (long)z * a.slab_stride;
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int n = n0 + wn * 16 * TN + tn * 16 + nq;
    f32x4 e0 = {0.f, 0.f, 0.f, 0.f}, e1 = {1.f, 1.f, 1.f, 1.f};
    if (a.epi != EPI_RAW) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (n + r < a.N) {
          if (a.epi == EPI_AN_RELU) e0[r] = a.e0[n + r];
          e1[r] = a.e1[n + r];
        }
    }
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      const int m = m0 + wm * 16 * TM + tm * 16 + mi;
      if (m >= a.M || n >= a.N) continue;
      f32x4 v = acc[tn][tm];
      if (a.epi == EPI_AN_RELU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf((v[r] + e0[r]) * e1[r], 0.f);
      } else if (a.epi == EPI_DGRAD) {
        f32x4 h = {0.f, 0.f, 0.f, 0.f};
        const float* hp = a.hprev + (long)m * a.ldh + n;
        if (n + 3 < a.N && ((a.ldh & 3) == 0)) h = *reinterpret_cast<const f32x4*>(hp);
        else
#pragma unroll
          for (int r = 0; r < 4; ++r) if (n + r < a.N) h[r] = hp[r];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float g = v[r];
          cdl[tn][r] += g * h[r];
          v[r] = h[r] > 0.f ? g * e1[r] : 0.f;
          cdb[tn][r] += v[r];
        }
      }
      float* op = outp + (long)m * a.ldo + n;
      if (ovec && n + 3 < a.N) *reinterpret_cast<f32x4*>(op) = v;
      else
#pragma unroll
        for (int r = 0; r < 4; ++r) if (n + r < a.N) op[r] = v[r];
    }
  }
  if (a.epi == EPI_DGRAD) {
    // column partial sums: over the 16 pixel-lanes, then over the WM waves through LDS
    __syncthreads();
    float* red = smem;  // [WM][2][BN]
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float s0 = cdl[tn][r], s1 = cdb[tn][r];
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) {
          s0 += __shfl_xor(s0, off, 64);
          s1 += __shfl_xor(s1, off, 64);
        }
        if (mi == 0) {
          const int nl = wn * 16 * TN + tn * 16 + nq + r;
          red[(wm * 2 + 0) * BN + nl] = s0;
          red[(wm * 2 + 1) * BN + nl] = s1;
        }
      }
    __syncthreads();
    const int mb = bid / nNB;
    for (int t = tid; t < 2 * BN; t += 256) {
      const int which = t / BN, nl = t % BN;
      if (n0 + nl < a.N) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < WM; ++w) s += red[(w * 2 + which) * BN + nl];
        a.colsum[((long)mb * 2 + which) * a.N + n0 + nl] = s;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// wgrad kernel: out[slab][i][j] = sum_{m in slab} P[m][i] * Qim2col[m][j],  j = tap*Cq + cq
// ------------------------------------------------------------------------------------------

template <int TN, int TM, int WN, int WM>
__global__ void __launch_bounds__(256) wgrad_kernel(WgradArgs a) {
  constexpr int BI = 16 * TN * WN, BJ = 16 * TM * WM;
  constexpr int PL = (BI * 8 + 255) / 256, QL = (BJ * 8 + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wn = wid % WN, wm = wid / WN;
  const int nIB = (a.NI + BI - 1) / BI, nJB = (a.J + BJ - 1) / BJ;
  const int bid = xcd_remap(blockIdx.x, nIB * nJB);
  const int i0 = (bid % nIB) * BI, j0 = (bid / nIB) * BJ;
  const int HW = a.H * a.W;
  const int S = gridDim.y, z = blockIdx.y;
  const int nchunks = (a.M + BK - 1) / BK;
  const int c_begin = (int)((long)nchunks * z / S), c_end = (int)((long)nchunks * (z + 1) / S);
  const bool pvec = ((a.NI | a.ldp) & 3) == 0, qvec = ((a.Cq | a.ldq) & 3) == 0;

  // thread -> (pixel within chunk, channel group of 4): channel groups fastest => coalesced rows.
  // Sources are read with buffer loads (bounds check = zero fill); bf16 sources are widened to f32 here.
  constexpr int PG = BI / 4, QG = BJ / 4;
  const unsigned pes = a.pbf16 ? 2u : 4u, qes = a.qbf16 ? 2u : 4u;
  const __amdgpu_buffer_rsrc_t rP = make_rsrc(a.P, (unsigned)((long)a.M * a.ldp * pes));
  const __amdgpu_buffer_rsrc_t rQ = make_rsrc(a.Q, (unsigned)((long)a.M * a.ldq * qes));
  auto load4 = [&](__amdgpu_buffer_rsrc_t r, unsigned off, bool isbf) {
    if (isbf) {
      const u32x2 w = __builtin_amdgcn_raw_buffer_load_b64(r, (int)off, 0, 0);
      return f32x4{__builtin_bit_cast(float, w[0] << 16), __builtin_bit_cast(float, w[0] & 0xffff0000u),
                   __builtin_bit_cast(float, w[1] << 16), __builtin_bit_cast(float, w[1] & 0xffff0000u)};
    }
    return buf_load4(r, off);
  };
  auto load1 = [&](__amdgpu_buffer_rsrc_t r, unsigned off, bool isbf) {
    if (isbf) {
      const unsigned h = (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(r, (int)off, 0, 0);
      return __builtin_bit_cast(float, h << 16);
    }
    return buf_load1(r, off);
  };
  // per-thread constants: P column group, Q (tap, channel) of each slot
  int p_pl[PL], q_pl[QL], q_dyx[QL][4];
  unsigned p_col[PL], q_col[QL][4];
#pragma unroll
  for (int t = 0; t < PL; ++t) {
    const int idx = tid + 256 * t, g = idx % PG, i = i0 + g * 4;
    p_pl[t] = idx / PG;
    p_col[t] = (idx < BI * 8 && i < a.NI) ? (unsigned)i * pes : BUF_OOB;
  }
#pragma unroll
  for (int t = 0; t < QL; ++t) {
    const int idx = tid + 256 * t, g = idx % QG, j = j0 + g * 4;
    q_pl[t] = idx / QG;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int jj = j + r;
      int tap = 0, cq = jj;
      if (a.taps != 1) { tap = jj / a.Cq; cq = jj - tap * a.Cq; }
      const int dy = (a.taps == 1) ? 0 : tap / 3 - 1, dx = (a.taps == 1) ? 0 : tap % 3 - 1;
      q_dyx[t][r] = (dy << 8) | (dx & 0xff);
      q_col[t][r] = (idx < BJ * 8 && jj < a.J) ? (unsigned)((dy * a.W + dx) * a.ldq + cq) * qes : BUF_OOB;
    }
  }
  f32x4 preg[PL], qreg[QL];
  auto gload = [&](int chunk) {
    const int mb = chunk * BK;
#pragma unroll
    for (int t = 0; t < PL; ++t) {
      const int m = mb + p_pl[t];
      const unsigned base = (m < a.M && p_col[t] != BUF_OOB) ? (unsigned)m * (unsigned)a.ldp * pes + p_col[t] : BUF_OOB;
      if (pvec) preg[t] = load4(rP, base, a.pbf16);
      else {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r)
          v[r] = load1(rP, (base != BUF_OOB && (int)(p_col[t] / pes) + r < a.NI) ? base + r * pes : BUF_OOB, a.pbf16);
        preg[t] = v;
      }
    }
#pragma unroll
    for (int t = 0; t < QL; ++t) {
      const int m = mb + q_pl[t];
      const int rem = m % HW, y = rem / a.W, x = rem - y * a.W;
      const unsigned rowb = (unsigned)m * (unsigned)a.ldq * qes;
      auto okr = [&](int r) {
        const int dy = q_dyx[t][r] >> 8, dx = (int)(signed char)(q_dyx[t][r] & 0xff);
        return m < a.M && q_col[t][r] != BUF_OOB && (unsigned)(y + dy) < (unsigned)a.H &&
               (unsigned)(x + dx) < (unsigned)a.W;
      };
      if (qvec) qreg[t] = load4(rQ, okr(0) ? rowb + q_col[t][0] : BUF_OOB, a.qbf16);
      else {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = load1(rQ, okr(r) ? rowb + q_col[t][r] : BUF_OOB, a.qbf16);
        qreg[t] = v;
      }
    }
  };
  constexpr int LDP = BI + 4, LDQ = BJ + 4, STAGE = BK * (LDP + LDQ);
  auto lstore = [&](int buf) {
    float* Ps = smem + buf * STAGE;
    float* Qs = Ps + BK * LDP;
#pragma unroll
    for (int t = 0; t < PL; ++t) {
      const int idx = tid + 256 * t;
      const int pl = idx / PG, g = idx % PG;
      if (idx < BI * 8) *reinterpret_cast<f32x4*>(Ps + pl * LDP + g * 4) = preg[t];
    }
#pragma unroll
    for (int t = 0; t < QL; ++t) {
      const int idx = tid + 256 * t;
      const int pl = idx / QG, g = idx % QG;
      if (idx < BJ * 8) *reinterpret_cast<f32x4*>(Qs + pl * LDQ + g * 4) = qreg[t];
    }
  };

  f32x4 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (c_begin < c_end) {
    gload(c_begin);
    lstore(0);
  }
  __syncthreads();
  for (int c = c_begin; c < c_end; ++c) {
    const int buf = (c - c_begin) & 1;
    if (c + 1 < c_end) gload(c + 1);
    const float* Ps = smem + buf * STAGE;
    mfma_tile_step_kmajor<TN, TM>(Ps, LDP, Ps + BK * LDP, LDQ, wn * 16 * TN, wm * 16 * TM, lane, acc);
    if (c + 1 < c_end) lstore(buf ^ 1);
    __syncthreads();
  }
  const int iq = (lane >> 4) * 4, jl = lane & 15;
  float* outp = a.out + (long)z * a.slab_stride;
#pragma unroll
  for (int tn = 0; tn < TN; ++tn)
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      const int i = i0 + wn * 16 * TN + tn * 16 + iq, j = j0 + wm * 16 * TM + tm * 16 + jl;
      if (j >= a.J) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (i + r < a.NI) outp[(long)(i + r) * a.J + j] = acc[tn][tm][r];
    }
}

// ------------------------------------------------------------------------------------------
// weight repack:  W[co][ci][taps]  ->  fwd  Wt[co][tap*Cin + ci]      (zero padded to Kp)
//                                 ->  dgrad Wd[ci][tap'*Cout + co] = W[co][ci][taps-1-tap']
// ------------------------------------------------------------------------------------------
__global__ void repack_kernel(const float* __restrict__ W, int Cout, int Cin, int taps, float* __restrict__ Wt,
                              int Kp, float* __restrict__ Wd, int Kpd) {
  const long total_f = (long)Cout * Kp, total_d = Wd ? (long)Cin * Kpd : 0;
  for (long t = blockIdx.x * (long)blockDim.x + threadIdx.x; t < total_f + total_d;
       t += (long)gridDim.x * blockDim.x) {
    if (t < total_f) {
      const int co = t / Kp, k = t % Kp;
      float v = 0.f;
      if (k < taps * Cin) {
        const int tap = k / Cin, ci = k % Cin;
        v = W[((long)co * Cin + ci) * taps + tap];
      }
      Wt[t] = v;
    } else {
      const long u = t - total_f;
      const int ci = u / Kpd, k = u % Kpd;
      float v = 0.f;
      if (k < taps * Cout) {
        const int tap = k / Cout, co = k % Cout;
        v = W[((long)co * Cin + ci) * taps + (taps - 1 - tap)];
      }
      Wd[u] = v;
    }
  }
}

__global__ void __launch_bounds__(256) repack_batch_kernel(const nfk_repack_desc* __restrict__ descs) {
  repack_f32_body(descs[blockIdx.y], blockIdx.x, gridDim.x, threadIdx.x, 256);
}

// wgrad finalize: gradW[co][ci][tap] += sum_slabs out[s][co][tap*Cin + ci]   (fixed order)
__global__ void wgrad_finalize_kernel(const float* __restrict__ slabs, int S, long slab_stride, int Cout,
                                      int Cin, int taps, float* __restrict__ gradW) {
  const long total = (long)Cout * Cin * taps;
  const int J = Cin * taps;
  for (long t = blockIdx.x * (long)blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const int tap = t % taps;
    const long r = t / taps;
    const int ci = r % Cin, co = r / Cin;
    const long src = (long)co * J + tap * Cin + ci;
    gradW[t] += slab_sum(slabs, S, slab_stride, src);
  }
}

// ------------------------------------------------------------------------------------------
// optional per-kernel-class timing with HIP events (bench.py's roofline leg); off by default
// ------------------------------------------------------------------------------------------
struct ProfClass {
  std::vector<hipEvent_t> ev;   // start/stop pairs
  size_t used = 0;
  double flops = 0.0, bytes = 0.0;
};
static bool g_prof_on = false;
static ProfClass g_prof[NFK_PROF_CLASSES];

struct ProfScope {
  ProfClass* c = nullptr;
  hipStream_t st;
  ProfScope(int cls, double flops, double bytes, hipStream_t s) : st(s) {
    if (!g_prof_on) return;
    c = &g_prof[cls];
    if (c->used + 2 > c->ev.size()) {
      c->ev.resize(c->used + 2);
      (void)hipEventCreate(&c->ev[c->used]);
      (void)hipEventCreate(&c->ev[c->used + 1]);
    }
    c->flops += flops;
    c->bytes += bytes;
    (void)hipEventRecord(c->ev[c->used], st);
  }
  ~ProfScope() {
    if (!c) return;
    (void)hipEventRecord(c->ev[c->used + 1], st);
    c->used += 2;
  }
};

void prof_scope_begin(int cls, double flops, double bytes, hipStream_t st) {
  if (!g_prof_on) return;
  ProfClass* c = &g_prof[cls];
  if (c->used + 2 > c->ev.size()) {
    c->ev.resize(c->used + 2);
    (void)hipEventCreate(&c->ev[c->used]);
    (void)hipEventCreate(&c->ev[c->used + 1]);
  }
  c->flops += flops;
  c->bytes += bytes;
  (void)hipEventRecord(c->ev[c->used], st);
}
void prof_scope_end(int cls, hipStream_t st) {
  if (!g_prof_on) return;
  ProfClass* c = &g_prof[cls];
  (void)hipEventRecord(c->ev[c->used + 1], st);
  c->used += 2;
}

void prof_begin() {
  for (auto& c : g_prof) { c.used = 0; c.flops = 0.0; c.bytes = 0.0; }
  g_prof_on = true;
}
// ms/launches/flops/bytes: arrays of NFK_PROF_CLASSES
void prof_end(double* ms, long long* launches, double* flops, double* bytes) {
  g_prof_on = false;
  (void)hipDeviceSynchronize();
  for (int i = 0; i < NFK_PROF_CLASSES; ++i) {
    double t = 0.0;
    for (size_t k = 0; k + 1 < g_prof[i].used; k += 2) {
      float e = 0.f;
      (void)hipEventElapsedTime(&e, g_prof[i].ev[k], g_prof[i].ev[k + 1]);
      t += e;
    }
    ms[i] = t;
    launches[i] = (long long)(g_prof[i].used / 2);
    flops[i] = g_prof[i].flops;
    bytes[i] = g_prof[i].bytes;
  }
}

// ------------------------------------------------------------------------------------------
// host-side launchers
// ------------------------------------------------------------------------------------------
template <int TN, int TM, int WN, int WM, bool SWZ = true>
static int launch_conv(const ConvGemmArgs& a, int S, hipStream_t st, int cls) {
  constexpr int BN = 16 * TN * WN, BM = 16 * TM * WM;
  constexpr size_t lds = 2 * (BN + BM) * (SWZ ? BK : LS) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    if (lds > 48 * 1024)
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_gemm_kernel<TN, TM, WN, WM, SWZ>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  dim3 grid(ceil_div(a.N, BN) * ceil_div(a.M, BM), S);
  ProfScope ps(cls, 2.0 * a.M * (double)a.N * a.K,
               4.0 * ((double)a.M * a.Cin + (double)a.N * a.K + (double)S * a.M * a.N), st);
  hipLaunchKernelGGL((conv_gemm_kernel<TN, TM, WN, WM, SWZ>), grid, dim3(256), lds, st, a);
  NFK_LAUNCH_CHECK();
  return NFK_OK;
}

template <int TN, int TM, int WN, int WM>
static int launch_wgrad(const WgradArgs& a, int S, hipStream_t st, int cls) {
  constexpr int BI = 16 * TN * WN, BJ = 16 * TM * WM;
  constexpr size_t lds = 2 * BK * (BI + BJ + 8) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    if (lds > 48 * 1024)
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_kernel<TN, TM, WN, WM>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  dim3 grid(ceil_div(a.NI, BI) * ceil_div(a.J, BJ), S);
  ProfScope ps(cls, 2.0 * a.M * (double)a.NI * a.J,
               4.0 * ((double)a.M * a.NI + (double)a.M * a.Cq + (double)S * a.NI * a.J), st);
  hipLaunchKernelGGL((wgrad_kernel<TN, TM, WN, WM>), grid, dim3(256), lds, st, a);
  NFK_LAUNCH_CHECK();
  return NFK_OK;
}

int padded_k(int K) { return ceil_div(K, BK) * BK; }

// How many split-K slabs conv_gemm() will write for this problem (RAW epilogue only).
int conv_splitk(int M, int N, int K, int epi) {
  if (epi != EPI_RAW || N > 96) return 1;
  const int nMB = ceil_div(M, 128), nk = ceil_div(K, BK);
  int S = ceil_div(512, nMB);
  if (S > nk / 4) S = nk / 4;
  if (S < 1) S = 1;
  if (S > 16) S = 16;
  return S;
}

int conv_mblocks(int M, int N) {
  if (N > 96) {
    const bool big = (long)ceil_div(N, 128) * ceil_div(M, 128) >= 256;
    return ceil_div(M, big ? 128 : 64);
  }
  return ceil_div(M, 128);
}

static int g_tune = 0;
void set_tuning(int v) { g_tune = v; }

int conv_gemm(ConvGemmArgs a, hipStream_t st) {
  if (a.taps != 1 && a.taps != 9) return NFK_ERR_ARG;
  a.K = a.taps * a.Cin;
  a.M = a.B * a.H * a.W;
  a.nkchunks = ceil_div(a.K, BK);
  if (a.Kp < a.nkchunks * BK) return NFK_ERR_ARG;
  if (a.M <= 0 || a.N <= 0) return NFK_OK;
  if ((long)a.M * a.ldx * 4 >= 0x7fffffffL || (long)a.N * a.Kp * 4 >= 0x7fffffffL) return NFK_ERR_ARG;
  const int S = conv_splitk(a.M, a.N, a.K, a.epi);
  if (a.N > 96) {
    const bool big = (long)ceil_div(a.N, 128) * ceil_div(a.M, 128) >= 256;
    if (g_tune == 1) return launch_conv<4, 4, 2, 2, false>(a, S, st, NFK_PROF_CONV_WIDE128);
    if (g_tune == 2) return launch_conv<4, 2, 2, 2>(a, S, st, NFK_PROF_CONV_WIDE128);
    if (g_tune == 3) return launch_conv<2, 4, 2, 2>(a, S, st, NFK_PROF_CONV_WIDE128);
    if (g_tune == 4) return launch_conv<2, 2, 2, 2>(a, S, st, NFK_PROF_CONV_WIDE64);
    if (g_tune == 5) return launch_conv<4, 4, 2, 2>(a, S, st, NFK_PROF_CONV_WIDE128);
    return big ? launch_conv<4, 4, 2, 2>(a, S, st, NFK_PROF_CONV_WIDE128) : launch_conv<2, 2, 2, 2>(a, S, st, NFK_PROF_CONV_WIDE64);
  }
  switch (ceil_div(a.N, 16)) {
    case 1: return launch_conv<1, 2, 1, 4>(a, S, st, NFK_PROF_CONV_NARROW);
    case 2: return launch_conv<2, 2, 1, 4>(a, S, st, NFK_PROF_CONV_NARROW);
    case 3: return launch_conv<3, 2, 1, 4>(a, S, st, NFK_PROF_CONV_NARROW);
    case 4: return launch_conv<4, 2, 1, 4>(a, S, st, NFK_PROF_CONV_NARROW);
    default: return launch_conv<6, 2, 1, 4>(a, S, st, NFK_PROF_CONV_NARROW);
  }
}

int wgrad_splits(int M, int NI, int J) {
  const int tiles = (NI > 96 ? ceil_div(NI, 128) : 1) * ceil_div(J, 128);
  const int nchunks = ceil_div(M, BK);
  int S = ceil_div(768, tiles);
  if (S > nchunks / 2) S = nchunks / 2;
  if (S < 1) S = 1;
  if (S > 64) S = 64;
  return S;
}

int wgrad_gemm(WgradArgs a, int S, hipStream_t st) {
  a.M = a.B * a.H * a.W;
  a.J = a.taps * a.Cq;
  if (a.M <= 0 || a.NI <= 0 || a.J <= 0) return NFK_OK;
  if ((long)a.M * a.ldp * 4 >= 0x7fffffffL || (long)a.M * a.ldq * 4 >= 0x7fffffffL) return NFK_ERR_ARG;
  if (a.NI > 96) return launch_wgrad<4, 4, 2, 2>(a, S, st, NFK_PROF_WGRAD_WIDE);
  switch (ceil_div(a.NI, 16)) {
    case 1: return launch_wgrad<1, 2, 1, 4>(a, S, st, NFK_PROF_WGRAD_NARROW);
    case 2: return launch_wgrad<2, 2, 1, 4>(a, S, st, NFK_PROF_WGRAD_NARROW);
    case 3: return launch_wgrad<3, 2, 1, 4>(a, S, st, NFK_PROF_WGRAD_NARROW);
    case 4: return launch_wgrad<4, 2, 1, 4>(a, S, st, NFK_PROF_WGRAD_NARROW);
    default: return launch_wgrad<6, 2, 1, 4>(a, S, st, NFK_PROF_WGRAD_NARROW);
  }
}

int repack_weights(const float* W, int Cout, int Cin, int taps, float* Wt, float* Wd, hipStream_t st) {
  const int Kp = padded_k(taps * Cin), Kpd = padded_k(taps * Cout);
  const long total = (long)Cout * Kp + (Wd ? (long)Cin * Kpd : 0);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(repack_kernel, dim3(blocks), dim3(256), 0, st, W, Cout, Cin, taps, Wt, Kp, Wd, Kpd);
  NFK_LAUNCH_CHECK();
  return NFK_OK;
}

int repack_batch(const nfk_repack_desc* descs, int n, hipStream_t st) {
  if (n <= 0) return NFK_OK;
  hipLaunchKernelGGL(repack_batch_kernel, dim3(64, n), dim3(256), 0, st, descs);
  NFK_LAUNCH_CHECK();
  return NFK_OK;
}

int wgrad_finalize(const float* slabs, int S, long slab_stride, int Cout, int Cin, int taps, float* gradW,
                   hipStream_t st) {
  const long total = (long)Cout * Cin * taps;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(wgrad_finalize_kernel, dim3(blocks), dim3(256), 0, st, slabs, S, slab_stride, Cout, Cin,
                     taps, gradW);
  NFK_LAUNCH_CHECK();
  return NFK_OK;
}

}  // namespace nfk
