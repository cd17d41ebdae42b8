// nfk GEMM engine, bf16-operand / fp32-accumulate variant (throughput mode, BASELINE.json config 2):
// the same implicit-GEMM convolution as nfk_gemm.hip on v_mfma_f32_16x16x32_bf16.
//
//   * the 512-wide hidden tensors (h1, h2 and their gradients) live in HBM as bf16: half the bytes
//   * weights are repacked to bf16 [N][Kp64] once per forward (fp32 master weights stay with Adam)
//   * k-chunk = 64 bf16 = 128-byte LDS rows, the same XOR-swizzled geometry as the f32 kernel:
//     one ds_read_b128 yields a lane's 8 consecutive k of one MFMA operand, conflict-free
//   * accumulation, ActNorm scale/bias, ReLU, masks and per-channel sums stay fp32; narrow inputs
//     (z1, g_a3: C <= 96 channels) stay fp32 in HBM and are converted while staging (XF32)
#include "nfk_internal.h"
#include <type_traits>
#include "nfk_repack.h"
#include "nfk_tail.h"

namespace nfk {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4b __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2b __attribute__((ext_vector_type(2)));

constexpr int BKH = 64;                 // bf16 elements per k-chunk
static int g_narrow_halo = 1;            // conv1 dgrad through the halo-tile kernel (nfk_tune NFK_TUNE_NARROW_HALO_DGRAD)
// Split caps of the weight-gradient GEMMs.  With the fused backward all three GEMMs of a step share one launch, every
// workgroup is a short latency-bound pipeline (load a 64-pixel chunk, 32 MFMAs, next chunk) and every split leaves a slab
// the step tail reads back: 16 splits (16 chunks per workgroup at CIFAR level 1, 16 MiB of conv2 slabs instead of 32)
// measured +2.7 % step throughput over 32 / 64.
static int g_wgrad_wide_max_splits = 16;   // 512x512: 1 MiB per split
static int g_wgrad_target_wgs = 768, g_wgrad_max_splits = 16;   // nfk_tune NFK_TUNE_WGRAD_MAX_SPLITS / _TARGET_WGS
static int g_big_tile_min_k = 256;      // see big_tiles(); nfk_tune NFK_TUNE_BF16_BIG_TILE_MIN_K64 sets it to v*64 (experiments)
constexpr unsigned OOB = 0x80000000u;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ u32x4b ld16(__amdgpu_buffer_rsrc_t r, unsigned off) {
  return __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0);
}
__device__ __forceinline__ float ld4f(__amdgpu_buffer_rsrc_t r, unsigned off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, 0));
}
__device__ __forceinline__ unsigned pack2(float lo, float hi) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  bf16x2 v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ int xcd_remap_b(int id, int n) {
  const int q = n >> 3, r = n & 7, x = id & 7, s = id >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + s;
}
// byte offset of 16-byte slot `slot` (0..7) of tile row `row` (128-byte rows, XOR swizzle)
__device__ __forceinline__ int lds_boff(int row, int slot) { return row * 128 + ((slot ^ ((row >> 1) & 7)) << 4); }

// The kernel bodies are device functions of (arguments, tile index, split count, split index) so that one launch can
// run a dgrad GEMM and a wgrad GEMM side by side (dual_dgrad_wgrad_bf16_kernel below).
template <int TN, int TM, int WN, int WM, bool XF32>
__device__ __forceinline__ void conv_gemm_bf16_body(const ConvGemmArgs& a, const int bx, const int S, const int z) {
  static_assert(WN * WM == 4, "256-thread workgroups");
  constexpr int BN = 16 * TN * WN, BM = 16 * TM * WM;
  constexpr int WL = (BN * 8 + 255) / 256, XL = BM / 32;
  constexpr int STAGE = (BN + BM) * 128;  // bytes
  extern __shared__ __attribute__((aligned(16))) char smem_b[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wn = wid % WN, wm = wid / WN;
  const int nNB = (a.N + BN - 1) / BN, nMB = (a.M + BM - 1) / BM;
  const int bid = xcd_remap_b(bx, nNB * nMB);
  const int n0 = (bid % nNB) * BN, m0 = (bid / nNB) * BM;
  const int HW = a.H * a.W;
  constexpr int XE = XF32 ? 4 : 2;  // bytes per X element
  const int c_begin = (int)((long)a.nkchunks * z / S), c_end = (int)((long)a.nkchunks * (z + 1) / S);

  const __amdgpu_buffer_rsrc_t rX = rsrc_of(a.X, (unsigned)((long)a.M * a.ldx * XE));
  const __amdgpu_buffer_rsrc_t rW = rsrc_of(a.Wt, (unsigned)((long)a.N * a.Kp * 2));
  const int lq = tid & 7, lr = tid >> 3;
  unsigned rowoff[XL], woff[WL];
  int yx[XL];
#pragma unroll
  for (int i = 0; i < XL; ++i) {
    const int m = m0 + lr + 32 * i;
    const int rem = m % HW;
    rowoff[i] = m < a.M ? (unsigned)m * (unsigned)a.ldx * XE : OOB;
    yx[i] = ((rem / a.W) << 16) | (rem % a.W);
  }
#pragma unroll
  for (int i = 0; i < WL; ++i) {
    const int r = (tid + 256 * i) >> 3, n = n0 + r;
    woff[i] = (r < BN && n < a.N) ? (unsigned)n * (unsigned)a.Kp * 2u : OOB;
  }
  // granularity at which 8 consecutive k stay inside one tap and are vector-loadable
  const bool v8 = ((a.Cin | a.ldx) & 7) == 0;   // bf16: one 16-B load; f32: two 16-B loads, one decode
  const bool v4 = ((a.Cin | a.ldx) & 3) == 0;   // f32 only: two 16-B loads, decoded separately
  const bool unif = v8 && (a.Cin % BKH) == 0;

  u32x4b wreg[WL], xreg[XL];
  auto decode = [&](int k, unsigned& add, bool& kok) {
    int tap = 0, ci = k;
    if (a.taps != 1) { tap = k / a.Cin; ci = k - tap * a.Cin; }
    const int dy = (a.taps == 1) ? 0 : tap / 3 - 1, dx = (a.taps == 1) ? 0 : tap % 3 - 1;
    add = (unsigned)((dy * a.W + dx) * a.ldx + ci) * XE;
    kok = k < a.K;
    return (dy << 8) | (dx & 0xff);
  };
  auto inb = [&](int i, int dyx) {
    const int dy = dyx >> 8, dx = (int)(signed char)(dyx & 0xff);
    const int yy = (yx[i] >> 16) + dy, xx = (yx[i] & 0xffff) + dx;
    return (unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W;
  };
  auto gload = [&](int chunk) {
    const int k = chunk * BKH + lq * 8;
#pragma unroll
    for (int i = 0; i < WL; ++i) wreg[i] = ld16(rW, woff[i] + (unsigned)k * 2u);
    if (!XF32) {
      if (v8) {
        unsigned add; bool kok;
        const int dyx = decode(unif ? chunk * BKH : k, add, kok);
        if (unif) add += (unsigned)lq * 8u * XE;
#pragma unroll
        for (int i = 0; i < XL; ++i) xreg[i] = ld16(rX, (kok && inb(i, dyx)) ? rowoff[i] + add : OOB);
      } else {
#pragma unroll
        for (int i = 0; i < XL; ++i) xreg[i] = u32x4b{0u, 0u, 0u, 0u};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          unsigned add; bool kok;
          const int dyx = decode(k + j, add, kok);
#pragma unroll
          for (int i = 0; i < XL; ++i) {
            const unsigned short h = (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(
                rX, (int)((kok && inb(i, dyx)) ? rowoff[i] + add : OOB), 0, 0);
            xreg[i][j >> 1] |= (unsigned)h << ((j & 1) * 16);
          }
        }
      }
    } else {
      if (v4) {
#pragma unroll
        for (int hlf = 0; hlf < 2; ++hlf) {
          unsigned add; bool kok;
          const int dyx = decode(k + 4 * hlf, add, kok);
#pragma unroll
          for (int i = 0; i < XL; ++i) {
            const f32x4 f = __builtin_bit_cast(f32x4, ld16(rX, (kok && inb(i, dyx)) ? rowoff[i] + add : OOB));
            xreg[i][2 * hlf] = pack2(f[0], f[1]);
            xreg[i][2 * hlf + 1] = pack2(f[2], f[3]);
          }
        }
      } else {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          unsigned add0, add1; bool k0, k1;
          const int d0 = decode(k + 2 * jj, add0, k0), d1 = decode(k + 2 * jj + 1, add1, k1);
#pragma unroll
          for (int i = 0; i < XL; ++i) {
            const float f0 = ld4f(rX, (k0 && inb(i, d0)) ? rowoff[i] + add0 : OOB);
            const float f1 = ld4f(rX, (k1 && inb(i, d1)) ? rowoff[i] + add1 : OOB);
            xreg[i][jj] = pack2(f0, f1);
          }
        }
      }
    }
  };
  auto lstore = [&](int buf) {
    char* Ws = smem_b + buf * STAGE;
    char* Xs = Ws + BN * 128;
#pragma unroll
    for (int i = 0; i < WL; ++i) {
      const int r = (tid + 256 * i) >> 3;
      if (r < BN) *reinterpret_cast<u32x4b*>(Ws + lds_boff(r, lq)) = wreg[i];
    }
#pragma unroll
    for (int i = 0; i < XL; ++i) *reinterpret_cast<u32x4b*>(Xs + lds_boff(lr + 32 * i, lq)) = xreg[i];
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
  const int r16 = lane & 15, q4 = lane >> 4;
  // dgrad epilogue operand (the layer's saved post-ReLU activation at this lane's outputs): requested now, so that its
  // round trip runs under the k loop instead of after it (short-K dgrads are little more than their epilogue)
  bf16x4 hpre[TN][TM];
  const bool hvec = a.epi == EPI_DGRAD && (a.ldh & 3) == 0 && (a.N & 3) == 0;
  if (hvec) {
    const __amdgpu_buffer_rsrc_t rH = rsrc_of(a.hprev, (unsigned)((long)a.M * a.ldh * 2));
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
        const int n = n0 + wn * 16 * TN + tn * 16 + q4 * 4, m = m0 + wm * 16 * TM + tm * 16 + r16;
        const u32x2b v = __builtin_amdgcn_raw_buffer_load_b64(rH, (m < a.M && n + 3 < a.N) ? (int)((m * a.ldh + n) * 2) : (int)OOB, 0, 0);
        hpre[tn][tm] = __builtin_bit_cast(bf16x4, v);
      }
  }
  // per-channel epilogue constants: also requested before the k loop
  f32x4 e0v[TN], e1v[TN];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int n = n0 + wn * 16 * TN + tn * 16 + q4 * 4;
    e0v[tn] = f32x4{0.f, 0.f, 0.f, 0.f};
    e1v[tn] = f32x4{1.f, 1.f, 1.f, 1.f};
    if (a.epi != EPI_RAW) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (n + r < a.N) {
          if (a.epi == EPI_AN_RELU) e0v[tn][r] = a.e0[n + r];
          e1v[tn][r] = a.e1[n + r];
        }
    }
  }
  __syncthreads();
  for (int c = c_begin; c < c_end; ++c) {
    const int buf = (c - c_begin) & 1;
    if (c + 1 < c_end) gload(c + 1);
    const char* Ws = smem_b + buf * STAGE;
    const char* Xs = Ws + BN * 128;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 fa[TN], fb[TM];
#pragma unroll
      for (int t = 0; t < TN; ++t)
        fa[t] = *reinterpret_cast<const bf16x8*>(Ws + lds_boff(wn * 16 * TN + t * 16 + r16, ks * 4 + q4));
#pragma unroll
      for (int t = 0; t < TM; ++t)
        fb[t] = *reinterpret_cast<const bf16x8*>(Xs + lds_boff(wm * 16 * TM + t * 16 + r16, ks * 4 + q4));
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
          acc[tn][tm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[tn], fb[tm], acc[tn][tm], 0, 0, 0);
    }
    if (c + 1 < c_end) lstore(buf ^ 1);
    __syncthreads();
  }

  // ---------------- epilogue (fp32 math; bf16 or f32 stores) ----------------
  const int nq = q4 * 4, mi = r16;
  const bool n4 = (a.N & 3) == 0;
  f32x4 cdl[TN], cdb[TN];
#pragma unroll
  for (int t = 0; t < TN; ++t) { cdl[t] = f32x4{0.f, 0.f, 0.f, 0.f}; cdb[t] = cdl[t]; }
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int n = n0 + wn * 16 * TN + tn * 16 + nq;
    const f32x4 e0 = e0v[tn], e1 = e1v[tn];
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
        const __bf16* hp = reinterpret_cast<const __bf16*>(a.hprev) + (long)m * a.ldh + n;
        if (hvec) {
#pragma unroll
          for (int r = 0; r < 4; ++r) h[r] = (float)hpre[tn][tm][r];
        } else if (n + 3 < a.N && ((a.ldh & 3) == 0)) {
          const bf16x4 hb = *reinterpret_cast<const bf16x4*>(hp);
#pragma unroll
          for (int r = 0; r < 4; ++r) h[r] = (float)hb[r];
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) if (n + r < a.N) h[r] = (float)hp[r];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float g = v[r];
          cdl[tn][r] += g * h[r];
          v[r] = h[r] > 0.f ? g * e1[r] : 0.f;
          cdb[tn][r] += v[r];
        }
      }
      if (a.obf16) {
        __bf16* op = reinterpret_cast<__bf16*>(a.out) + (long)m * a.ldo + n;
        if (n4 && ((a.ldo & 3) == 0) && n + 3 < a.N) {
          u32x2b pk = {pack2(v[0], v[1]), pack2(v[2], v[3])};
          *reinterpret_cast<u32x2b*>(op) = pk;
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) if (n + r < a.N) op[r] = (__bf16)v[r];
        }
      } else {
        float* op = a.out + (long)z * a.slab_stride + (long)m * a.ldo + n;
        if (n4 && ((a.ldo & 3) == 0) && n + 3 < a.N) *reinterpret_cast<f32x4*>(op) = v;
        else
#pragma unroll
          for (int r = 0; r < 4; ++r) if (n + r < a.N) op[r] = v[r];
      }
    }
  }
  if (a.epi == EPI_DGRAD) {
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem_b);  // [WM][2][BN]
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

template <int TN, int TM, int WN, int WM, bool XF32>
__global__ void __launch_bounds__(256) conv_gemm_bf16_kernel(ConvGemmArgs a) {
  conv_gemm_bf16_body<TN, TM, WN, WM, XF32>(a, blockIdx.x, gridDim.y, blockIdx.y);
}

__global__ void __launch_bounds__(256) repack_batch_bf16_kernel(const nfk_repack_desc* __restrict__ descs) {
  __shared__ float T[RP_LDS_FLOATS];
  repack_bf16_body(descs[blockIdx.y], blockIdx.x, gridDim.x, T, threadIdx.x, 256);
}

template <int TN, int TM, int WN, int WM>
static int launch_bf16(const ConvGemmArgs& a, int S, hipStream_t st, int cls) {
  constexpr int BN = 16 * TN * WN, BM = 16 * TM * WM;
  constexpr size_t lds = 2 * (BN + BM) * 128;
  dim3 grid(ceil_div(a.N, BN) * ceil_div(a.M, BM), S);
  static bool attr0 = false, attr1 = false;
  prof_scope_begin(cls, 2.0 * a.M * (double)a.N * a.K,
                   (a.xf32 ? 4.0 : 2.0) * a.M * a.Cin + 2.0 * a.N * a.K + (a.obf16 ? 2.0 : 4.0 * S) * a.M * a.N, st);
  if (a.xf32) {
    if (!attr1 && lds > 48 * 1024) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_gemm_bf16_kernel<TN, TM, WN, WM, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      attr1 = true;
    }
    hipLaunchKernelGGL((conv_gemm_bf16_kernel<TN, TM, WN, WM, true>), grid, dim3(256), lds, st, a);
  } else {
    if (!attr0 && lds > 48 * 1024) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_gemm_bf16_kernel<TN, TM, WN, WM, false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      attr0 = true;
    }
    hipLaunchKernelGGL((conv_gemm_bf16_kernel<TN, TM, WN, WM, false>), grid, dim3(256), lds, st, a);
  }
  prof_scope_end(cls, st);
  NFK_LAUNCH_CHECK();
  return NFK_OK;
}

int padded_k_bf16(int K) { return ceil_div(K, BKH) * BKH; }

int conv_splitk_bf16(int M, int N, int K, int epi) {
  if (epi != EPI_RAW || N > 96) return 1;
  const int nMB = ceil_div(M, 128), nk = ceil_div(K, BKH);
  int S = ceil_div(512, nMB);
  if (S > nk / 2) S = nk / 2;
  if (S < 1) S = 1;
  if (S > 16) S = 16;
  return S;
}
// 128x128 tiles only when they still fill the chip AND the k loop is long enough to amortise the tile's prologue and
// epilogue: a dgrad with K = 9*12 is two k-chunks of pure epilogue, where 64x64 tiles (4x the workgroups, 5 per CU)
// hide the latency better.
static bool big_tiles(int M, int N, int K) {
  return (long)ceil_div(N, 128) * ceil_div(M, 128) >= 256 && K > g_big_tile_min_k;
}
int conv_mblocks_bf16(int M, int N, int K) {
  return N > 96 ? ceil_div(M, big_tiles(M, N, K) ? 128 : 64) : ceil_div(M, 128);
}

int conv_gemm_bf16(ConvGemmArgs a, hipStream_t st) {
  if (a.taps != 1 && a.taps != 9) return NFK_ERR_ARG;
  a.K = a.taps * a.Cin;
  a.M = a.B * a.H * a.W;
  a.nkchunks = ceil_div(a.K, BKH);
  if (a.Kp < a.nkchunks * BKH) return NFK_ERR_ARG;
  if (a.M <= 0 || a.N <= 0) return NFK_OK;
  if (!a.xf32 && (((a.Cin | a.ldx) & 7) != 0) && false) return NFK_ERR_ARG;
  if ((long)a.M * a.ldx * 4 >= 0x7fffffffL || (long)a.N * a.Kp * 2 >= 0x7fffffffL) return NFK_ERR_ARG;
  const int S = conv_splitk_bf16(a.M, a.N, a.K, a.epi);
  if (a.N > 96)
    return big_tiles(a.M, a.N, a.K) ? launch_bf16<4, 4, 2, 2>(a, S, st, NFK_PROF_CONV_WIDE128)
                               : launch_bf16<2, 2, 2, 2>(a, S, st, NFK_PROF_CONV_WIDE64);
  switch (ceil_div(a.N, 16)) {
    case 1: return launch_bf16<1, 2, 1, 4>(a, S, st, NFK_PROF_CONV_NARROW);
    case 2: return launch_bf16<2, 2, 1, 4>(a, S, st, NFK_PROF_CONV_NARROW);
    case 3: return launch_bf16<3, 2, 1, 4>(a, S, st, NFK_PROF_CONV_NARROW);
    case 4: return launch_bf16<4, 2, 1, 4>(a, S, st, NFK_PROF_CONV_NARROW);
    default: return launch_bf16<6, 2, 1, 4>(a, S, st, NFK_PROF_CONV_NARROW);
  }
}

int repack_batch_bf16(const nfk_repack_desc* descs, int n, hipStream_t st) {
  if (n <= 0) return NFK_OK;
  hipLaunchKernelGGL(repack_batch_bf16_kernel, dim3(64, n), dim3(256), 0, st, descs);
  NFK_LAUNCH_CHECK();
  return NFK_OK;
}

// ------------------------------------------------------------------------------------------
// wgrad on the bf16 matrix cores: out[slab][i][j] = sum_m P[m][i] * Qim2col[m][j]
// Tiles are staged pixel-major [64 pixels][channels] exactly as they sit in HBM (16-byte copies);
// the MFMA operands need 8 consecutive PIXELS of one channel per lane, which the gfx950 transpose
// read ds_read_b64_tr_b16 delivers straight from that image (4 pixels x 16 channels per 16 lanes).
// Row stride = 32 B * odd so the 8 rows a half-wave touches fall on distinct banks.
// ------------------------------------------------------------------------------------------
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
constexpr int WPX = 64;  // pixels per chunk (two 32-deep MFMA k-steps)

__host__ __device__ constexpr int wg_stride(int ch) { return ((ch / 16) & 1) ? ch * 2 : ch * 2 + 32; }  // bytes

__device__ __forceinline__ bf16x8 tr_frag(const char* tile, int stride, int c0, int ks, int lane) {
  const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  const char* a0 = tile + (32 * ks + 4 * g + q) * stride + (c0 + 4 * p) * 2;
  typedef s16x4 __attribute__((address_space(3))) * lds_v4;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(a0));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(a0 + 16 * stride));
  const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}

// PLAIN: both operands are bf16, pixel-major, without taps and 16-byte loadable (the fused backward's three GEMMs): the
// per-load index arithmetic of the general case -- two integer divisions per Q piece for the tap geometry, element-wise
// fallbacks -- drops out at compile time (it sat in front of every chunk's loads: ~300 VALU instructions per chunk and thread
// beside 512 cycles of MFMA).
// PF: pixel chunks kept in flight in REGISTERS ahead of the one being stored to LDS (1: requested one iteration before it is
// needed; 3: three iterations ahead, 96 VGPRs of staging).  Round 3 measured 3 on the level launches (16-64 chunks per
// workgroup): results identical, time identical (CIFAR level 1: 245 us for 90 GFLOP either way) -- with two workgroups per
// CU the loads were hidden already; what bounds the body is the LDS: per 64-pixel chunk a wave reads 16 KB of transposed
// fragments (ds_read_b64_tr_b16) for 512 cycles of MFMA, i.e. 128 B/clk per workgroup, the array's whole bandwidth, and two
// workgroups share it.  Kept at 1.
constexpr int LEVEL_WGRAD_PF = 1;
template <int TN, int TM, int WN, int WM, bool PLAIN = false, int PF = 1>
__device__ __forceinline__ void wgrad_bf16_body(const WgradArgs& a, const int bx, const int S, const int z_in) {
  constexpr int BI = 16 * TN * WN, BJ = 16 * TM * WM;
  constexpr int SP = wg_stride(BI), SQ = wg_stride(BJ);
  constexpr int STAGE = WPX * (SP + SQ);
  constexpr int PL = (WPX * BI / 8 + 255) / 256, QL = (WPX * BJ / 8 + 255) / 256;
  constexpr int PG = BI / 8, QG = BJ / 8;
  extern __shared__ __attribute__((aligned(16))) char smem_b[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wn = wid % WN, wm = wid / WN;
  const int nIB = (a.NI + BI - 1) / BI, nJB = (a.J + BJ - 1) / BJ;
  // All tiles of one split read the same pixel range of P and Q: the remap runs over (split, tile) so that they land
  // on ONE XCD and share its L2 (remapping the tiles of each split alone spreads every pixel range over all 8 L2s:
  // 231 MB of HBM traffic per level-1 conv2 launch against 133 MB algorithmic, measured with tools/pmc_traffic.py).
  const int nT = nIB * nJB;
  const int lin = xcd_remap_b(z_in * nT + bx, nT * S);
  const int z = lin / nT, bid = lin - z * nT;
  const int i0 = (bid % nIB) * BI, j0 = (bid / nIB) * BJ;
  const int HW = a.H * a.W;
  const int nchunks = (a.M + WPX - 1) / WPX;
  const int c_begin = (int)((long)nchunks * z / S), c_end = (int)((long)nchunks * (z + 1) / S);
  const unsigned pes = a.pbf16 ? 2u : 4u, qes = a.qbf16 ? 2u : 4u;
  const __amdgpu_buffer_rsrc_t rP = rsrc_of(a.P, (unsigned)((long)a.M * a.ldp * pes));
  const __amdgpu_buffer_rsrc_t rQ = rsrc_of(a.Q, (unsigned)((long)a.M * a.ldq * qes));
  const bool pv = ((a.NI | a.ldp) & 7) == 0, qv = ((a.Cq | a.ldq) & 7) == 0;

  // 8 consecutive elements starting at byte offset `off` (OOB -> zeros), widened/narrowed to bf16
  auto load8 = [&](__amdgpu_buffer_rsrc_t r, unsigned off, bool isbf) {
    if (isbf) return ld16(r, off);
    const f32x4 f0 = __builtin_bit_cast(f32x4, ld16(r, off));
    const f32x4 f1 = __builtin_bit_cast(f32x4, ld16(r, off == OOB ? OOB : off + 16u));
    return u32x4b{pack2(f0[0], f0[1]), pack2(f0[2], f0[3]), pack2(f1[0], f1[1]), pack2(f1[2], f1[3])};
  };
  auto load1h = [&](__amdgpu_buffer_rsrc_t r, unsigned off, bool isbf) -> unsigned {
    if (isbf) return (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(r, (int)off, 0, 0);
    const __bf16 h = (__bf16)ld4f(r, off);
    return (unsigned)__builtin_bit_cast(unsigned short, h);
  };

  int p_pl[PL], q_pl[QL];
  unsigned p_col[PL];
#pragma unroll
  for (int t = 0; t < PL; ++t) {
    const int idx = tid + 256 * t, g = idx % PG, i = i0 + g * 8;
    p_pl[t] = idx / PG;
    p_col[t] = (idx < WPX * PG && i < a.NI) ? (unsigned)i : OOB;   // element index, OOB if the group is outside
  }
#pragma unroll
  for (int t = 0; t < QL; ++t) q_pl[t] = (tid + 256 * t) / QG;

  u32x4b pregs[PF][PL], qregs[PF][QL];
  auto gload = [&](int chunk, u32x4b (&preg)[PL], u32x4b (&qreg)[QL]) {
    const int mb = chunk * WPX;
#pragma unroll
    for (int t = 0; t < PL; ++t) {
      const int m = mb + p_pl[t];
      const bool ok = m < a.M && p_col[t] != OOB;
      const unsigned base = ((unsigned)m * (unsigned)a.ldp + p_col[t]) * pes;
      if (PLAIN) { preg[t] = ld16(rP, ok ? base : OOB); continue; }
      if (pv) preg[t] = load8(rP, ok ? base : OOB, a.pbf16);
      else {
        u32x4b v = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const bool oke = ok && (int)p_col[t] + e < a.NI;
          v[e >> 1] |= load1h(rP, oke ? base + e * pes : OOB, a.pbf16) << ((e & 1) * 16);
        }
        preg[t] = v;
      }
    }
#pragma unroll
    for (int t = 0; t < QL; ++t) {
      const int idx = tid + 256 * t, g = idx % QG, j = j0 + g * 8;
      const int m = mb + q_pl[t];
      if (PLAIN) {
        const bool okq = idx < WPX * QG && m < a.M && j < a.J;
        qreg[t] = ld16(rQ, okq ? ((unsigned)m * (unsigned)a.ldq + (unsigned)j) * 2u : OOB);
        continue;
      }
      const int rem = m % HW, y = rem / a.W, x = rem - y * a.W;
      const bool rowok = idx < WPX * QG && m < a.M;
      auto elem_off = [&](int jj, bool& ok) {
        int tap = 0, cq = jj;
        if (a.taps != 1) { tap = jj / a.Cq; cq = jj - tap * a.Cq; }
        const int dy = (a.taps == 1) ? 0 : tap / 3 - 1, dx = (a.taps == 1) ? 0 : tap % 3 - 1;
        ok = rowok && jj < a.J && (unsigned)(y + dy) < (unsigned)a.H && (unsigned)(x + dx) < (unsigned)a.W;
        return (unsigned)((m + dy * a.W + dx) * a.ldq + cq) * qes;
      };
      if (qv) {
        bool ok;
        const unsigned off = elem_off(j, ok);
        qreg[t] = load8(rQ, ok ? off : OOB, a.qbf16);
      } else {
        u32x4b v = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          bool ok;
          const unsigned off = elem_off(j + e, ok);
          v[e >> 1] |= load1h(rQ, ok ? off : OOB, a.qbf16) << ((e & 1) * 16);
        }
        qreg[t] = v;
      }
    }
  };
  auto lstore = [&](int buf, const u32x4b (&preg)[PL], const u32x4b (&qreg)[QL]) {
    char* Ps = smem_b + buf * STAGE;
    char* Qs = Ps + WPX * SP;
#pragma unroll
    for (int t = 0; t < PL; ++t) {
      const int idx = tid + 256 * t;
      if (idx < WPX * PG) *reinterpret_cast<u32x4b*>(Ps + (idx / PG) * SP + (idx % PG) * 16) = preg[t];
    }
#pragma unroll
    for (int t = 0; t < QL; ++t) {
      const int idx = tid + 256 * t;
      if (idx < WPX * QG) *reinterpret_cast<u32x4b*>(Qs + (idx / QG) * SQ + (idx % QG) * 16) = qreg[t];
    }
  };

  f32x4 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto compute = [&](int buf) {
    const char* Ps = smem_b + buf * STAGE;
    const char* Qs = Ps + WPX * SP;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 fa[TN], fb[TM];
#pragma unroll
      for (int t = 0; t < TN; ++t) fa[t] = tr_frag(Ps, SP, wn * 16 * TN + t * 16, ks, lane);
#pragma unroll
      for (int t = 0; t < TM; ++t) fb[t] = tr_frag(Qs, SQ, wm * 16 * TM + t * 16, ks, lane);
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
          acc[tn][tm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[tn], fb[tm], acc[tn][tm], 0, 0, 0);
    }
  };
  if (PF == 1) {
    if (c_begin < c_end) {
      gload(c_begin, pregs[0], qregs[0]);
      lstore(0, pregs[0], qregs[0]);
    }
    __syncthreads();
    for (int c = c_begin; c < c_end; ++c) {
      const int buf = (c - c_begin) & 1;
      if (c + 1 < c_end) gload(c + 1, pregs[0], qregs[0]);
      compute(buf);
      if (c + 1 < c_end) lstore(buf ^ 1, pregs[0], qregs[0]);
      __syncthreads();
    }
  } else {
    // chunk c_begin + r starts in register slot r; iteration c (slot r = (c - c_begin) % PF): chunk c is in LDS buffer
    // (c - c_begin) & 1 (stored one iteration ago, or by the prologue), so slot r is free for chunk c + PF; chunk c + 1
    // goes from slot (r + 1) % PF to the other LDS buffer, which every wave left at the previous iteration's barrier
#pragma unroll
    for (int r = 0; r < PF; ++r)
      if (c_begin + r < c_end) gload(c_begin + r, pregs[r], qregs[r]);
    if (c_begin < c_end) lstore(0, pregs[0], qregs[0]);
    __syncthreads();
    auto iter = [&](int c, auto rr) {
      constexpr int r = decltype(rr)::value, rn = (r + 1) % PF;
      const int buf = (c - c_begin) & 1;
      if (c + PF < c_end) gload(c + PF, pregs[r], qregs[r]);
      compute(buf);
      if (c + 1 < c_end) lstore(buf ^ 1, pregs[rn], qregs[rn]);
      __syncthreads();
    };
    static_assert(PF == 1 || PF == 3, "register ring: unrolled by hand for three slots");
    for (int c = c_begin; c < c_end; c += 3) {
      iter(c, std::integral_constant<int, 0>{});
      if (c + 1 < c_end) iter(c + 1, std::integral_constant<int, 1 % PF>{});
      if (c + 2 < c_end) iter(c + 2, std::integral_constant<int, 2 % PF>{});
    }
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
// Narrow 3x3 conv from a WIDE bf16 input (conv1's dgrad: 512 -> C/2 channels) with the input tile in LDS.
// The implicit-GEMM kernel above fetches every input element once per tap (9x: 151 MB for the 17 MB gradient of a
// CIFAR level-1 step, measured with tools/pmc_traffic.py).  Here a workgroup owns 64 consecutive pixels (raster
// order over [B*H*W]); the rows m0 - W - 1 ... m0 + 64 + W of the input are staged once, 128 channels at a time, in a
// swizzled LDS image (double-buffered), and the 9 taps are shifted row reads of that image.  Taps that fall outside the
// pixel's image read a 16-byte zero slot instead (lane-constant address select); rows outside [0, M) are zero-filled
// by the buffer loads' bounds check.  The K range of a slice is split over the 4 waves (32 channels each, all taps),
// reduced through LDS at the end.  Output: fp32 [M][ldo], no split-K slabs.
// ------------------------------------------------------------------------------------------
constexpr size_t NH_MAX_LDS = 80 * 1024;   // two workgroups per CU
__host__ __device__ constexpr int nh_rows(int W) { return 64 + 2 * W + 2; }
__host__ __device__ constexpr size_t nh_lds_bytes(int W, int NT) {
  return (size_t)2 * nh_rows(W) * 256 + (size_t)4 * 64 * NT * 16 * 4 + 16;
}
bool narrow_halo_usable(int bf16, int M, int H, int W, int N, int Cin, int taps) {
  return bf16 && taps == 9 && N >= 1 && N <= 48 && Cin >= 128 && Cin % 128 == 0 && W >= 1 && W <= 64 && M >= 64 &&
         nh_lds_bytes(W, (N + 15) / 16) <= NH_MAX_LDS;
}

template <int NT>
__device__ __forceinline__ void narrow_halo_body(const ConvGemmArgs& a, const int bx) {
  extern __shared__ __attribute__((aligned(16))) char smem_b[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, r16 = lane & 15, q4 = lane >> 4;
  const int W = a.W, HW = a.H * a.W, rows = nh_rows(W), TB = rows * 256;
  const int m0 = bx * 64, nslices = a.Cin / 128;
  char* Tbuf = smem_b;
  float* R = reinterpret_cast<float*>(smem_b + 2 * TB);
  const int zoff = 2 * TB + 4 * 64 * NT * 16 * 4;   // 16 zero bytes
  if (tid < 4) reinterpret_cast<unsigned*>(smem_b + zoff)[tid] = 0u;
  const __amdgpu_buffer_rsrc_t rX = rsrc_of(a.X, (unsigned)((long)a.M * a.ldx * 2));
  const __amdgpu_buffer_rsrc_t rW = rsrc_of(a.Wt, (unsigned)((long)a.N * a.Kp * 2));

  // staging: piece p = (row, 16-byte slot); rows outside [0, M) read as zeros (bounds check)
  constexpr int PMAX = (nh_rows(64) * 16 + 255) / 256;
  const int npieces = rows * 16;
  u32x4b xreg[PMAX];
  auto gload = [&](int sl) {
#pragma unroll
    for (int t = 0; t < PMAX; ++t) {
      const int p = tid + 256 * t, row = p >> 4, slot = p & 15;
      const int m = m0 - W - 1 + row;
      const bool ok = p < npieces && m >= 0 && m < a.M;
      xreg[t] = ld16(rX, ok ? (unsigned)((m * a.ldx + sl * 128 + slot * 8) * 2) : OOB);
    }
  };
  auto lstore = [&](int buf) {
    char* T = Tbuf + buf * TB;
#pragma unroll
    for (int t = 0; t < PMAX; ++t) {
      const int p = tid + 256 * t, row = p >> 4, slot = p & 15;
      if (p < npieces) *reinterpret_cast<u32x4b*>(T + row * 256 + ((slot ^ (row & 15)) << 4)) = xreg[t];
    }
  };
  // B-operand addresses: lane = (output pixel r16 of m-tile mo, channel group q4); this wave's 32 channels of the slice
  int boff[9][4];
  {
    const int slot = wid * 4 + q4;
#pragma unroll
    for (int mo = 0; mo < 4; ++mo) {
      const int p = mo * 16 + r16, m = m0 + p;
      const int rem = m % HW, y = rem / W, x = rem - y * W;
      const bool rok[3] = {y >= 1, true, y + 1 < a.H}, xok[3] = {x >= 1, true, x + 1 < W};
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int dy = tap / 3 - 1, dx = tap % 3 - 1;
        const int row = p + W + 1 + dy * W + dx;
        boff[tap][mo] = (m < a.M && rok[dy + 1] && xok[dx + 1]) ? row * 256 + ((slot ^ (row & 15)) << 4) : -1;
      }
    }
  }
  // A operands (weights): rows of the [N][Kp] array, k = tap * Cin + slice * 128 + wave * 32 + q4 * 8
  bf16x8 wreg[2][9][NT];
  auto wload = [&](int sl, bf16x8 (&f)[9][NT]) {
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int n = min(nt * 16 + r16, a.N - 1);   // clamped rows: never read back
        f[tap][nt] = __builtin_bit_cast(bf16x8, ld16(rW, (unsigned)((n * a.Kp + tap * a.Cin + sl * 128 + wid * 32 + q4 * 8) * 2)));
      }
  };
  f32x4 acc[NT][4];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  gload(0);
  wload(0, wreg[0]);
  lstore(0);
  __syncthreads();
#pragma unroll 1
  for (int sl = 0; sl < nslices; sl += 2) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int cur = sl + h;
      if (cur >= nslices) break;
      const bool more = cur + 1 < nslices;
      if (more) { gload(cur + 1); wload(cur + 1, wreg[h ^ 1]); }
      const char* T = Tbuf + h * TB;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        bf16x8 fb[4];
#pragma unroll
        for (int mo = 0; mo < 4; ++mo) {
          const int o = boff[tap][mo];
          fb[mo] = *reinterpret_cast<const bf16x8*>(o >= 0 ? T + o : smem_b + zoff);
        }
#pragma unroll
        for (int mo = 0; mo < 4; ++mo)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[nt][mo] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[h][tap][nt], fb[mo], acc[nt][mo], 0, 0, 0);
      }
      if (more) lstore(h ^ 1);
      __syncthreads();
    }
  }
  // cross-wave reduction: rows = output channels (4 per lane: q4*4 + r), columns = pixels (r16)
  constexpr int NP = NT * 16;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int mo = 0; mo < 4; ++mo)
      *reinterpret_cast<f32x4*>(R + ((wid * 64 + mo * 16 + r16) * NP + nt * 16 + q4 * 4)) = acc[nt][mo];
  __syncthreads();
  for (int idx = tid; idx < 64 * a.N; idx += 256) {
    const int po = idx / a.N, n = idx - po * a.N, m = m0 + po;
    if (m >= a.M) continue;
    const float* q = R + po * NP + n;
    a.out[(long)m * a.ldo + n] = (q[0] + q[64 * NP]) + (q[2 * 64 * NP] + q[3 * 64 * NP]);
  }
}

template <int NT, int GTN, int GTM, int GWN, int GWM>
__global__ void __launch_bounds__(256) dual_halo_dgrad_wgrad_bf16_kernel(ConvGemmArgs d, int nbD, WgradArgs w, int nbW,
                                                                         int SW) {
  const int b = blockIdx.x;
  if (b < nbD) narrow_halo_body<NT>(d, b);
  else {
    const int c = b - nbD;
    wgrad_bf16_body<GTN, GTM, GWN, GWM>(w, c % nbW, SW, c / nbW);
  }
}

// ------------------------------------------------------------------------------------------
// Narrow weight gradient of a 3x3 conv with a WIDE bf16 input (conv3: out[co][tap][c] = sum_px g[px][co] h2[px + d(tap)][c],
// co <= 16, c = 512) with the input tile in LDS.  The generic kernel above re-reads h2 once per tap (151 of the 170 MB a
// CIFAR level-1 conv3 dual launch moves).  Here a workgroup owns a slice of 128 input channels and a range of 64-pixel
// chunks; per chunk the rows m0 - W - 1 ... m0 + 64 + W of the slice are staged once (pixel-major, as they sit in
// HBM) and the taps are row-shifted transposing reads (ds_read_b64_tr_b16) of that image.  Taps that leave the pixel's
// image are removed on the OTHER operand: nine copies of the chunk's gradient tile, each with the rows of its invalid
// pixels zeroed.  Output: split slabs [S][NI][9 * Cq] like the generic kernel.
// ------------------------------------------------------------------------------------------
constexpr int WH_SQ = wg_stride(128), WH_SP = wg_stride(16);   // LDS row strides (bytes) of the h2 slice / gradient tiles
__host__ __device__ constexpr size_t wh_lds_bytes(int W) { return (size_t)nh_rows(W) * WH_SQ + (size_t)9 * 64 * WH_SP; }
static int g_wgrad_halo = 1;
static bool wgrad_halo_usable(int M, int NI, int Cq, int taps, int W) {
  return g_wgrad_halo && taps == 9 && NI >= 1 && NI <= 16 && Cq >= 128 && Cq % 128 == 0 && W >= 1 && W <= 64 && M >= 64 &&
         M % 64 == 0 && wh_lds_bytes(W) <= 64 * 1024;
}
static int g_wgrad_halo_max_splits = 32;   // nfk_tune NFK_TUNE_WGRAD_HALO_MAX_SPLITS
static int wgrad_halo_splits(int M) {
  const int nchunks = M / 64;
  return nchunks < g_wgrad_halo_max_splits ? nchunks : g_wgrad_halo_max_splits;
}

__device__ __forceinline__ void wgrad_halo_body(const WgradArgs& a, const int bx, const int S) {
  extern __shared__ __attribute__((aligned(16))) char smem_b[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int W = a.W, HW = a.H * a.W, rows = nh_rows(W);
  const int nsl = a.Cq / 128, sl = bx % nsl, z = bx / nsl;
  const int nchunks = a.M / 64;
  const int c_begin = (int)((long)nchunks * z / S), c_end = (int)((long)nchunks * (z + 1) / S);
  char* T = smem_b;                       // [rows][WH_SQ]: row r = pixel m0 - W - 1 + r, 128 channels
  char* P9 = smem_b + rows * WH_SQ;       // [9][64][WH_SP]: gradient tile per tap, invalid pixels zeroed
  const __amdgpu_buffer_rsrc_t rP = rsrc_of(a.P, (unsigned)((long)a.M * a.ldp * 4));
  const __amdgpu_buffer_rsrc_t rQ = rsrc_of(a.Q, (unsigned)((long)a.M * a.ldq * 2));
  constexpr int PMAX = (nh_rows(64) * 16 + 255) / 256;
  const int npieces = rows * 16;
  u32x4b qreg[PMAX];
  float preg[8];
  // gradient tile: thread (pixel p = tid >> 1, half = tid & 1) holds 8 output channels of one pixel (tid < 128)
  const int pp = tid >> 1, ph = tid & 1;
  auto gload = [&](int chunk) {
    const int m0 = chunk * 64;
#pragma unroll
    for (int t = 0; t < PMAX; ++t) {
      const int p = tid + 256 * t, row = p >> 4, slot = p & 15;
      const int m = m0 - W - 1 + row;
      const bool ok = p < npieces && m >= 0 && m < a.M;
      qreg[t] = ld16(rQ, ok ? (unsigned)((m * a.ldq + sl * 128 + slot * 8) * 2) : OOB);
    }
    if (tid < 128) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int co = ph * 8 + e;
        preg[e] = ld4f(rP, co < a.NI ? (unsigned)(((m0 + pp) * a.ldp + co) * 4) : OOB);
      }
    }
  };
  auto lstore = [&](int chunk) {
#pragma unroll
    for (int t = 0; t < PMAX; ++t) {
      const int p = tid + 256 * t, row = p >> 4, slot = p & 15;
      if (p < npieces) *reinterpret_cast<u32x4b*>(T + row * WH_SQ + slot * 16) = qreg[t];
    }
    if (tid < 128) {
      const int m = chunk * 64 + pp, rem = m % HW, y = rem / W, x = rem - y * W;
      const bool rok[3] = {y >= 1, true, y + 1 < a.H}, xok[3] = {x >= 1, true, x + 1 < W};
      const u32x4b v = {pack2(preg[0], preg[1]), pack2(preg[2], preg[3]), pack2(preg[4], preg[5]), pack2(preg[6], preg[7])};
      const u32x4b zero = {0u, 0u, 0u, 0u};
#pragma unroll
      for (int tap = 0; tap < 9; ++tap)
        *reinterpret_cast<u32x4b*>(P9 + (tap * 64 + pp) * WH_SP + ph * 16) = (rok[tap / 3] && xok[tap % 3]) ? v : zero;
    }
  };
  // wave w: channel tiles 2w, 2w+1 of the slice's eight, all nine taps
  f32x4 acc[9][2];
#pragma unroll
  for (int i = 0; i < 9; ++i) { acc[i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[i][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  if (c_begin < c_end) gload(c_begin);
  for (int c = c_begin; c < c_end; ++c) {
    __syncthreads();            // the previous chunk's fragments have been read
    lstore(c);
    __syncthreads();
    if (c + 1 < c_end) gload(c + 1);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int dy = tap / 3 - 1, dx = tap % 3 - 1;
      const char* Tt = T + (W + 1 + dy * W + dx) * WH_SQ;   // row of output pixel 0 for this tap
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const bf16x8 fa = tr_frag(P9 + tap * 64 * WH_SP, WH_SP, 0, ks, lane);
        const bf16x8 fb0 = tr_frag(Tt, WH_SQ, wid * 32, ks, lane);
        const bf16x8 fb1 = tr_frag(Tt, WH_SQ, wid * 32 + 16, ks, lane);
        acc[tap][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb0, acc[tap][0], 0, 0, 0);
        acc[tap][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb1, acc[tap][1], 0, 0, 0);
      }
    }
  }
  const int iq = (lane >> 4) * 4, jl = lane & 15;
  float* outp = a.out + (long)z * a.slab_stride;
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      const int j = tap * a.Cq + sl * 128 + wid * 32 + ct * 16 + jl;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (iq + r < a.NI) outp[(long)(iq + r) * a.J + j] = acc[tap][ct][r];
    }
}

template <int CTN, int CTM, int CWN, int CWM, bool XF32>
__global__ void __launch_bounds__(256) dual_dgrad_wgradhalo_bf16_kernel(ConvGemmArgs d, int nbD, int SD, WgradArgs w,
                                                                        int SW) {
  const int b = blockIdx.x;
  if (b < nbD * SD) conv_gemm_bf16_body<CTN, CTM, CWN, CWM, XF32>(d, b % nbD, SD, b / nbD);
  else wgrad_halo_body(w, b - nbD * SD, SW);
}

template <int TN, int TM, int WN, int WM>
__global__ void __launch_bounds__(256) wgrad_bf16_kernel(WgradArgs a) {
  wgrad_bf16_body<TN, TM, WN, WM>(a, blockIdx.x, gridDim.y, blockIdx.y);
}

// One launch = the dgrad GEMM of a conv (first nbD * SD workgroups) AND the weight-gradient GEMM that reads the same
// output gradient (the rest).  The two are independent; launched together they share the chip without a second
// stream, so the backward chain stays a chain: no fork / join edges (each costs 15-35 us in a replayed hipGraph).
template <int CTN, int CTM, int CWN, int CWM, bool XF32, int GTN, int GTM, int GWN, int GWM>
__global__ void __launch_bounds__(256) dual_dgrad_wgrad_bf16_kernel(ConvGemmArgs d, int nbD, int SD, WgradArgs w,
                                                                    int nbW, int SW) {
  const int b = blockIdx.x;
  if (b < nbD * SD) {
    conv_gemm_bf16_body<CTN, CTM, CWN, CWM, XF32>(d, b % nbD, SD, b / nbD);
  } else {
    const int c = b - nbD * SD;
    wgrad_bf16_body<GTN, GTM, GWN, GWM>(w, c % nbW, SW, c / nbW);
  }
}

template <int CTN, int CTM, int CWN, int CWM, bool XF32, int GTN, int GTM, int GWN, int GWM>
static int launch_dual(const ConvGemmArgs& d, int SD, const WgradArgs& w, int SW, hipStream_t st) {
  constexpr int BN = 16 * CTN * CWN, BM = 16 * CTM * CWM, BI = 16 * GTN * GWN, BJ = 16 * GTM * GWM;
  constexpr size_t ldsD = 2 * (BN + BM) * 128, ldsW = 2 * WPX * (wg_stride(BI) + wg_stride(BJ));
  constexpr size_t lds = ldsD > ldsW ? ldsD : ldsW;
  static bool attr = false;
  if (!attr && lds > 48 * 1024) {
    (void)hipFuncSetAttribute(
        reinterpret_cast<const void*>(dual_dgrad_wgrad_bf16_kernel<CTN, CTM, CWN, CWM, XF32, GTN, GTM, GWN, GWM>),
        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr = true;
  }
  const int nbD = ceil_div(d.N, BN) * ceil_div(d.M, BM), nbW = ceil_div(w.NI, BI) * ceil_div(w.J, BJ);
  const int cls = NFK_PROF_DUAL_BWD + 3 * ((d.N > 96) ? ((w.NI > 96) ? 1 : 0) : 2) + (d.M >= 8192 ? 0 : (d.M >= 2048 ? 1 : 2));
  prof_scope_begin(cls, 2.0 * d.M * (double)d.N * d.K + 2.0 * w.M * (double)w.NI * w.J,
                   (d.xf32 ? 4.0 : 2.0) * d.M * d.Cin + 2.0 * d.N * d.K + (d.obf16 ? 2.0 : 4.0 * SD) * d.M * d.N +
                       (w.pbf16 ? 2.0 : 4.0) * w.M * w.NI + (w.qbf16 ? 2.0 : 4.0) * w.M * w.Cq + 4.0 * SW * w.NI * w.J,
                   st);
  hipLaunchKernelGGL((dual_dgrad_wgrad_bf16_kernel<CTN, CTM, CWN, CWM, XF32, GTN, GTM, GWN, GWM>),
                     dim3(nbD * SD + nbW * SW), dim3(256), lds, st, d, nbD, SD, w, nbW, SW);
  prof_scope_end(cls, st);
  NFK_LAUNCH_CHECK();
  return NFK_OK;
}

// Per-channel sums behind the two ActNorm layers' gradients (block.2: g_a2 with h2, block.0: g_a1 with h1), over this
// workgroup's share of the pixels:  col[part][0][c] = sum_m ga[m][c] h[m][c] / e^logs[c],  col[part][1][c] = sum_m ga[m][c]
// (the [rows][2][N] partial-row layout of the dgrad epilogues, reduced by the step tail).  A thread owns 8 channels
// (one 16-byte piece of a row); the 4 waves take every 4th pixel and are summed through LDS in fixed order.
constexpr int AN_PARTS_MAX = 64;
int actnorm_sum_parts(int M) {
  const int p = M / 64;   // at least 64 pixels per workgroup
  return p < 1 ? 1 : (p > AN_PARTS_MAX ? AN_PARTS_MAX : p);
}
__device__ __forceinline__ void actnorm_sum_body(const ActnormSumArgs& a, const int part) {
  extern __shared__ __attribute__((aligned(16))) char smem_b[];
  float* red = reinterpret_cast<float*>(smem_b);   // [4 waves][2 layers][2 sums][512]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int m_begin = (int)((long)a.M * part / a.nparts), m_end = (int)((long)a.M * (part + 1) / a.nparts);
  const unsigned bytes = (unsigned)((long)a.M * a.hid * 2);
  const __amdgpu_buffer_rsrc_t rG2 = rsrc_of(a.ga2, bytes), rH2 = rsrc_of(a.h2, bytes), rG1 = rsrc_of(a.ga1, bytes),
                               rH1 = rsrc_of(a.h1, bytes);
  float s2l[8], s2b[8], s1l[8], s1b[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { s2l[e] = 0.f; s2b[e] = 0.f; s1l[e] = 0.f; s1b[e] = 0.f; }
  auto acc8 = [](const u32x4b g, const u32x4b h, float (&sl)[8], float (&sb)[8]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float g0 = __builtin_bit_cast(float, g[e] << 16), g1 = __builtin_bit_cast(float, g[e] & 0xFFFF0000u);
      const float h0 = __builtin_bit_cast(float, h[e] << 16), h1 = __builtin_bit_cast(float, h[e] & 0xFFFF0000u);
      sl[2 * e] += g0 * h0; sl[2 * e + 1] += g1 * h1;
      sb[2 * e] += g0; sb[2 * e + 1] += g1;
    }
  };
  // 4 pixels per wave in flight (16 loads): a chain of dependent round trips otherwise
  for (int m = m_begin + wid; m < m_end; m += 16) {
    u32x4b g2[4], h2[4], g1[4], h1[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int mm = m + 4 * u;
      const unsigned off = mm < m_end ? (unsigned)((mm * a.hid + lane * 8) * 2) : OOB;
      g2[u] = ld16(rG2, off); h2[u] = ld16(rH2, off); g1[u] = ld16(rG1, off); h1[u] = ld16(rH1, off);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) { acc8(g2[u], h2[u], s2l, s2b); acc8(g1[u], h1[u], s1l, s1b); }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    float* r = red + wid * 2048 + lane * 8 + e;
    r[0] = s2l[e]; r[512] = s2b[e]; r[1024] = s1l[e]; r[1536] = s1b[e];
  }
  __syncthreads();
  for (int t = tid; t < 2048; t += 256) {
    const int layer = t >> 10, which = (t >> 9) & 1, c = t & 511;
    float v = (red[t] + red[2048 + t]) + (red[4096 + t] + red[6144 + t]);
    if (which == 0) v /= (layer == 0 ? a.es2 : a.es1)[c];
    (layer == 0 ? a.col2 : a.col1)[((long)part * 2 + which) * a.hid + c] = v;
  }
}

// The three weight-gradient GEMMs of one FlowStep in ONE launch (the fused backward, nfk_fusedb.hip, has produced all
// three output gradients by then).  Block order: the ActNorm sums first (64 workgroups of pure streaming: started last
// they were the tail of the launch), then conv2's 512x512 tiles (the longest GEMM workgroups), conv3's, conv1's.
// V3: body of conv3's weight gradient -- 4 = the wide GEMM body on dgrad3's im2col rows (P = [M][9 Cout padded], taps
// folded into P: see nfk_api.hip), 0 = the halo-tile kernel, 1..3 = the generic narrow body with V3 row tiles.
template <int V3>
__global__ void __launch_bounds__(256) triple_wgrad_bf16_kernel(WgradArgs w2, int nb2, int S2, WgradArgs w1, int nb1, int S1,
                                                                WgradArgs w3, int nb3, int S3, int n3, ActnormSumArgs an) {
  int b = blockIdx.x;
  if (b < an.nparts) { actnorm_sum_body(an, b); return; }
  b -= an.nparts;
  if (b < nb2 * S2) { wgrad_bf16_body<4, 4, 2, 2>(w2, b % nb2, S2, b / nb2); return; }
  b -= nb2 * S2;
  if (b < n3) {
    if (V3 == 4) wgrad_bf16_body<4, 4, 2, 2>(w3, b % nb3, S3, b / nb3);
    else if (V3 == 0) wgrad_halo_body(w3, b, S3);
    else wgrad_bf16_body<((V3 > 0 && V3 < 4) ? V3 : 1), 2, 1, 4>(w3, b % nb3, S3, b / nb3);
    return;
  }
  b -= n3;
  wgrad_bf16_body<4, 4, 2, 2>(w1, b % nb1, S1, b / nb1);
}

// The weight-gradient launch of a step merged with the launch that used to follow it (tail_chanmix_kernel): nothing in
// the latter waits for the GEMMs of ITS OWN step except the slab reductions -- so those are taken from the PREVIOUS step
// of the backward (whose GEMMs finished two launches ago; the caller carries them along: nfk_step2d_pending), and the
// channel mix + next step's coupling backward + this step's coupling-tail gradients, which only wait for the dgrad
// launch, run beside the GEMMs.  A step's backward is then two launches on the critical path instead of three:
//   [fused dgrad chain]  [this: channel mix + coupling backward(next) || ActNorm sums || 3 GEMMs || slab reductions(previous)]
// Block order: the channel-mix workgroups first (few, each a chain of dependent loads: the next launch waits for them).
__global__ void __launch_bounds__(256) wgrad_tail_kernel(WgradArgs w2, int nb2, int S2, WgradArgs w1, int nb1, int S1, WgradArgs w3,
                                                         int nb3, int S3, ActnormSumArgs an, int n_cm, ChanMixArgs m,
                                                         CouplingBwdArgs nx, SmallGradArgs sg, StepTailArgs tf) {
  int b = blockIdx.x;
  if (b < n_cm) {
    if (nx.ga3) chanmix_body<true>(m, b, n_cm, nx);
    else chanmix_body<false>(m, b, n_cm, nx);
    return;
  }
  b -= n_cm;
  if (b < an.nparts) { actnorm_sum_body(an, b); return; }
  b -= an.nparts;
  if (b < nb2 * S2) { wgrad_bf16_body<4, 4, 2, 2, true>(w2, b % nb2, S2, b / nb2); return; }
  b -= nb2 * S2;
  if (b < nb3 * S3) { wgrad_bf16_body<4, 4, 2, 2, true>(w3, b % nb3, S3, b / nb3); return; }
  b -= nb3 * S3;
  if (b < nb1 * S1) { wgrad_bf16_body<4, 4, 2, 2, true>(w1, b % nb1, S1, b / nb1); return; }
  b -= nb1 * S1;
  step_tail_body(sg, tf, b);
}

int wgrad_tail_bf16(WgradArgs w3, int S3, WgradArgs w2, int S2, WgradArgs w1, int S1, const ActnormSumArgs& an,
                    const SmallGradArgs& sg, StepTailArgs tf, const ChanMixArgs& m, const CouplingBwdArgs* next, hipStream_t st,
                    double alg_nj) {
  for (WgradArgs* w : {&w3, &w2, &w1}) {
    w->M = w->B * w->H * w->W;
    w->J = w->taps * w->Cq;
    if (w->M <= 0 || w->NI <= 0 || w->J <= 0) return NFK_ERR_ARG;
    if ((long)w->M * w->ldp * 4 >= 0x7fffffffL || (long)w->M * w->ldq * 4 >= 0x7fffffffL) return NFK_ERR_ARG;
    if (!w->pbf16 || !w->qbf16 || w->taps != 1 || ((w->NI | w->ldp | w->Cq | w->ldq) & 7)) return NFK_ERR_ARG;
  }
  if (an.hid != 512 || an.nparts < 1) return NFK_ERR_ARG;
  // channel-mix part: as launch_tail_chanmix
  const int C = m.C, Cp = (C + 3) & ~3, TP = cm_tile_pixels(C), C2 = C - C / 2;
  size_t lds_cm = (size_t)(C * Cp + 2 * TP * (C + 1) + (m.part ? C * C + C : 0)) * sizeof(float);
  CouplingBwdArgs nx{};
  if (next) {
    if (C2 < 1 || C2 > 256) return NFK_ERR_ARG;
    nx = *next;
    lds_cm += (size_t)(TP * (C + 1) + (256 / C2) * 4 * C2 + 4 * C2) * sizeof(float);
  }
  if (lds_cm > 64 * 1024 || m.M <= 0) return NFK_ERR_ARG;
  int cb = ceil_div(m.M, TP);
  if (m.part && cb > CM_MAX_BLOCKS) cb = CM_MAX_BLOCKS;
  constexpr size_t ldsWide = 2 * WPX * (wg_stride(128) + wg_stride(128));
  const size_t lds = ldsWide > lds_cm ? ldsWide : lds_cm;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_tail_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    attr = true;
  }
  const int nb2 = ceil_div(w2.NI, 128) * ceil_div(w2.J, 128), nb1 = ceil_div(w1.NI, 128) * ceil_div(w1.J, 128);
  const int nb3 = ceil_div(w3.NI, 128) * ceil_div(w3.J, 128);
  const int nt = tail_blocks(sg, tf);
  const int cls = NFK_PROF_TRIPLE_WGRAD + (w2.M >= 8192 ? 0 : (w2.M >= 2048 ? 1 : 2));
  const double nj = alg_nj > 0 ? alg_nj : ((double)w2.NI * w2.J + (double)w1.NI * w1.J + (double)w3.NI * w3.J);
  prof_scope_begin(cls, 2.0 * w2.M * nj,
                   w2.M * (2.0 * w2.NI + 2.0 * w2.Cq + 2.0 * w1.NI + 4.0 * w1.Cq + 4.0 * w3.NI + 2.0 * w3.Cq) +
                       4.0 * (S2 * (double)w2.NI * w2.J + S1 * (double)w1.NI * w1.J + S3 * (double)w3.NI * w3.J), st);
  hipLaunchKernelGGL(wgrad_tail_kernel, dim3(cb + an.nparts + nb2 * S2 + nb3 * S3 + nb1 * S1 + nt), dim3(256), lds, st, w2, nb2, S2,
                     w1, nb1, S1, w3, nb3, S3, an, cb, m, nx, sg, tf);
  prof_scope_end(cls, st);
  NFK_LAUNCH_CHECK();
  return NFK_OK;
}

// ------------------------------------------------------------------------------------------
// Level mode (nfk_level2d_bwd): the weight-gradient GEMMs and ActNorm sums of ALL steps of a level in ONE launch -- with
// 3 K GEMMs side by side a split covers a quarter of the pixels or more (64+ chunks at CIFAR level 1) instead of a
// sixteenth (16 chunks, each most of a memory round trip behind one chunk of prefetch), and writes 4 slabs instead of 16
// -- and every reduction into the parameter gradients in ONE more.  Per-step operands are base + k * stride.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) level_wgrad_kernel(LevelWgradArgs a) {
  const int ks = (int)blockIdx.x / a.per_step, k = a.k_lo + ks;
  int b = (int)blockIdx.x - ks * a.per_step;
  const float* gablk = a.ga + (long)k * a.ga_stride;
  const __bf16* h1 = reinterpret_cast<const __bf16*>(a.h1) + (long)k * a.h_stride;
  const __bf16* h2 = reinterpret_cast<const __bf16*>(a.h2) + (long)k * a.h_stride;
  float* ws = a.wslab + (long)k * a.wslab_stride;
  if (b < a.nparts) {
    ActnormSumArgs an{};
    an.ga2 = gablk + a.o_ga2; an.h2 = h2; an.ga1 = gablk + a.o_ga1; an.h1 = h1;
    an.es2 = a.steps[k].es2; an.es1 = a.steps[k].es1; an.col2 = ws + a.o_col2; an.col1 = ws + a.o_col1;
    an.M = a.M; an.hid = a.hid; an.nparts = a.nparts;
    actnorm_sum_body(an, b);
    return;
  }
  b -= a.nparts;
  WgradArgs w{};
  w.B = a.B; w.H = a.H; w.W = a.W; w.M = a.M; w.taps = 1; w.pbf16 = 1; w.qbf16 = 1;
  if (b < a.nb2 * a.S) {          // block.2: [hid x M] [M x hid]
    w.P = gablk + a.o_ga2; w.ldp = a.hid; w.NI = a.hid;
    w.Q = reinterpret_cast<const float*>(h1); w.ldq = a.hid; w.Cq = a.hid; w.J = a.hid;
    w.out = ws + a.o_s2; w.slab_stride = (long)a.hid * a.hid;
    wgrad_bf16_body<4, 4, 2, 2, true, LEVEL_WGRAD_PF>(w, b % a.nb2, a.S, b / a.nb2);
    return;
  }
  b -= a.nb2 * a.S;
  if (b < a.nb3 * a.S) {          // block.4 through dgrad3's im2col rows: [K3p x M] [M x hid]
    w.P = gablk + a.o_x3; w.ldp = a.K3p; w.NI = a.K3p;
    w.Q = reinterpret_cast<const float*>(h2); w.ldq = a.hid; w.Cq = a.hid; w.J = a.hid;
    w.out = ws; w.slab_stride = (long)a.K3p * a.hid;
    wgrad_bf16_body<4, 4, 2, 2, true, LEVEL_WGRAD_PF>(w, b % a.nb3, a.S, b / a.nb3);
    return;
  }
  b -= a.nb3 * a.S;
  if (b < a.nb1 * a.S) {          // block.0 on the forward's im2col rows: [hid x M] [M x K1p]
    w.P = gablk + a.o_ga1; w.ldp = a.hid; w.NI = a.hid;
    w.Q = reinterpret_cast<const float*>(a.masks + (long)k * a.mask_stride + a.o_x1); w.ldq = a.K1p; w.Cq = a.K1p; w.J = a.K1p;
    w.out = ws + a.o_s1; w.slab_stride = (long)a.hid * a.K1p;
    wgrad_bf16_body<4, 4, 2, 2, true, LEVEL_WGRAD_PF>(w, b % a.nb1, a.S, b / a.nb1);
  }
}

__global__ void __launch_bounds__(256) level_tail_kernel(LevelWgradArgs a, const StepTailArgs f, int per_tail) {
  const int ks = (int)blockIdx.x / per_tail, k = a.k_lo + ks, blk = (int)blockIdx.x - ks * per_tail;
  float* ws = a.wslab + (long)k * a.wslab_stride;
  const nfk_level_grads g = a.grads[k];
  SmallGradArgs sg{};
  sg.col1 = ws + a.o_col1; sg.col2 = ws + a.o_col2; sg.mb1 = sg.mb2 = a.nparts; sg.n1 = sg.n2 = a.hid;
  sg.tail = a.tailpart + (long)k * NFK_LEVEL_ROWS * 2 * a.Cout; sg.B = (k == a.k_top) ? a.rows_top : a.rows; sg.n3 = a.Cout;
  sg.g_l1 = g.g_l1; sg.g_b1 = g.g_b1; sg.g_l2 = g.g_l2; sg.g_b2 = g.g_b2; sg.g_bias3 = g.g_bias3; sg.g_l3 = g.g_l3;
  step_tail_body<true>(sg, f, blk, ws, ws + a.o_s2, ws + a.o_s1, g.g_w3, g.g_w2, g.g_w1);
}

int level_wgrad_tiles(int hid, int K3p, int K1p) {
  return ceil_div(hid, 128) * ceil_div(hid, 128) + ceil_div(K3p, 128) * ceil_div(hid, 128) + ceil_div(hid, 128) * ceil_div(K1p, 128);
}
// splits per GEMM for a walk of `nwalk` steps: enough workgroups for the chip (two per CU fit), never fewer than 4 chunks each
int level_wgrad_splits(int M, int hid, int K3p, int K1p, int nwalk) {
  const int tiles = level_wgrad_tiles(hid, K3p, K1p) * (nwalk > 0 ? nwalk : 1), nchunks = ceil_div(M, WPX);
  int S = ceil_div(640, tiles);
  if (S > nchunks / 4) S = nchunks / 4;
  if (S > NFK_LEVEL_MAX_SPLITS) S = NFK_LEVEL_MAX_SPLITS;
  return S < 1 ? 1 : S;
}

int level_wgrad_launch(LevelWgradArgs a, hipStream_t st, hipStream_t st_tail) {
  const int nwalk = a.k_hi - a.k_lo + 1;
  if (nwalk < 1 || a.hid != 512 || ((a.K3p | a.K1p | a.hid) & 7) || a.M <= 0) return NFK_ERR_ARG;
  if ((long)a.M * a.hid * 4 >= 0x7fffffffL) return NFK_ERR_ARG;
  a.nb2 = ceil_div(a.hid, 128) * ceil_div(a.hid, 128);
  a.nb3 = ceil_div(a.K3p, 128) * ceil_div(a.hid, 128);
  a.nb1 = ceil_div(a.hid, 128) * ceil_div(a.K1p, 128);
  a.nparts = actnorm_sum_parts(a.M);
  // (a multiple of 8: a workgroup's XCD is its index mod 8, and wgrad_bf16_body's remap -- all tiles of one pixel range on one
  //  XCD's L2 -- counts on the GEMM segments of every step starting on a multiple of 8; surplus workgroups exit at once)
  a.per_step = (a.nparts + (a.nb2 + a.nb3 + a.nb1) * a.S + 7) & ~7;
  constexpr size_t ldsWide = 2 * WPX * (wg_stride(128) + wg_stride(128));
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(level_wgrad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    attr = true;
  }
  const int cls = NFK_PROF_TRIPLE_WGRAD + (a.M >= 8192 ? 0 : (a.M >= 2048 ? 1 : 2));
  const double nj = (double)a.hid * (9.0 * a.Cout + a.hid + 9.0 * a.Ch);
  prof_scope_begin(cls, nwalk * 2.0 * a.M * nj,
                   nwalk * (a.M * (4.0 * a.hid + 2.0 * a.hid + 2.0 * a.K3p + 2.0 * a.K1p + 2.0 * a.hid) +
                            4.0 * a.S * ((double)a.hid * a.hid + (double)a.K3p * a.hid + (double)a.hid * a.K1p)), st);
  hipLaunchKernelGGL(level_wgrad_kernel, dim3(nwalk * a.per_step), dim3(256), ldsWide, st, a);
  prof_scope_end(cls, st);
  NFK_LAUNCH_CHECK();
  // the reductions: per step as step_tail_body lays them out
  SmallGradArgs sg{};
  sg.n1 = sg.n2 = a.hid; sg.n3 = a.Cout;
  StepTailArgs f{};
  const int Co[3] = {a.Cout, a.hid, a.hid}, Ci[3] = {a.hid, a.hid, a.Ch}, tp[3] = {9, 1, 9}, md[3] = {1, 0, 0}, lj[3] = {a.hid, 0, a.K1p};
  const long strd[3] = {(long)a.K3p * a.hid, (long)a.hid * a.hid, (long)a.hid * a.K1p};
  for (int q = 0; q < 3; ++q) {
    f.stride[q] = strd[q]; f.S[q] = a.S; f.Cout[q] = Co[q]; f.Cin[q] = Ci[q]; f.taps[q] = tp[q]; f.mode[q] = md[q]; f.ldj[q] = lj[q];
  }
  const int per_tail = tail_blocks(sg, f);
  hipLaunchKernelGGL(level_tail_kernel, dim3(nwalk * per_tail), dim3(256), 0, st_tail, a, f, per_tail);
  NFK_LAUNCH_CHECK();
  return NFK_OK;
}

static int g_triple_skip = 0;
void triple_wgrad_debug_skip(int mask) { g_triple_skip = mask; }

template <int V3>
static int launch_triple(const WgradArgs& w3, int S3, const WgradArgs& w2, int S2, const WgradArgs& w1, int S1,
                         const ActnormSumArgs& an, hipStream_t st, double alg_nj) {
  constexpr size_t ldsWide = 2 * WPX * (wg_stride(128) + wg_stride(128));
  const size_t lds3 = V3 == 4 ? ldsWide
                              : (V3 == 0 ? wh_lds_bytes(w3.W) : 2 * WPX * (wg_stride(16 * ((V3 > 0 && V3 < 4) ? V3 : 1)) + wg_stride(128)));
  const size_t lds = ldsWide > lds3 ? ldsWide : lds3;   // (the ActNorm sums need 32 KiB)
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(triple_wgrad_bf16_kernel<V3>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    attr = true;
  }
  if (lds > 80 * 1024 || an.hid != 512 || an.nparts < 1) return NFK_ERR_ARG;
  const int nb2 = ceil_div(w2.NI, 128) * ceil_div(w2.J, 128), nb1 = ceil_div(w1.NI, 128) * ceil_div(w1.J, 128);
  const int nb3 = V3 == 4 ? ceil_div(w3.NI, 128) * ceil_div(w3.J, 128)
                          : (V3 == 0 ? (w3.Cq / 128) * S3 : ceil_div(w3.NI, 16 * ((V3 > 0 && V3 < 4) ? V3 : 1)) * ceil_div(w3.J, 128));
  const int n3 = V3 == 0 ? nb3 : nb3 * S3;
  const int cls = NFK_PROF_TRIPLE_WGRAD + (w2.M >= 8192 ? 0 : (w2.M >= 2048 ? 1 : 2));
  // algorithmic work (bench.py roofline leg): `alg_nj` = sum over the three convs of Cout * Cin * taps when the operands
  // carry padding (im2col rows), else from the arguments
  const double nj = alg_nj > 0 ? alg_nj : ((double)w2.NI * w2.J + (double)w1.NI * w1.J + (double)w3.NI * w3.J);
  prof_scope_begin(cls, 2.0 * w2.M * nj,
                   w2.M * (2.0 * w2.NI + 2.0 * w2.Cq + 2.0 * w1.NI + 4.0 * w1.Cq + 4.0 * w3.NI + 2.0 * w3.Cq) +
                       4.0 * (S2 * (double)w2.NI * w2.J + S1 * (double)w1.NI * w1.J + S3 * (double)w3.NI * w3.J), st);
  // diagnostic (nfk_debug_triple_parts): leave parts of the launch out to time the others (bit 0: ActNorm sums, 1: conv2,
  // 2: conv3, 3: conv1); results are then wrong
  ActnormSumArgs anx = an;
  if (g_triple_skip & 1) anx.nparts = 0;
  const int S2x = (g_triple_skip & 2) ? 0 : S2, n3x = (g_triple_skip & 4) ? 0 : n3, S1x = (g_triple_skip & 8) ? 0 : S1;
  if (nb2 * S2x + nb1 * S1x + n3x + anx.nparts == 0) return NFK_OK;
  hipLaunchKernelGGL((triple_wgrad_bf16_kernel<V3>), dim3(nb2 * S2x + nb1 * S1x + n3x + anx.nparts), dim3(256), lds, st, w2,
                     nb2, S2x, w1, nb1, S1x, w3, nb3, S3, n3x, anx);
  prof_scope_end(cls, st);
  NFK_LAUNCH_CHECK();
  return NFK_OK;
}

int triple_wgrad_bf16(WgradArgs w3, int S3, WgradArgs w2, int S2, WgradArgs w1, int S1, const ActnormSumArgs& an,
                      hipStream_t st, double alg_nj) {
  for (WgradArgs* w : {&w3, &w2, &w1}) {
    w->M = w->B * w->H * w->W;
    w->J = w->taps * w->Cq;
    if (w->M <= 0 || w->NI <= 0 || w->J <= 0) return NFK_ERR_ARG;
    if ((long)w->M * w->ldp * 4 >= 0x7fffffffL || (long)w->M * w->ldq * 4 >= 0x7fffffffL) return NFK_ERR_ARG;
  }
  if (w2.NI <= 96 || w1.NI <= 96 || !w2.pbf16 || !w2.qbf16 || !w1.pbf16) return NFK_ERR_ARG;
  if (w3.taps == 1 && w3.pbf16) {   // conv3 through dgrad3's im2col rows (taps folded into P): a plain GEMM like the other two
    if (!w3.qbf16 || ((w3.NI | w3.ldp | w3.Cq | w3.ldq) & 7)) return NFK_ERR_ARG;
    return launch_triple<4>(w3, S3, w2, S2, w1, S1, an, st, alg_nj);
  }
  if (w3.NI > 48) return NFK_ERR_ARG;
  if (!w3.pbf16 && w3.qbf16 && wgrad_halo_usable(w3.M, w3.NI, w3.Cq, w3.taps, w3.W) && S3 == wgrad_halo_splits(w3.M))
    return launch_triple<0>(w3, S3, w2, S2, w1, S1, an, st, alg_nj);
  switch (ceil_div(w3.NI, 16)) {
    case 1: return launch_triple<1>(w3, S3, w2, S2, w1, S1, an, st, alg_nj);
    case 2: return launch_triple<2>(w3, S3, w2, S2, w1, S1, an, st, alg_nj);
    case 3: return launch_triple<3>(w3, S3, w2, S2, w1, S1, an, st, alg_nj);
  }
  return NFK_ERR_ARG;
}

template <int TN, int TM, int WN, int WM>
static int launch_wgrad_bf16(const WgradArgs& a, int S, hipStream_t st, int cls) {
  constexpr int BI = 16 * TN * WN, BJ = 16 * TM * WM;
  constexpr size_t lds = 2 * WPX * (wg_stride(BI) + wg_stride(BJ));
  static bool attr = false;
  if (!attr && lds > 48 * 1024) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_bf16_kernel<TN, TM, WN, WM>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr = true;
  }
  dim3 grid(ceil_div(a.NI, BI) * ceil_div(a.J, BJ), S);
  prof_scope_begin(cls, 2.0 * a.M * (double)a.NI * a.J,
                   (a.pbf16 ? 2.0 : 4.0) * a.M * a.NI + (a.qbf16 ? 2.0 : 4.0) * a.M * a.Cq + 4.0 * S * a.NI * a.J, st);
  hipLaunchKernelGGL((wgrad_bf16_kernel<TN, TM, WN, WM>), grid, dim3(256), lds, st, a);
  prof_scope_end(cls, st);
  NFK_LAUNCH_CHECK();
  return NFK_OK;
}

// split count of the conv3-type weight gradient of a step (NI = Cout, J = 9 * Cq): the halo-tile kernel's when it applies
int wgrad3_splits_bf16(int M, int NI, int Cq, int W) {
  return wgrad_halo_usable(M, NI, Cq, 9, W) ? wgrad_halo_splits(M) : wgrad_splits_bf16(M, NI, 9 * Cq);
}

int wgrad_splits_bf16(int M, int NI, int J) {
  const int tiles = (NI > 96 ? ceil_div(NI, 128) : 1) * ceil_div(J, 128);
  const int nchunks = ceil_div(M, WPX);
  int S = ceil_div(g_wgrad_target_wgs, tiles);
  if (S > nchunks / 2) S = nchunks / 2;
  if (S < 1) S = 1;
  if (S > g_wgrad_max_splits) S = g_wgrad_max_splits;
  if (NI > 96 && J > 96 && S > g_wgrad_wide_max_splits) S = g_wgrad_wide_max_splits;   // 1 MiB per split for 512x512
  return S;
}

int wgrad_gemm_bf16(WgradArgs a, int S, hipStream_t st) {
  a.M = a.B * a.H * a.W;
  a.J = a.taps * a.Cq;
  if (a.M <= 0 || a.NI <= 0 || a.J <= 0) return NFK_OK;
  if ((long)a.M * a.ldp * 4 >= 0x7fffffffL || (long)a.M * a.ldq * 4 >= 0x7fffffffL) return NFK_ERR_ARG;
  if (a.NI > 96) return launch_wgrad_bf16<4, 4, 2, 2>(a, S, st, NFK_PROF_WGRAD_WIDE);
  switch (ceil_div(a.NI, 16)) {
    case 1: return launch_wgrad_bf16<1, 2, 1, 4>(a, S, st, NFK_PROF_WGRAD_NARROW);
    case 2: return launch_wgrad_bf16<2, 2, 1, 4>(a, S, st, NFK_PROF_WGRAD_NARROW);
    case 3: return launch_wgrad_bf16<3, 2, 1, 4>(a, S, st, NFK_PROF_WGRAD_NARROW);
    case 4: return launch_wgrad_bf16<4, 2, 1, 4>(a, S, st, NFK_PROF_WGRAD_NARROW);
    default: return launch_wgrad_bf16<6, 2, 1, 4>(a, S, st, NFK_PROF_WGRAD_NARROW);
  }
}

template <int CTN, int CTM, int CWN, int CWM>
static int launch_dual_wgradhalo(const ConvGemmArgs& d, int SD, const WgradArgs& w, int SW, hipStream_t st) {
  constexpr int BN = 16 * CTN * CWN, BM = 16 * CTM * CWM;
  const size_t ldsD = 2 * (BN + BM) * 128, ldsW = wh_lds_bytes(w.W);
  const size_t lds = ldsD > ldsW ? ldsD : ldsW;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dual_dgrad_wgradhalo_bf16_kernel<CTN, CTM, CWN, CWM, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    attr = true;
  }
  const int nbD = ceil_div(d.N, BN) * ceil_div(d.M, BM), nbW = (w.Cq / 128) * SW;
  const int cls = NFK_PROF_DUAL_BWD + (d.M >= 8192 ? 0 : (d.M >= 2048 ? 1 : 2));
  prof_scope_begin(cls, 2.0 * d.M * (double)d.N * d.K + 2.0 * w.M * (double)w.NI * w.J,
                   4.0 * d.M * d.Cin + 2.0 * d.N * d.K + (d.obf16 ? 2.0 : 4.0 * SD) * d.M * d.N + 4.0 * w.M * w.NI +
                       2.0 * w.M * w.Cq + 4.0 * SW * w.NI * w.J, st);
  hipLaunchKernelGGL((dual_dgrad_wgradhalo_bf16_kernel<CTN, CTM, CWN, CWM, true>), dim3(nbD * SD + nbW), dim3(256), lds, st,
                     d, nbD, SD, w, SW);
  prof_scope_end(cls, st);
  NFK_LAUNCH_CHECK();
  return NFK_OK;
}

template <int NT>
static int launch_dual_halo(const ConvGemmArgs& d, const WgradArgs& w, int SW, hipStream_t st) {
  constexpr int BI = 128, BJ = 128;   // the wide weight gradient: wgrad_bf16_body<4, 4, 2, 2>
  const size_t ldsD = nh_lds_bytes(d.W, NT), ldsW = 2 * WPX * (wg_stride(BI) + wg_stride(BJ));
  const size_t lds = ldsD > ldsW ? ldsD : ldsW;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dual_halo_dgrad_wgrad_bf16_kernel<NT, 4, 4, 2, 2>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)NH_MAX_LDS);
    attr = true;
  }
  const int nbD = ceil_div(d.M, 64), nbW = ceil_div(w.NI, BI) * ceil_div(w.J, BJ);
  const int cls = NFK_PROF_DUAL_BWD + 3 * 2 + (d.M >= 8192 ? 0 : (d.M >= 2048 ? 1 : 2));
  prof_scope_begin(cls, 2.0 * d.M * (double)d.N * d.K + 2.0 * w.M * (double)w.NI * w.J,
                   2.0 * d.M * d.Cin + 2.0 * d.N * d.K + 4.0 * d.M * d.N + (w.pbf16 ? 2.0 : 4.0) * w.M * w.NI +
                       (w.qbf16 ? 2.0 : 4.0) * w.M * w.Cq + 4.0 * SW * w.NI * w.J, st);
  hipLaunchKernelGGL((dual_halo_dgrad_wgrad_bf16_kernel<NT, 4, 4, 2, 2>), dim3(nbD + nbW * SW), dim3(256), lds, st, d, nbD,
                     w, nbW, SW);
  prof_scope_end(cls, st);
  NFK_LAUNCH_CHECK();
  return NFK_OK;
}

// dgrad + wgrad of one conv in ONE launch.  Returns NFK_ERR_ARG when the pair of tile configurations the two
// single launchers would pick is not instantiated (the caller then launches them separately).
int dual_dgrad_wgrad_bf16(ConvGemmArgs d, WgradArgs w, int SW, hipStream_t st) {
  d.M = d.B * d.H * d.W;
  d.K = d.taps * d.Cin;
  d.nkchunks = ceil_div(d.K, BKH);
  w.M = w.B * w.H * w.W;
  w.J = w.taps * w.Cq;
  if (d.M <= 0 || d.N <= 0 || d.K <= 0 || w.NI <= 0 || w.J <= 0) return NFK_ERR_ARG;
  if ((long)d.M * d.ldx * 4 >= 0x7fffffffL || (long)d.N * d.Kp * 2 >= 0x7fffffffL) return NFK_ERR_ARG;
  if ((long)w.M * w.ldp * 4 >= 0x7fffffffL || (long)w.M * w.ldq * 4 >= 0x7fffffffL) return NFK_ERR_ARG;
  const int SD = conv_splitk_bf16(d.M, d.N, d.K, d.epi);
  const bool wideD = d.N > 96, bigD = wideD && big_tiles(d.M, d.N, d.K), wideW = w.NI > 96;
  const int ntD = ceil_div(d.N, 16), ntW = ceil_div(w.NI, 16);
  if (wideD && !wideW) {   // conv3: wide dgrad (fp32 gradient in), narrow weight gradient
    if (!d.xf32) return NFK_ERR_ARG;
    if (!w.pbf16 && w.qbf16 && wgrad_halo_usable(w.M, w.NI, w.Cq, w.taps, w.W) && SW == wgrad_halo_splits(w.M))
      return bigD ? launch_dual_wgradhalo<4, 4, 2, 2>(d, SD, w, SW, st) : launch_dual_wgradhalo<2, 2, 2, 2>(d, SD, w, SW, st);
    if (bigD) {
      if (ntW == 1) return launch_dual<4, 4, 2, 2, true, 1, 2, 1, 4>(d, SD, w, SW, st);
      if (ntW == 2) return launch_dual<4, 4, 2, 2, true, 2, 2, 1, 4>(d, SD, w, SW, st);
      if (ntW == 3) return launch_dual<4, 4, 2, 2, true, 3, 2, 1, 4>(d, SD, w, SW, st);
    } else {
      if (ntW == 1) return launch_dual<2, 2, 2, 2, true, 1, 2, 1, 4>(d, SD, w, SW, st);
      if (ntW == 2) return launch_dual<2, 2, 2, 2, true, 2, 2, 1, 4>(d, SD, w, SW, st);
      if (ntW == 3) return launch_dual<2, 2, 2, 2, true, 3, 2, 1, 4>(d, SD, w, SW, st);
    }
    return NFK_ERR_ARG;
  }
  if (wideD && wideW) {    // conv2: bf16 in, both wide
    if (d.xf32) return NFK_ERR_ARG;
    return bigD ? launch_dual<4, 4, 2, 2, false, 4, 4, 2, 2>(d, SD, w, SW, st)
                : launch_dual<2, 2, 2, 2, false, 4, 4, 2, 2>(d, SD, w, SW, st);
  }
  if (!wideD && wideW) {   // conv1: narrow dgrad (split-K slabs), wide weight gradient
    if (d.xf32) return NFK_ERR_ARG;
    if (g_narrow_halo && d.epi == EPI_RAW && narrow_halo_usable(1, d.M, d.H, d.W, d.N, d.Cin, d.taps)) {
      if (ntD == 1) return launch_dual_halo<1>(d, w, SW, st);
      if (ntD == 2) return launch_dual_halo<2>(d, w, SW, st);
      if (ntD == 3) return launch_dual_halo<3>(d, w, SW, st);
    }
    if (ntD == 1) return launch_dual<1, 2, 1, 4, false, 4, 4, 2, 2>(d, SD, w, SW, st);
    if (ntD == 2) return launch_dual<2, 2, 1, 4, false, 4, 4, 2, 2>(d, SD, w, SW, st);
    if (ntD == 3) return launch_dual<3, 2, 1, 4, false, 4, 4, 2, 2>(d, SD, w, SW, st);
    return NFK_ERR_ARG;
  }
  return NFK_ERR_ARG;
}

void set_tuning_bf16(int v) {
  if (v >= 3000) g_wgrad_halo_max_splits = v - 3000;
  else if (v >= 2910) g_wgrad_halo = (v - 2910) & 1;
  else if (v >= 2900) g_narrow_halo = (v - 2900) & 1;
  else if (v >= 2000) g_wgrad_wide_max_splits = v - 2000;
  else if (v >= 900) g_wgrad_target_wgs = v - 900;
  else if (v >= 100) g_wgrad_max_splits = v - 100;
  else g_big_tile_min_k = v * 64;
}

// does the backward of a step with these dims run conv1's dgrad through the halo-tile kernel (one output slab)?
bool narrow_halo_dgrad_active(int bf16, int M, int H, int W, int N, int Cin) {
  return g_narrow_halo && narrow_halo_usable(bf16, M, H, W, N, Cin, 9) && N <= 48;
}

}  // namespace nfk
