// Device bodies shared by the launches that end a FlowStep's backward (nfk_flow.hip: tail_chanmix_kernel, step_tail_kernel)
// and by the launch that merges them into the step's weight-gradient launch (nfk_gemm_bf16.hip: wgrad_tail_kernel):
// channel-mix backward (+ the next step's coupling backward), partial-row sums, split-slab reductions.
#pragma once
#include "nfk_internal.h"

namespace nfk {

// ==========================================================================================
// chanmix:  z[m][o] = sum_c Wm[o][c] x[m][c] + cv[o]      (trans: uses Wm[c][o], no cv)
// ==========================================================================================
constexpr int CM_MAX_BLOCKS = 256;  // backward: at most this many partial (gWf, gcf) slabs per step

// pixels per tile: TP * ceil(C/4) output-groups ~ 256 threads
static inline __host__ __device__ int cm_tile_pixels(int C) {
  const int g4 = (C + 3) / 4;
  int tp = 64;
  while (tp > 1 && tp * g4 > 256) tp >>= 1;
  return tp;
}

// `nx` (optional, nx.ga3 != null): the coupling backward of the step that FOLLOWS in backward order, whose output gradient
// is exactly the g_x this launch produces -- done here per pixel tile instead of in a launch of its own (the tail /
// channel-mix launch of a step is followed by nothing else).  Its per-column sums become one partial row per
// workgroup (instead of one per sample), reduced by the same small-gradient code later.
// (FUSE is a template parameter and `n` a reference to the kernel's own argument: through a pointer to it the argument
// struct was copied to scratch memory and read back from there)
template <bool FUSE>
__device__ __forceinline__ void chanmix_body(const ChanMixArgs& a, const int bx, const int nblk, const CouplingBwdArgs& n) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int C = a.C, Cp = (C + 3) & ~3, XS = C + 1, G4 = Cp / 4;
  const int TP = cm_tile_pixels(C);
  float* Wl = smem;                 // [C][Cp]  stored [in][out]
  float* Xl = Wl + C * Cp;          // [TP][C+1]
  float* Gl = Xl + TP * XS;         // [TP][C+1] (backward: x_in tile)
  float* Pl = Gl + TP * XS;         // [C*C + C] (backward: partial sums, each element owned by one thread)
  const int E = C * C + C;
  constexpr bool fuse = FUSE;
  const int Chn = C / 2, C2n = C - Chn, Gn = 256 / (C2n > 0 ? C2n : 1);
  float* Ol = Pl + (a.part ? E : 0);      // [TP][C+1] this launch's g_x tile      (fused coupling only)
  float* Al = Ol + TP * XS;               // [Gn][4][C2] per-group column sums     (fused coupling only)
  float* Pc = Al + Gn * 4 * C2n;          // [4][C2] running column sums of this workgroup
  if (fuse)
    for (int idx = threadIdx.x; idx < 4 * C2n; idx += blockDim.x) Pc[idx] = 0.f;
  const int ntiles = (a.M + TP - 1) / TP;
  // The workgroup is a chain of memory round trips on very little data (W, the pixel tile, the fused coupling's
  // operands, the stores): the tile's elements (TP * C <= 1024: at most four per thread) are requested before the
  // wait for W, and the coupling's operands (below) at the top of the tile instead of after the channel mix.
  constexpr int XPF = 4;
  float xv[XPF], gv[XPF];   // the tile about to be processed: loaded one tile ahead
  auto load_x = [&](int m0) {
#pragma unroll
    for (int e = 0; e < XPF; ++e) {
      const int idx = threadIdx.x + e * 256;
      const int pp = idx / C, c = idx - pp * C, m = m0 + pp;
      float v = 0.f, gq = 0.f;
      // (32-bit element offsets from the arrays' base pointers: one address register per load instead of two; the
      // tensors of a FlowStep are far below 2^32 elements)
      if (idx < TP * C && m < a.M) {
        if (a.ga && c < a.Ch) {
          v = a.ga[(unsigned)(m * a.lda + c)];
          if (a.gext) v += a.gext[(unsigned)(m * a.ldge + c)];
          // conv1's dgrad slabs (one with the halo-tile kernel, a few otherwise), four loads in flight per element:
          // four elements are loaded side by side here, and slab_sum's sixteen each cost the launch its occupancy
          const float* sp = a.slab + (unsigned)(m * a.Ch + c);
          int q = 0;
          for (; q + 4 <= a.S; q += 4, sp += 4 * a.slab_stride)
            v += (sp[0] + sp[a.slab_stride]) + (sp[2 * a.slab_stride] + sp[3 * a.slab_stride]);
          for (; q < a.S; ++q, sp += a.slab_stride) v += sp[0];
        } else {
          v = a.x[(unsigned)(m * a.ldx + c)];
        }
        if (a.part) gq = a.xin[(unsigned)(m * a.ldxin + c)];
      }
      xv[e] = v; gv[e] = gq;
    }
  };
  load_x(bx * TP);   // (a workgroup beyond the last tile loads nothing: every element fails m < M)
  for (int idx = threadIdx.x; idx < C * C; idx += blockDim.x) {
    const int r = idx / C, q = idx - r * C;  // Wm[r][q]
    if (a.trans) Wl[r * Cp + q] = a.Wm[idx];      // in = r (rows of Wm), out = q
    else Wl[q * Cp + r] = a.Wm[idx];               // in = q, out = r
  }
  if (a.part)
    for (int idx = threadIdx.x; idx < E; idx += blockDim.x) Pl[idx] = 0.f;
  const int p = threadIdx.x / G4, og = threadIdx.x - p * G4;
  // fused coupling: thread (g, j) = column j of pixels g, g + Gn, ... of the tile (at most CPF of them)
  constexpr int CPF = 4;
  const int jn = threadIdx.x % C2n, gn = threadIdx.x / C2n;
  const bool cpl_pre = fuse && (TP + Gn - 1) / Gn <= CPF;
  for (int tile = bx; tile < ntiles; tile += nblk) {
    const int m0 = tile * TP;
    float ch0[CPF], ch1[CPF], cz[CPF], cg[CPF];   // next step's h (shift | scale), z2, external gradient
    if (cpl_pre && gn < Gn) {
#pragma unroll
      for (int e = 0; e < CPF; ++e) {
        const int pp = gn + e * Gn, m = m0 + pp;
        const bool ok = pp < TP && m < a.M;
        const unsigned mc = ok ? (unsigned)m : 0u;
        cg[e] = (ok && n.gext) ? n.gext[mc * n.ldge + Chn + jn] : 0.f;
        if (n.additive) { ch0[e] = n.h[mc * n.Cout + jn]; ch1[e] = 0.f; cz[e] = 0.f; }
        else { ch0[e] = n.h[mc * n.Cout + 2 * jn]; ch1[e] = n.h[mc * n.Cout + 2 * jn + 1]; cz[e] = n.z[mc * n.ldz + Chn + jn]; }
      }
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < XPF; ++e) {
      const int idx = threadIdx.x + e * 256;
      if (idx < TP * C) {
        const int pp = idx / C, c = idx - pp * C;
        Xl[pp * XS + c] = xv[e];
        if (a.part) Gl[pp * XS + c] = gv[e];
      }
    }
    if (tile + nblk < ntiles) load_x((tile + nblk) * TP);   // next tile, under this tile's work
    __syncthreads();
    if (p < TP) {
      const int o = og * 4, m = m0 + p;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      if (a.cv)
#pragma unroll
        for (int r = 0; r < 4; ++r) if (o + r < C) acc[r] = a.cv[o + r];
      const float* xr = Xl + p * XS;
      for (int c = 0; c < C; ++c) {
        const float xv = xr[c];
        const f32x4 w = *reinterpret_cast<const f32x4*>(Wl + c * Cp + o);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] += w[r] * xv;
      }
      if (m < a.M) {
        float* zp = a.z + (long)m * a.ldz + o;
        if (o + 3 < C && (a.ldz & 3) == 0) *reinterpret_cast<f32x4*>(zp) = acc;
        else
#pragma unroll
          for (int r = 0; r < 4; ++r) if (o + r < C) zp[r] = acc[r];
      }
      if (fuse)
#pragma unroll
        for (int r = 0; r < 4; ++r) if (o + r < C) Ol[p * XS + o + r] = acc[r];
    }
    if (a.part) {
      for (int idx = threadIdx.x; idx < E; idx += blockDim.x) {
        float s = 0.f;
        if (idx < C * C) {
          const int o = idx / C, c = idx - o * C;
          for (int q = 0; q < TP; ++q) s += Xl[q * XS + o] * Gl[q * XS + c];
        } else {
          const int o = idx - C * C;
          for (int q = 0; q < TP; ++q) s += Xl[q * XS + o];
        }
        Pl[idx] += s;
      }
    }
    if (fuse) {
      // coupling backward of the next step on this tile (same arithmetic as coupling_bwd_kernel)
      __syncthreads();
      const int j = threadIdx.x % C2n, g = threadIdx.x / C2n;
      float sb0 = 0.f, sl0 = 0.f, sb1 = 0.f, sl1 = 0.f;
      if (g < Gn) {
        // one pixel of this thread's column: operands either preloaded at the top of the tile or read here
        auto elem = [&](int pp, unsigned m, float hv0, float hv1, float zv, float gv) {
          float g2 = Ol[pp * XS + Chn + j];
          if (n.gext) g2 += gv;
          if (n.additive) {
            n.gz[m * n.ldgz + Chn + j] = g2;
            const float ga = g2 * n.es3[j];
            n.ga3[m * n.Cout + j] = ga;
            sb0 += ga; sl0 += g2 * hv0;
          } else {
            const float gl = n.gld ? n.gld[m / n.HW] : 0.f;
            const float shf = hv0, sr = hv1;
            const float sc = sigmoidf(sr + 2.f);
            const float gshift = g2 * sc;
            const float gsr = g2 * (zv + shf) * sc * (1.f - sc) + gl * (1.f - sc);
            n.gz[m * n.ldgz + Chn + j] = gshift;
            const float ga0 = gshift * n.es3[2 * j], ga1 = gsr * n.es3[2 * j + 1];
            n.ga3[m * n.Cout + 2 * j] = ga0;
            n.ga3[m * n.Cout + 2 * j + 1] = ga1;
            sb0 += ga0; sl0 += gshift * shf; sb1 += ga1; sl1 += gsr * sr;
          }
        };
        if (cpl_pre) {
#pragma unroll
          for (int e = 0; e < CPF; ++e) {
            const int pp = g + e * Gn, m = m0 + pp;
            if (pp < TP && m < a.M) elem(pp, (unsigned)m, ch0[e], ch1[e], cz[e], cg[e]);
          }
        } else {
          for (int pp = g; pp < TP; pp += Gn) {
            const unsigned m = (unsigned)(m0 + pp);
            if ((int)m >= a.M) break;
            const float gv = n.gext ? n.gext[m * n.ldge + Chn + j] : 0.f;
            if (n.additive) elem(pp, m, n.h[m * n.Cout + j], 0.f, 0.f, gv);
            else elem(pp, m, n.h[m * n.Cout + 2 * j], n.h[m * n.Cout + 2 * j + 1], n.z[m * n.ldz + Chn + j], gv);
          }
        }
        float* row = Al + (long)g * 4 * C2n;
        row[j] = sb0; row[C2n + j] = sl0; row[2 * C2n + j] = sb1; row[3 * C2n + j] = sl1;
      }
      __syncthreads();
      for (int jj = threadIdx.x; jj < 4 * C2n; jj += blockDim.x) {
        float t = 0.f;
        for (int gg = 0; gg < Gn; ++gg) t += Al[(long)gg * 4 * C2n + jj];
        Pc[jj] += t;
      }
    }
  }
  if (a.part) {
    float* part = a.part + (long)bx * E;
    for (int idx = threadIdx.x; idx < E; idx += blockDim.x) part[idx] = Pl[idx];
  }
  if (fuse) {
    __syncthreads();
    float* part = n.part + (long)bx * 2 * n.Cout;   // one partial row per workgroup
    for (int jj = threadIdx.x; jj < C2n; jj += blockDim.x) {
      const float t0 = Pc[jj], t1 = Pc[C2n + jj], t2 = Pc[2 * C2n + jj], t3 = Pc[3 * C2n + jj];
      if (n.additive) { part[jj] = t0; part[n.Cout + jj] = 3.f * t1; }
      else {
        part[2 * jj] = t0; part[2 * jj + 1] = t2;
        part[n.Cout + 2 * jj] = 3.f * t1; part[n.Cout + 2 * jj + 1] = 3.f * t3;
      }
    }
  }
}

// Column `col` of the partial rows rg, rg + 8, ... of a [rows][2][n] array, summed in that order.  Eight rows' loads
// are issued before the first add: a plain loop waits for every pair of loads in turn, and with up to 256 partial rows
// (chained backward: one per channel-mix workgroup) that was a chain of 32 memory latencies.
__device__ __forceinline__ void partial_rows_sum(const float* __restrict__ src, int rows, int n, int col, int rg, float& s0,
                                                 float& s1) {
  int i = rg;
  for (; i + 56 < rows; i += 64) {
    float u[8], v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      u[e] = src[((long)(i + 8 * e) * 2) * n + col];
      v[e] = src[((long)(i + 8 * e) * 2 + 1) * n + col];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { s0 += u[e]; s1 += v[e]; }
  }
  for (; i < rows; i += 8) {
    s0 += src[((long)i * 2) * n + col];
    s1 += src[((long)i * 2 + 1) * n + col];
  }
}


// One launch for everything a FlowStep's backward still owes its parameters once the GEMMs are done: the three
// split-slab reductions of the weight gradients (wgrad finalize: gradW[co][ci][tap] += sum_s slab[s][co][tap*Cin+ci])
// and the small per-channel gradients above.  blockIdx.x ranges select the task.
// (`ov`: level mode -- the three slab bases and gradient pointers of THIS step, the rest of `f` being the same for every step
//  of the level: passed beside it, because a modified private copy of the argument struct would live in scratch memory)
struct StepTailOverride { const float* slabs[3]; float* gradW[3]; };
template <bool OV = false>
__device__ __forceinline__ void step_tail_body(const SmallGradArgs& a, const StepTailArgs& f, const int blk,
                                               const float* s0 = nullptr, const float* s1 = nullptr, const float* s2 = nullptr,
                                               float* w0 = nullptr, float* w1 = nullptr, float* w2 = nullptr) {
  const int nsg = f.sg_blocks * 3;
  if (blk < nsg) {
    __shared__ float red[2][8][32];
    const int seg = blk / f.sg_blocks, bx = blk - seg * f.sg_blocks;
    const float* src; int rows, n; float *g0, *g1;
    if (seg == 0) { src = a.col1; rows = a.mb1; n = a.n1; g0 = a.g_l1; g1 = a.g_b1; }
    else if (seg == 1) { src = a.col2; rows = a.mb2; n = a.n2; g0 = a.g_l2; g1 = a.g_b2; }
    else { src = a.tail; rows = a.B; n = a.n3; g0 = a.g_bias3; g1 = a.g_l3; }
    const int col = bx * 32 + (threadIdx.x & 31), rg = threadIdx.x >> 5;
    if (bx * 32 >= n) return;
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
    return;
  }
  int b = blk - nsg, k = 0;
  while (k < 2 && b >= f.blocks[k]) { b -= f.blocks[k]; ++k; }
  const int S = f.S[k], Cin = f.Cin[k], taps = f.taps[k], J = Cin * taps;
  const int total = f.Cout[k] * J;
  const float* slabs = OV ? (k == 0 ? s0 : (k == 1 ? s1 : s2)) : f.slabs[k];
  float* gw = OV ? (k == 0 ? w0 : (k == 1 ? w1 : w2)) : f.gradW[k];
  const int ldj = f.ldj[k] ? f.ldj[k] : J, mode = f.mode[k];
  for (int t = b * 256 + threadIdx.x; t < total; t += f.blocks[k] * 256) {
    const int r = t / taps, tap = t - r * taps;
    const int co = r / Cin, ci = r - co * Cin;
    const long off = mode == 1 ? (long)((taps - 1 - tap) * f.Cout[k] + co) * ldj + ci : (long)co * ldj + tap * Cin + ci;
    gw[t] += slab_sum(slabs, S, f.stride[k], off);
  }
}


// workgroups of a step tail (filled into f.blocks / f.sg_blocks): 3 * sg_blocks small-gradient blocks, then the slab blocks
static inline int tail_blocks(const SmallGradArgs& a, StepTailArgs& f) {
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
  return total;
}

}  // namespace nfk
