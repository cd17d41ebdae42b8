// nfk fused Split2d (models/layers.py:293-313): the Conv2dZeros prior on the dropped half and its Gaussian log-density in
// ONE launch per direction, in fp32 on the vector ALUs.
//
//   forward   z1, z2 = split(x);  (mean, logs) = cross(Conv2dZeros_3x3(z1));  logdet += sum log N(z2; mean, e^logs);  out = z1
//   backward  g_x = (g_z1 + dgrad(g_a), -g_mean),  partial weight gradient and per-channel sums per workgroup
//
// The convolution is tiny (C/2 -> C channels: 6 -> 12 at CIFAR level 1, 12 -> 24 at level 2; 21 / 84 MFLOP per batch of 64)
// and sits directly on the log-likelihood, so it stays exact fp32.  Through the generic MFMA GEMMs it cost six launches in
// the backward (coupling-style elementwise kernel, split-K weight gradient, dgrad, slab add, two reductions) and two in
// the forward, 160 us per training step of pure launch latency on the student's chain.  Here a workgroup owns 64 pixels
// (whole image rows + one halo row each side, or whole small images), keeps the x / (mean, logs) / g_a tiles and the 3x3
// weights in LDS, and does the convolution, its transpose and the weight-gradient outer products from there.
// Per-sample log-density sums and all gradient reductions are fixed-order (per-tile partials, reduced by nfk_ld_finish /
// the step-tail kernel): bitwise reproducible.
#include "nfk_internal.h"

namespace nfk {

constexpr float SP_LN_2PI = 1.8378770664093453f;

struct SplitGeom {
  int NPX, R, TS, halo, RS, lw, tps, ntiles, npix_s;   // pixels per tile, rows per tile, samples per tile, halo rows, staged
};                                                     // rows per sample, log2 W, tiles per sample, tiles, staged pixels

static int sp_ilog2(int v) {
  if (v <= 0 || (v & (v - 1))) return -1;
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}

static bool sp_geom(const nfk_split2d& s, SplitGeom* g) {
  const int lw = sp_ilog2(s.W), lh = sp_ilog2(s.H), HW = s.H * s.W;
  if (lw < 0 || lh < 0 || s.W > 64 || (s.C & 3) || s.C < 4 || s.C > 96) return false;
  // 64-pixel tiles, halved while the launch would leave CUs idle (the kernels are latency-bound: more, smaller workgroups)
  for (int npx = 64;; npx >>= 1) {
    g->NPX = npx;
    if (HW <= npx) { g->R = s.H; g->TS = npx / HW; g->halo = 0; }
    else { g->R = npx / s.W; g->TS = 1; g->halo = 1; }
    g->RS = g->R + 2 * g->halo;
    g->lw = lw;
    g->tps = g->TS == 1 ? s.H / g->R : 1;
    g->ntiles = g->TS == 1 ? s.B * g->tps : (s.B + g->TS - 1) / g->TS;
    g->npix_s = g->TS * g->RS * s.W;   // staged pixels (incl. halo rows)
    if (g->ntiles >= 256 || npx / 2 < 16 || npx / 2 < s.W) break;
  }
  return true;
}

struct SplitFusedFwd {
  nfk_split2d s;
  SplitGeom g;
  const float* x;       // [M][C]
  const float* w;       // conv.weight [C][C/2][3][3]
  float* z1out;         // [M][C/2]
  float* hsave;         // [M][C] or null
  float* ldpart;        // [B][tps]
};

// LDS: Zt [npix_s][Ch] | Wl [9*Ch][Cp] | Hl [NPX][C] | LS [NPX * Ch]
__global__ void __launch_bounds__(256) split2d_fwd_fused_kernel(SplitFusedFwd a) {
  extern __shared__ __attribute__((aligned(16))) float smem_sp[];
  const nfk_split2d& s = a.s;
  const SplitGeom& g = a.g;
  const int C = s.C, Ch = C / 2, W = s.W, H = s.H, HW = H * W, lw = g.lw, Cp = (C + 3) & ~3;
  const int R = g.R, TS = g.TS, halo = g.halo, RS = g.RS, NPX = g.NPX;
  float* Zt = smem_sp;
  float* Wl = Zt + ((g.npix_s * Ch + 3) & ~3);
  float* Hl = Wl + 9 * Ch * Cp;
  float* LS = Hl + NPX * C;
  const int tid = threadIdx.x, wg = blockIdx.x;
  const int b0 = TS == 1 ? wg / g.tps : wg * TS, tile = TS == 1 ? wg - b0 * g.tps : 0, row0 = tile * R;
  // stage z1 (+ halo rows; zeros outside the image / the batch) and the weights as Wl[tap * Ch + ci][co]
  // (four loads in flight per thread: one workgroup per CU is all the launch has, nothing else hides the latency)
  for (int base = tid; base < g.npix_s * Ch; base += 1024) {
    float v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = base + u * 256;
      const int ps = idx / Ch, ci = idx - ps * Ch;
      const int sm = ps / (RS * W), rem = ps - sm * RS * W, r = rem >> lw, xx = rem & (W - 1);
      const int y = row0 + r - halo, b = b0 + sm;
      v[u] = (idx < g.npix_s * Ch && (unsigned)y < (unsigned)H && b < s.B) ? a.x[((long)b * HW + y * W + xx) * C + ci] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (base + u * 256 < g.npix_s * Ch) Zt[base + u * 256] = v[u];
  }
  for (int base = tid; base < C * Ch * 9; base += 1024) {
    float v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = base + u * 256 < C * Ch * 9 ? a.w[base + u * 256] : 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = base + u * 256;
      const int co = idx / (Ch * 9), r = idx - co * Ch * 9, ci = r / 9, tap = r - ci * 9;
      if (idx < C * Ch * 9) Wl[(tap * Ch + ci) * Cp + co] = v[u];
    }
  }
  __syncthreads();
  // convolution: thread item = (output pixel, 4 output channels); h = (acc + bias) * e^{3 logs}
  const int G4 = Cp / 4;
  for (int t = tid; t < NPX * G4; t += 256) {
    const int p = t / G4, o = (t - p * G4) * 4;
    const int sm = p / (R * W), rem = p - sm * R * W, rr = rem >> lw, xx = rem & (W - 1);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int dy = tap / 3 - 1, dx = tap % 3 - 1;
      const int rs = rr + halo + dy, xn = xx + dx;
      if ((unsigned)rs >= (unsigned)RS || (unsigned)xn >= (unsigned)W) continue;   // (rows outside the image are zero rows)
      const float* zr = Zt + ((sm * RS + rs) * W + xn) * Ch;
      const float* wr = Wl + tap * Ch * Cp + o;
#pragma unroll 2
      for (int ci = 0; ci < Ch; ci += 2) {
        const float z0 = zr[ci], z1 = zr[ci + 1];
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(wr + ci * Cp), w1 = *reinterpret_cast<const f32x4*>(wr + (ci + 1) * Cp);
        acc += w0 * z0;
        acc += w1 * z1;
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (o + r < C) Hl[p * C + o + r] = (acc[r] + s.bias[o + r]) * s.es[o + r];
  }
  __syncthreads();
  // log-density of z2 under N(mean, e^logs), z1 out, (mean, logs) kept for the backward
  for (int idx = tid; idx < NPX * Ch; idx += 256) {
    const int p = idx / Ch, j = idx - p * Ch;
    const int sm = p / (R * W), rem = p - sm * R * W, rr = rem >> lw, xx = rem & (W - 1);
    const int b = b0 + sm;
    float lp = 0.f;
    if (b < s.B) {
      const long m = (long)b * HW + (row0 + rr) * W + xx;
      const float mean = Hl[p * C + 2 * j], logs = Hl[p * C + 2 * j + 1];
      if (a.hsave) { a.hsave[m * C + 2 * j] = mean; a.hsave[m * C + 2 * j + 1] = logs; }
      const float d = a.x[m * C + Ch + j] - mean;
      lp = -0.5f * (2.f * logs + d * d * expf(-2.f * logs) + SP_LN_2PI);
      a.z1out[m * Ch + j] = Zt[((sm * RS + rr + halo) * W + xx) * Ch + j];
    }
    LS[idx] = lp;
  }
  __syncthreads();
  // per-sample sums in fixed order: wave w takes samples w, w + 4, ...
  const int lane = tid & 63, wid = tid >> 6, pps = R * W * Ch;
  for (int sm = wid; sm < TS; sm += 4) {
    float v = 0.f;
    for (int e = lane; e < pps; e += 64) v += LS[sm * pps + e];
    v = wave_sum(v);
    if (lane == 0 && b0 + sm < s.B) a.ldpart[(long)(b0 + sm) * g.tps + tile] = v;
  }
}

struct SplitFusedBwd {
  nfk_split2d s;
  SplitGeom g;
  const float* x;        // [M][C] saved input
  const float* h;        // [M][C] saved (mean, logs)
  const float* w;        // conv.weight [C][C/2][3][3]
  const float* gz1;      // [M][C/2] gradient wrt the output z1
  const float* gext;     // [M][C/2] or null
  const float* gld;      // [B]
  float* gx;             // [M][C]
  float* part;           // [ntiles][2][C]: (sum g_a, 3 sum g_h h) per workgroup
  float* wslab;          // [ntiles][C][9 * C/2]: partial weight gradient per workgroup, column tap * C/2 + ci
};

// LDS: Xt [npix_s][C] | Ht [npix_s][C] | Ga [npix_s][C] | Wl [9 * C][Chp] (dgrad layout: [tap][co][ci]) | red [256 / Ch][4 * Ch]
__global__ void __launch_bounds__(256) split2d_bwd_fused_kernel(SplitFusedBwd a) {
  extern __shared__ __attribute__((aligned(16))) float smem_sp[];
  const nfk_split2d& s = a.s;
  const SplitGeom& g = a.g;
  const int C = s.C, Ch = C / 2, W = s.W, H = s.H, HW = H * W, lw = g.lw, Chp = (Ch + 3) & ~3;
  const int R = g.R, TS = g.TS, halo = g.halo, RS = g.RS, NS = g.npix_s, NPX = g.NPX;
  float* Xt = smem_sp;
  float* Ht = Xt + NS * C;
  float* Ga = Ht + NS * C;
  float* Wl = Ga + NS * C;
  float* red = Wl + 9 * C * Chp;
  const int tid = threadIdx.x, wg = blockIdx.x;
  const int b0 = TS == 1 ? wg / g.tps : wg * TS, tile = TS == 1 ? wg - b0 * g.tps : 0, row0 = tile * R;
  for (int base = tid; base < NS * C; base += 1024) {   // (loads batched four deep, as in the forward)
    float vx[4], vh[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = base + u * 256;
      const int ps = idx / C, c = idx - ps * C;
      const int sm = ps / (RS * W), rem = ps - sm * RS * W, r = rem >> lw, xx = rem & (W - 1);
      const int y = row0 + r - halo, b = b0 + sm;
      const bool ok = idx < NS * C && (unsigned)y < (unsigned)H && b < s.B;
      const long m = (long)b * HW + y * W + xx;
      vx[u] = ok ? a.x[m * C + c] : 0.f;
      vh[u] = ok ? a.h[m * C + c] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (base + u * 256 < NS * C) { Xt[base + u * 256] = vx[u]; Ht[base + u * 256] = vh[u]; }
  }
  for (int base = tid; base < C * Ch * 9; base += 1024) {
    float v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = base + u * 256 < C * Ch * 9 ? a.w[base + u * 256] : 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = base + u * 256;
      const int co = idx / (Ch * 9), r = idx - co * Ch * 9, ci = r / 9, tap = r - ci * 9;
      if (idx < C * Ch * 9) Wl[(tap * C + co) * Chp + ci] = v[u];
    }
  }
  __syncthreads();
  // g_a on every staged pixel (halo rows too: the transposed convolution reads them); own pixels also give g_z2 and the
  // per-channel sums.  Same arithmetic as split_bwd_kernel.
  float sb0 = 0.f, sl0 = 0.f, sb1 = 0.f, sl1 = 0.f;   // this thread's column j = tid % Ch, if tid < (256 / Ch) * Ch
  const int G = 256 / Ch, jn = tid % Ch, gn = tid / Ch;
  if (gn < G)
    for (int ps = gn; ps < NS; ps += G) {
      const int sm = ps / (RS * W), rem = ps - sm * RS * W, r = rem >> lw, xx = rem & (W - 1);
      const int y = row0 + r - halo, b = b0 + sm;
      const bool inside = (unsigned)y < (unsigned)H && b < s.B;
      float ga0 = 0.f, ga1 = 0.f;
      if (inside) {
        const float gl = a.gld[b];
        const float mean = Ht[ps * C + 2 * jn], logs = Ht[ps * C + 2 * jn + 1];
        const float d = Xt[ps * C + Ch + jn] - mean, iv = expf(-2.f * logs);
        const float gmean = gl * d * iv, glogs = gl * (d * d * iv - 1.f);
        ga0 = gmean * s.es[2 * jn];
        ga1 = glogs * s.es[2 * jn + 1];
        if (r >= halo && r < halo + R) {   // one of the tile's own pixels
          a.gx[((long)b * HW + y * W + xx) * C + Ch + jn] = -gmean;
          sb0 += ga0; sl0 += gmean * mean; sb1 += ga1; sl1 += glogs * logs;
        }
      }
      Ga[ps * C + 2 * jn] = ga0;
      Ga[ps * C + 2 * jn + 1] = ga1;
    }
  // column sums over the thread groups in fixed order (as split_bwd_kernel)
  if (gn < G) {
    float* row = red + gn * 4 * Ch;
    row[jn] = sb0; row[Ch + jn] = sl0; row[2 * Ch + jn] = sb1; row[3 * Ch + jn] = sl1;
  }
  __syncthreads();
  for (int jj = tid; jj < Ch; jj += 256) {
    float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
    for (int gg = 0; gg < G; ++gg) {
      const float* row = red + gg * 4 * Ch;
      t0 += row[jj]; t1 += row[Ch + jj]; t2 += row[2 * Ch + jj]; t3 += row[3 * Ch + jj];
    }
    float* prt = a.part + (long)wg * 2 * C;
    prt[2 * jj] = t0; prt[2 * jj + 1] = t2;
    prt[C + 2 * jj] = 3.f * t1; prt[C + 2 * jj + 1] = 3.f * t3;
  }
  // transposed convolution: g_z1[p][ci] = g_z1_out + g_ext + sum_tap sum_co g_a[p - d(tap)][co] W[co][ci][tap]
  const int G4 = Chp / 4;
  for (int t = tid; t < NPX * G4; t += 256) {
    const int p = t / G4, o = (t - p * G4) * 4;
    const int sm = p / (R * W), rem = p - sm * R * W, rr = rem >> lw, xx = rem & (W - 1);
    const int b = b0 + sm;
    if (b >= s.B) continue;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int dy = tap / 3 - 1, dx = tap % 3 - 1;
      const int rs = rr + halo - dy, xn = xx - dx;   // the output pixel q = p - d(tap) whose tap `tap` reads p
      if ((unsigned)rs >= (unsigned)RS || (unsigned)xn >= (unsigned)W) continue;
      const float* gr = Ga + ((sm * RS + rs) * W + xn) * C;
      const float* wr = Wl + tap * C * Chp + o;
#pragma unroll 2
      for (int co = 0; co < C; co += 2) {
        const float g0 = gr[co], g1 = gr[co + 1];
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(wr + co * Chp), w1 = *reinterpret_cast<const f32x4*>(wr + (co + 1) * Chp);
        acc += w0 * g0;
        acc += w1 * g1;
      }
    }
    const long m = (long)b * HW + (row0 + rr) * W + xx;
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (o + r < Ch) {
        float v = a.gz1 ? a.gz1[m * Ch + o + r] : 0.f;
        if (a.gext) v += a.gext[m * Ch + o + r];
        a.gx[m * C + o + r] = v + acc[r];
      }
  }
  // partial weight gradient of this tile: gW[co][tap * Ch + ci] = sum over own pixels p of g_a[p][co] z1[p + d(tap)][ci]
  const int J = 9 * Ch;
  float* slab = a.wslab + (long)wg * C * J;
  for (int idx = tid; idx < C * J; idx += 256) {
    const int co = idx / J, k = idx - co * J, tap = k / Ch, ci = k - tap * Ch;
    const int dy = tap / 3 - 1, dx = tap % 3 - 1;
    const int r_lo = -(halo + dy) > 0 ? -(halo + dy) : 0, r_hi = RS - halo - dy < R ? RS - halo - dy : R;
    const int x_lo = dx < 0 ? -dx : 0, x_hi = dx > 0 ? W - dx : W;
    float acc = 0.f;
    for (int sm = 0; sm < TS; ++sm)
      for (int rr = r_lo; rr < r_hi; ++rr) {
        const float* gp = Ga + ((sm * RS + rr + halo) * W + x_lo) * C + co;
        const float* xp = Xt + ((sm * RS + rr + halo + dy) * W + x_lo + dx) * C + ci;
#pragma unroll 4
        for (int xx = x_lo; xx < x_hi; ++xx, gp += C, xp += C) acc += gp[0] * xp[0];
      }
    slab[idx] = acc;
  }
}

constexpr size_t SP_LDS_MAX = 160 * 1024;
static size_t sp_fwd_lds(const nfk_split2d& s, const SplitGeom& g) {
  const int C = s.C, Ch = C / 2, Cp = (C + 3) & ~3;
  return (size_t)(((g.npix_s * Ch + 3) & ~3) + 9 * Ch * Cp + g.NPX * C + g.NPX * Ch) * sizeof(float);
}
static size_t sp_bwd_lds(const nfk_split2d& s, const SplitGeom& g) {
  const int C = s.C, Ch = C / 2, Chp = (Ch + 3) & ~3;
  return (size_t)(3 * g.npix_s * C + 9 * C * Chp + 1024) * sizeof(float);
}

int split2d_fused_tiles(const nfk_split2d& s) {   // 0: geometry not supported; else workgroups of the fused kernels
  SplitGeom g;
  if (!sp_geom(s, &g)) return 0;
  if (sp_fwd_lds(s, g) > SP_LDS_MAX || sp_bwd_lds(s, g) > SP_LDS_MAX) return 0;
  return g.ntiles;
}
int split2d_fused_tps(const nfk_split2d& s) {
  SplitGeom g;
  return sp_geom(s, &g) ? g.tps : 0;
}

int split2d_fused_fwd(const nfk_split2d& s, const float* x, const float* w, float* z1out, float* hsave, float* ldpart,
                      hipStream_t st) {
  SplitFusedFwd a{};
  if (!sp_geom(s, &a.g)) return NFK_ERR_ARG;
  a.s = s; a.x = x; a.w = w; a.z1out = z1out; a.hsave = hsave; a.ldpart = ldpart;
  const size_t lds = sp_fwd_lds(s, a.g);
  if (lds > SP_LDS_MAX) return NFK_ERR_ARG;
  static const bool attr = hipFuncSetAttribute(reinterpret_cast<const void*>(split2d_fwd_fused_kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)SP_LDS_MAX) == hipSuccess;
  if (!attr) return NFK_ERR_LAUNCH;
  hipLaunchKernelGGL(split2d_fwd_fused_kernel, dim3(a.g.ntiles), dim3(256), lds, st, a);
  NFK_LAUNCH_CHECK();
  return NFK_OK;
}

int split2d_fused_bwd(const nfk_split2d& s, const float* x, const float* h, const float* w, const float* gz1, const float* gext,
                      const float* gld, float* gx, float* part, float* wslab, hipStream_t st) {
  SplitFusedBwd a{};
  if (!sp_geom(s, &a.g)) return NFK_ERR_ARG;
  a.s = s; a.x = x; a.h = h; a.w = w; a.gz1 = gz1; a.gext = gext; a.gld = gld; a.gx = gx; a.part = part; a.wslab = wslab;
  const size_t lds = sp_bwd_lds(s, a.g);
  if (lds > SP_LDS_MAX) return NFK_ERR_ARG;
  static const bool attr = hipFuncSetAttribute(reinterpret_cast<const void*>(split2d_bwd_fused_kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)SP_LDS_MAX) == hipSuccess;
  if (!attr) return NFK_ERR_LAUNCH;
  hipLaunchKernelGGL(split2d_bwd_fused_kernel, dim3(a.g.ntiles), dim3(256), lds, st, a);
  NFK_LAUNCH_CHECK();
  return NFK_OK;
}

}  // namespace nfk
