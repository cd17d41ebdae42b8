// Weight repack bodies shared by the stand-alone repack launches (nfk_gemm.hip, nfk_gemm_bf16.hip) and the merged
// prep + repack launch (nfk_flow.hip): device code cannot be linked across translation units without -fgpu-rdc.
#pragma once
#include "nfk_internal.h"

namespace nfk {

typedef unsigned int rp_u32x4 __attribute__((ext_vector_type(4)));
constexpr int RP_BK = 32;    // f32 k-chunk
constexpr int RP_BKH = 64;   // bf16 k-chunk
__device__ __forceinline__ unsigned rp_pack2(float lo, float hi) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  typedef float f32x2p __attribute__((ext_vector_type(2)));
  const f32x2p f = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2));
}

// f32: W[co][ci][taps] -> wt[co][tap*Cin + ci] (zero padded to Kp) and wd[ci][tap'*Cout + co] (taps flipped)
// (bx, nbx): this worker's index among the workers of the weight; (tid, nthr): the thread's index among the worker's
// threads -- a worker is a workgroup, or one 256-thread half of a wider one (prep_repack_kernel)
__device__ __forceinline__ void repack_f32_body(const nfk_repack_desc& d, const int bx, const int nbx, const int tid,
                                                const int nthr) {
  const int Kp = (d.taps * d.Cin + RP_BK - 1) / RP_BK * RP_BK, Kpd = (d.taps * d.Cout + RP_BK - 1) / RP_BK * RP_BK;
  const long total_f = (long)d.Cout * Kp, total_d = d.wd ? (long)d.Cin * Kpd : 0;
  for (long t = bx * (long)nthr + tid; t < total_f + total_d; t += nbx * (long)nthr) {
    if (t < total_f) {
      const int co = t / Kp, k = t % Kp;
      float v = 0.f;
      if (k < d.taps * d.Cin) v = d.w[((long)co * d.Cin + k % d.Cin) * d.taps + k / d.Cin];
      d.wt[t] = v;
    } else {
      const long u = t - total_f;
      const int ci = u / Kpd, k = u % Kpd;
      float v = 0.f;
      if (k < d.taps * d.Cout) v = d.w[((long)(k % d.Cout) * d.Cin + ci) * d.taps + (d.taps - 1 - k / d.Cout)];
      d.wd[u] = v;
    }
  }
}

// conv.weight [Cout][Cin][taps] (fp32) -> bf16 wt [Cout][Kp64], wd [Cin][Kp64(taps*Cout)] (flipped taps) and wf (wt in
// MFMA-fragment order).  All three are permutations of the source; read straight from HBM they are gathers (wd is a
// transpose: every element of a 16-byte store from another 2-18 KB row), which made this launch 104 us per training
// step.  Here a workgroup stages a tile of the source -- RC output channels x RI input channels x all taps, read as
// contiguous runs -- in LDS and writes every 16-byte piece of the three layouts that the tile owns from there.
//   tile shapes: 1x1 conv 64 x 64; wide 3x3 (conv1) 32 x (16 | all of Cin <= 24); narrow 3x3 (conv3, Cout <= 48)
//   all of Cout x 16.  A piece is 8 consecutive k: 8 input channels of one tap (wt / wf) or 8 output channels of one
//   tap (wd) when the channel count is a multiple of 8; otherwise the tile must span that whole channel dimension.
//   Shapes no tile scheme divides take the gather path below.
constexpr int RP_LDS_FLOATS = 7168;   // 28 KiB
struct RepackTiling { int RC, RI, ok; };
__device__ __forceinline__ RepackTiling repack_tiling(int Cout, int Cin, int taps) {
  RepackTiling t;
  t.RI = Cin <= 24 ? Cin : (taps == 1 ? 64 : 16);
  if (t.RI > Cin) t.RI = Cin;
  t.RC = Cout <= 48 ? Cout : (taps == 1 ? 64 : 32);
  t.ok = (t.RI == Cin || (Cin & 7) == 0) && (t.RC == Cout || (Cout & 7) == 0) && (t.RI == Cin || (t.RI & 7) == 0) &&
         t.RC * (t.RI * taps + 1) <= RP_LDS_FLOATS;
  return t;
}

__device__ __forceinline__ void repack_gather_bf16(const nfk_repack_desc& d, const int bx, const int nbx, const int tid,
                                                   const int nthr) {
  // one thread = 8 consecutive output elements (one 16-byte store); 32-bit index math, one division per 8 elements
  const int K = d.taps * d.Cin, Kd = d.taps * d.Cout;
  const int Kp = (K + RP_BKH - 1) / RP_BKH * RP_BKH, Kpd = (Kd + RP_BKH - 1) / RP_BKH * RP_BKH;
  const int rows_g = (d.frag == 1) ? d.Cout : (d.Cout + 15) / 16 * 16;
  const int n8_f = d.Cout * (Kp / 8), n8_d = d.wd ? d.Cin * (Kpd / 8) : 0;
  const int n8_g = (d.wf && d.frag) ? rows_g * (Kp / 8) : 0;
  __bf16* wt = reinterpret_cast<__bf16*>(d.wt);
  __bf16* wd = reinterpret_cast<__bf16*>(d.wd);
  __bf16* wf = reinterpret_cast<__bf16*>(d.wf);
  __bf16* wg = (d.wd && d.gfrag) ? reinterpret_cast<__bf16*>(d.wg) : nullptr;
  const bool fonly = (d.flags & NFK_REPACK_FRAG_ONLY) != 0;
  for (int t = bx * nthr + tid; t < n8_f + n8_d + n8_g; t += nbx * nthr) {
    float v[8];
    __bf16* dst;
    if (t < n8_f + n8_g) {
      // forward layouts: value(co, k) = w[co][k % Cin][k / Cin], zero outside
      int co, k0;
      if (t < n8_f) {                       // row-major [Cout][Kp]
        co = t / (Kp / 8); k0 = (t - co * (Kp / 8)) * 8;
        dst = wt + (long)t * 8;
      } else {                              // fragment order: 8 elements of lane `lane` of fragment `fr`
        const int u = t - n8_f, lane = u & 63, fr = u >> 6;
        if (d.frag == 1) {                  // 32x32x16 A fragments: 32 rows x 16 k
          const int KS = Kp / 16, nt = fr / KS, ks = fr - nt * KS;
          co = nt * 32 + (lane & 31); k0 = ks * 16 + (lane >> 5) * 8;
        } else {                            // 16x16x32 A fragments: 16 rows x 32 k
          const int KS = Kp / 32, nt = fr / KS, ks = fr - nt * KS;
          co = nt * 16 + (lane & 15); k0 = ks * 32 + (lane >> 4) * 8;
        }
        dst = wf + (long)u * 8;
      }
      int tap = k0 / d.Cin, ci = k0 - tap * d.Cin;
      const float* wrow = d.w + (long)co * d.Cin * d.taps;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        v[e] = (co < d.Cout && k0 + e < K) ? wrow[ci * d.taps + tap] : 0.f;
        if (++ci == d.Cin) { ci = 0; ++tap; }
      }
    } else {
      // dgrad layout [Cin][Kpd]: value(ci, k) = w[k % Cout][ci][taps - 1 - k / Cout]
      const int u = t - n8_f - n8_g, ci = u / (Kpd / 8), k0 = (u - ci * (Kpd / 8)) * 8;
      int tap = k0 / d.Cout, co = k0 - tap * d.Cout;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        v[e] = (k0 + e < Kd) ? d.w[((long)co * d.Cin + ci) * d.taps + (d.taps - 1 - tap)] : 0.f;
        if (++co == d.Cout) { co = 0; ++tap; }
      }
      dst = wd + (long)u * 8;
      if (wg) {   // the same 8 elements once more, at their place in the fragment-ordered copy
        const rp_u32x4 pkg = {rp_pack2(v[0], v[1]), rp_pack2(v[2], v[3]), rp_pack2(v[4], v[5]), rp_pack2(v[6], v[7])};
        long ug;
        if (d.gfrag == 1) ug = ((long)((ci >> 5) * (Kpd / 16) + (k0 >> 4)) << 6) + (ci & 31) + 32 * ((k0 >> 3) & 1);
        else ug = ((long)((ci >> 4) * (Kpd / 32) + (k0 >> 5)) << 6) + (ci & 15) + 16 * ((k0 >> 3) & 3);
        *reinterpret_cast<rp_u32x4*>(wg + ug * 8) = pkg;
      }
    }
    // (fragment-only repack: the row-major copies are skipped where their fragment-ordered twins exist)
    if (fonly && ((t < n8_f && n8_g > 0) || (t >= n8_f + n8_g && wg))) continue;
    const rp_u32x4 pk = {rp_pack2(v[0], v[1]), rp_pack2(v[2], v[3]), rp_pack2(v[4], v[5]), rp_pack2(v[6], v[7])};
    *reinterpret_cast<rp_u32x4*>(dst) = pk;
  }
}

__device__ __forceinline__ void repack_tiles_bf16(const nfk_repack_desc& d, const RepackTiling tl, const int bx, const int nbx,
                                                  float* T, const int tid, const int nthr) {
  const int Cout = d.Cout, Cin = d.Cin, taps = d.taps, RC = tl.RC, RI = tl.RI;
  const int K = taps * Cin, Kd = taps * Cout;
  const int Kp = (K + RP_BKH - 1) / RP_BKH * RP_BKH, Kpd = (Kd + RP_BKH - 1) / RP_BKH * RP_BKH;
  const int nCT = (Cout + RC - 1) / RC, nIT = (Cin + RI - 1) / RI;
  const int RL = RI * taps, RLp = RL + 1;   // LDS row: one output channel's (ci, tap) run; odd pitch for the column reads
  const bool have_wf = d.wf && d.frag;
  const int rows_g = !have_wf ? Cout : ((d.frag == 1) ? Cout : (Cout + 15) / 16 * 16);   // wf rows (zero behind Cout)
  __bf16* wt = reinterpret_cast<__bf16*>(d.wt);
  __bf16* wd = reinterpret_cast<__bf16*>(d.wd);
  __bf16* wf = reinterpret_cast<__bf16*>(d.wf);
  __bf16* wg = (d.wd && d.gfrag) ? reinterpret_cast<__bf16*>(d.wg) : nullptr;
  const bool fonly = (d.flags & NFK_REPACK_FRAG_ONLY) != 0;
  const bool put_wt = !(fonly && have_wf), put_wd = !(fonly && wg);   // the row-major copies: skipped where a fragment copy exists
  const unsigned mCin = 0xFFFFFFFFu / (unsigned)Cin + 1u, mCout = 0xFFFFFFFFu / (unsigned)Cout + 1u;   // n / C for n < 2^16
  // the trip count is the same for every worker of the weight (barriers are workgroup-wide, and a workgroup may hold
  // two workers): a worker without a tile in the last trip runs it with an empty tile
  for (int tile0 = 0; tile0 < nCT * nIT; tile0 += nbx) {
    const int tile = tile0 + bx;
    const bool live = tile < nCT * nIT;
    const int ct = tile / nIT, it = tile - ct * nIT, c0 = ct * RC, i0 = it * RI;
    const int rc = live ? min(RC, Cout - c0) : 0, ri = live ? min(RI, Cin - i0) : 0;   // ragged last tiles
    __syncthreads();
    for (int idx = tid; idx < rc * ri * taps; idx += nthr) {
      const int r = idx / (ri * taps), x = idx - r * (ri * taps);
      T[r * RLp + x] = d.w[((long)(c0 + r) * Cin + i0) * taps + x];
    }
    __syncthreads();
    // ---- wt / wf: rows of this tile, k-pieces of 8 whose input channels lie in [i0, i0 + ri)
    const int per_tap = ri / 8;                                  // used when the tile does not span all of Cin
    const int nq = (ri == Cin) ? Kp / 8 : taps * per_tap + (i0 == 0 ? (Kp - K) / 8 : 0);
    const int rows = !live ? 0 : ((nCT == 1) ? rows_g : rc);     // a lone row tile also writes wf's zero rows
    for (int p = tid; p < rows * nq; p += nthr) {
      const int r = p / nq, q = p - r * nq, co = c0 + r;
      int k0;
      if (ri == Cin) k0 = q * 8;
      else if (q < taps * per_tap) { const int tap = q / per_tap; k0 = tap * Cin + i0 + (q - tap * per_tap) * 8; }
      else k0 = K + (q - taps * per_tap) * 8;
      float v[8];
      int tap = (int)__umulhi((unsigned)k0, mCin), ci = k0 - tap * Cin;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        v[e] = (co < Cout && k0 + e < K) ? T[r * RLp + (ci - i0) * taps + tap] : 0.f;
        if (++ci == Cin) { ci = 0; ++tap; }
      }
      const rp_u32x4 pk = {rp_pack2(v[0], v[1]), rp_pack2(v[2], v[3]), rp_pack2(v[4], v[5]), rp_pack2(v[6], v[7])};
      if (co < Cout && put_wt) *reinterpret_cast<rp_u32x4*>(wt + (long)co * Kp + k0) = pk;
      if (have_wf) {
        long u;
        if (d.frag == 1) u = ((long)((co >> 5) * (Kp / 16) + (k0 >> 4)) << 6) + (co & 31) + 32 * ((k0 >> 3) & 1);
        else u = ((long)((co >> 4) * (Kp / 32) + (k0 >> 5)) << 6) + (co & 15) + 16 * ((k0 >> 3) & 3);
        *reinterpret_cast<rp_u32x4*>(wf + u * 8) = pk;
      }
    }
    // ---- wd: rows ci of this tile, k-pieces of 8 whose output channels lie in [c0, c0 + rc)
    if (wd) {
      const int per_tap_d = rc / 8;
      const int nqd = (rc == Cout) ? Kpd / 8 : taps * per_tap_d + (c0 == 0 ? (Kpd - Kd) / 8 : 0);
      for (int p = tid; p < ri * nqd; p += nthr) {
        const int x = p / nqd, q = p - x * nqd, ci = i0 + x;
        int k0;
        if (rc == Cout) k0 = q * 8;
        else if (q < taps * per_tap_d) { const int tap = q / per_tap_d; k0 = tap * Cout + c0 + (q - tap * per_tap_d) * 8; }
        else k0 = Kd + (q - taps * per_tap_d) * 8;
        float v[8];
        int tap = (int)__umulhi((unsigned)k0, mCout), co = k0 - tap * Cout;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          v[e] = (k0 + e < Kd) ? T[(co - c0) * RLp + x * taps + (taps - 1 - tap)] : 0.f;
          if (++co == Cout) { co = 0; ++tap; }
        }
        const rp_u32x4 pk = {rp_pack2(v[0], v[1]), rp_pack2(v[2], v[3]), rp_pack2(v[4], v[5]), rp_pack2(v[6], v[7])};
        if (put_wd) *reinterpret_cast<rp_u32x4*>(wd + (long)ci * Kpd + k0) = pk;
        if (wg) {
          long ug;
          if (d.gfrag == 1) ug = ((long)((ci >> 5) * (Kpd / 16) + (k0 >> 4)) << 6) + (ci & 31) + 32 * ((k0 >> 3) & 1);
          else ug = ((long)((ci >> 4) * (Kpd / 32) + (k0 >> 5)) << 6) + (ci & 15) + 16 * ((k0 >> 3) & 3);
          *reinterpret_cast<rp_u32x4*>(wg + ug * 8) = pk;
        }
      }
    }
  }
}

// one workgroup's share of one conv weight; T: RP_LDS_FLOATS floats of LDS
__device__ __forceinline__ void repack_bf16_body(const nfk_repack_desc& d, const int bx, const int nbx, float* T,
                                                 const int tid, const int nthr) {
  const RepackTiling tl = repack_tiling(d.Cout, d.Cin, d.taps);
  // (several row tiles cannot write the zero rows a 16-row fragment layout keeps behind Cout)
  const bool ragged_wf = d.wf && d.frag == 2 && (d.Cout & 15) && tl.RC != d.Cout;
  if (tl.ok && !ragged_wf) repack_tiles_bf16(d, tl, bx, nbx, T, tid, nthr);
  else repack_gather_bf16(d, bx, nbx, tid, nthr);
}

}  // namespace nfk
