// nfk fused FlowStep forward (bf16 operands / fp32 accumulate): ONE launch per FlowStep
// (models/flows.py:142-171) instead of five.
//
//   workgroup = a spatial tile of whole image rows (+1 halo row each side when the tile is not the whole
//   image), 64 output pixels:   CIFAR level 1: 4 rows of one image, level 2: one 8x8 image, level 3: four 4x4 images
//   1. mix      z = Wf x + cf on the tile rows (+2 halo rows)                      -> LDS (fp32)
//   2. conv1    3x3 C/2->hid, ActNorm, ReLU: im2col tile in LDS x W1 rows streamed from L2 straight into
//               MFMA A-operand registers (every wave owns its own output channels: no LDS staging)   -> h1 in LDS (bf16)
//   3. conv2    1x1 hid->hid in 64-channel chunks (W2 rows from L2, h1 from LDS)   -> h2 chunk in LDS
//   4. conv3    3x3 hid->C: each chunk of h2 is consumed immediately as a K-slice (9 taps = shifted LDS rows),
//               the K range is split over the 4 waves and reduced through LDS at the end
//   5. tail     bias, exp(3 logs), sigmoid(.+2), (z2+shift)*scale, partial sum of log-scales per tile
//   h1 / h2 never travel through HBM between the convolutions (they are only written out, once, when a
//   backward will follow); per-step HBM/L2 traffic is the weights (once per workgroup) plus O(C) per pixel.
// The per-sample log-det is deferred: every step leaves per-tile partial sums and nfk_ld_finish adds them
// in fixed order at the end of a run of steps (bitwise reproducible, no atomics).
#include "nfk_internal.h"

namespace nfk {

typedef __bf16 bf16x8f __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4f __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4f __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2f __attribute__((ext_vector_type(2)));

struct FusedArgs {
  nfk_step2d s;
  const float* x; int ldx;
  float* zsave;          // [M][C]   mixed z (training) or null
  __bf16* h1save;        // [M][hid] or null
  __bf16* h2save;        // [M][hid] or null
  float* hsave;          // [M][Cout] or null
  float* zout;           // [M][C]
  float* ldpart;         // [B][tps] partial log-det of this step
  int R, TS, halo;       // rows per tile, samples per tile, 1 if the tile needs halo rows
  unsigned long long* stamps;  // diagnostic: [grid][16] cycle stamps (null in production)
  int dbg;                     // diagnostic bisect bits (0 in production)
};

struct FusedGeom {
  int RH, RZ, P_h, P_out, MT_h, MT_o, K1, K1p, X1S, NT3;
  size_t oZm, oWf, oPar, oX1, oH1, oH2c, total;
};

__host__ __device__ static inline FusedGeom fused_geom(const nfk_step2d& s, int R, int TS, int halo) {
  FusedGeom g;
  const int W = s.W, C = s.C, Ch = C / 2;
  g.RH = R + 2 * halo;
  g.RZ = R + 4 * halo;
  g.P_h = TS * g.RH * W;
  g.P_out = TS * R * W;
  g.MT_h = (g.P_h + 15) / 16;
  g.MT_o = g.P_out / 16;
  g.K1 = 9 * Ch;
  g.K1p = g.K1 <= 64 ? 64 : (g.K1 <= 128 ? 128 : 256);   // conv1 k-steps padded to 2 / 4 / 8 (zeros in the im2col tile)
  g.X1S = g.K1p * 2 + 16;                       // bytes per im2col row (+16: rows start on different banks)
  g.NT3 = (s.Cout + 15) / 16;
  size_t o = 0;
  g.oZm = o; o += (size_t)TS * g.RZ * W * C * 4;
  o = (o + 15) & ~size_t(15);
  g.oWf = o; o += (size_t)(C * ((C + 3) & ~3) + C) * 4;
  o = (o + 15) & ~size_t(15);
  g.oPar = o; o += (size_t)(4 * s.hidden + 2 * ((s.Cout + 3) & ~3)) * 4;
  g.oX1 = o; o += (size_t)g.MT_h * 16 * g.X1S;
  o = (o + 15) & ~size_t(15);
  g.oH1 = o; o += (size_t)g.MT_h * 16 * s.hidden * 2;
  g.oH2c = o; o += (size_t)g.MT_h * 16 * 128;
  g.total = o;
  return g;
}

// byte offset of element (px, ch) in the h1 image: 2*hid-byte rows, low 4 bits of the 16-byte slot XORed with px
__device__ __forceinline__ int h1_off(int px, int ch, int hid) {
  const int slot = ch >> 3;
  return px * hid * 2 + ((((slot ^ px) & 15) | (slot & ~15)) << 4) + ((ch & 7) << 1);
}
// h2 chunk image: 128-byte rows (64 channels), slot ^ (px>>1)&7  (same geometry as the GEMM tiles)
__device__ __forceinline__ int h2_off(int px, int ch) {
  return px * 128 + ((((ch >> 3) ^ ((px >> 1) & 7))) << 4) + ((ch & 7) << 1);
}
__device__ __forceinline__ unsigned pack2f(float lo, float hi) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  bf16x2 v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(unsigned, v);
}

template <int MT_H, int NT3, int K1M>
__global__ void __launch_bounds__(256) fused_step_fwd_kernel(FusedArgs a) {
  constexpr int K2S = 16;   // hidden == 512: 16 k-steps of 32 in the 1x1 conv
  extern __shared__ __attribute__((aligned(16))) char smem_f[];
  const nfk_step2d& s = a.s;
  const int W = s.W, H = s.H, C = s.C, Ch = C / 2, C2 = C - Ch, hid = s.hidden, Cout = s.Cout, HW = H * W;
  const int R = a.R, TS = a.TS, halo = a.halo;
  const FusedGeom g = fused_geom(s, R, TS, halo);
  constexpr int MT_O = 4;  // 64 output pixels
  float* Zm = reinterpret_cast<float*>(smem_f + g.oZm);
  float* Wl = reinterpret_cast<float*>(smem_f + g.oWf);
  char* X1 = smem_f + g.oX1;
  char* H1 = smem_f + g.oH1;
  char* H2c = smem_f + g.oH2c;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, r16 = lane & 15, q4 = lane >> 4;

  // tile -> (first sample, first row)
  const int tps = (TS == 1) ? H / R : 1;
  const int b0 = (TS == 1) ? blockIdx.x / tps : blockIdx.x * TS;
  const int row0 = (TS == 1) ? (blockIdx.x % tps) * R : 0;
  const int zh = 2 * halo;  // halo rows of the Zm region
  auto stamp = [&](int i) {
    if (a.stamps && tid == 0) a.stamps[(long)blockIdx.x * 16 + i] = __builtin_readcyclecounter();
  };
  stamp(0);

  // ------------------------------------------------------------------ 1. mix
  const int Cp = (C + 3) & ~3;
  for (int idx = tid; idx < C * C; idx += 256) {
    const int o = idx / C, c = idx - o * C;
    Wl[c * Cp + o] = s.Wf[idx];
  }
  for (int idx = tid; idx < C; idx += 256) Wl[C * Cp + idx] = s.cf[idx];
  // per-channel parameters of the three convs -> LDS: epilogues must not issue global loads (vmcnt is an in-order
  // counter: a late load would make the wave wait for every prefetched weight fragment issued before it)
  float* Pb1 = reinterpret_cast<float*>(smem_f + g.oPar);
  float* Pe1 = Pb1 + hid;
  float* Pb2 = Pe1 + hid;
  float* Pe2 = Pb2 + hid;
  float* Pb3 = Pe2 + hid;
  float* Pe3 = Pb3 + ((Cout + 3) & ~3);
  for (int idx = tid; idx < hid; idx += 256) {
    Pb1[idx] = s.b1[idx]; Pe1[idx] = s.es1[idx]; Pb2[idx] = s.b2[idx]; Pe2[idx] = s.es2[idx];
  }
  for (int idx = tid; idx < Cout; idx += 256) { Pb3[idx] = s.bias3[idx]; Pe3[idx] = s.es3[idx]; }
  // raw x rows of the tile (+halo): contiguous in HBM per sample -> coalesced copy into the h1 region (free until conv1)
  float* Xraw = reinterpret_cast<float*>(H1);
  {
    const int rowlen = W * C, nrows = TS * g.RZ;
    for (int idx = tid; idx < nrows * rowlen; idx += 256) {
      const int rrow = idx / rowlen, e = idx - rrow * rowlen;
      const int sm = rrow / g.RZ, rz = rrow - sm * g.RZ;
      const int y = row0 + rz - zh, b = b0 + sm;
      float v = 0.f;
      if ((unsigned)y < (unsigned)H && b < s.B) {
        const int xx = e / C, c = e - xx * C;
        v = a.x[((long)b * HW + y * W + xx) * a.ldx + c];
      }
      Xraw[idx] = v;
    }
  }
  __syncthreads();
  {
    const int G4 = Cp / 4, npz = TS * g.RZ * W;
    for (int t = tid; t < npz * G4; t += 256) {
      const int pz = t / G4, o = (t - pz * G4) * 4;
      const int sm = pz / (g.RZ * W), rem = pz - sm * g.RZ * W, rz = rem / W, xx = rem - rz * W;
      const int y = row0 + rz - zh, b = b0 + sm;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      const bool ok = (unsigned)y < (unsigned)H && b < s.B;
      if (ok) {
        const float* xr = Xraw + pz * C;
#pragma unroll
        for (int r = 0; r < 4; ++r) if (o + r < C) acc[r] = Wl[C * Cp + o + r];
        for (int c = 0; c < C; ++c) {
          const float xv = xr[c];
          const f32x4 w = *reinterpret_cast<const f32x4*>(Wl + c * Cp + o);
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[r] += w[r] * xv;
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) if (o + r < C) Zm[pz * C + o + r] = acc[r];
      if (ok && a.zsave && rz >= zh && rz < zh + R) {
        float* zp = a.zsave + ((long)b * HW + y * W + xx) * C + o;
#pragma unroll
        for (int r = 0; r < 4; ++r) if (o + r < C) zp[r] = acc[r];
      }
    }
  }
  __syncthreads();

  stamp(1);
  // ------------------------------------------------------------------ 2. im2col of z1 for the h-region (bf16)
  // one (h-pixel, tap) pair per thread iteration: Ch contiguous channels copied (or zeroed), no per-element index math
  {
    const int kpad = g.K1p - g.K1;   // zero columns behind the last tap
    for (int t = tid; t < MT_H * 16 * 10; t += 256) {
      const int ph = t / 10, tap = t - ph * 10;
      char* dst = X1 + ph * g.X1S;
      if (tap == 9) {   // tail padding of the row
        __bf16* d = reinterpret_cast<__bf16*>(dst) + g.K1;
        for (int e = 0; e < kpad; ++e) d[e] = (__bf16)0.f;
        continue;
      }
      const int sm = ph / (g.RH * W), rem = ph - sm * g.RH * W, rr = rem / W, xx = rem - rr * W;
      const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
      const int rz = rr + halo + dy, xn = xx + dx;   // Zm row index (Zm has `zh` halo rows, the h-region has `halo`)
      const bool ok = ph < g.P_h && (unsigned)rz < (unsigned)g.RZ && (unsigned)xn < (unsigned)W;
      const float* src = Zm + ((sm * g.RZ + rz) * W + xn) * C;
      __bf16* d = reinterpret_cast<__bf16*>(dst) + tap * Ch;
      for (int ci = 0; ci < Ch; ++ci) d[ci] = (__bf16)(ok ? src[ci] : 0.f);
    }
  }
  __syncthreads();

  stamp(2);
  // validity of h-region pixels (inside the image?) for this lane's column in each m-tile
  bool hvalid[MT_H];
  int hglob[MT_H];   // global pixel row index (for saving h1/h2) or -1
#pragma unroll
  for (int mt = 0; mt < MT_H; ++mt) {
    const int ph = mt * 16 + r16;
    const int sm = ph / (g.RH * W), rem = ph - sm * g.RH * W, rr = rem / W, xx = rem - rr * W;
    const int y = row0 + rr - halo, b = b0 + sm;
    hvalid[mt] = ph < g.P_h && (unsigned)y < (unsigned)H && b < s.B;
    hglob[mt] = (hvalid[mt] && rr >= halo && rr < halo + R) ? (b * HW + y * W + xx) : -1;
  }

  // ------------------------------------------------------------------ 3. conv1 -> H1
  {
    const __bf16* W1 = reinterpret_cast<const __bf16*>(s.wt1);
    const int Kp64 = (g.K1 + 63) / 64 * 64;   // row stride of the repacked weights
    const int ntiles = hid / 16;
    auto load_w1 = [&](int nt, bf16x8f (&fa)[K1M]) {
      // rows are Kp64 wide; k-steps beyond it read the next row (finite data) against zero im2col columns
      const __bf16* wrow = W1 + (long)(nt * 16 + r16) * Kp64 + q4 * 8;
#pragma unroll
      for (int ks = 0; ks < K1M; ++ks) fa[ks] = *reinterpret_cast<const bf16x8f*>(wrow + ks * 32);
    };
    auto conv1_tile = [&](int nt, const bf16x8f (&fa)[K1M]) {
      f32x4 acc[MT_H];
#pragma unroll
      for (int mt = 0; mt < MT_H; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
      bf16x8f fb[2][MT_H];
      auto rd1 = [&](int ks, bf16x8f (&f)[MT_H]) {
#pragma unroll
        for (int mt = 0; mt < MT_H; ++mt)
          f[mt] = *reinterpret_cast<const bf16x8f*>(X1 + (mt * 16 + r16) * g.X1S + (ks * 32 + q4 * 8) * 2);
      };
      rd1(0, fb[0]);
#pragma unroll
      for (int ks = 0; ks < K1M; ++ks) {
        if (ks + 1 < K1M) rd1(ks + 1, fb[(ks + 1) & 1]);   // next k-step's fragments in flight during these MFMAs
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mt = 0; mt < MT_H; ++mt)
          acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[ks], fb[ks & 1][mt], acc[mt], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      const int n = nt * 16 + q4 * 4;
      const f32x4 bb = *reinterpret_cast<const f32x4*>(Pb1 + n), es = *reinterpret_cast<const f32x4*>(Pe1 + n);
#pragma unroll
      for (int mt = 0; mt < MT_H; ++mt) {
        f32x4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf((acc[mt][r] + bb[r]) * es[r], 0.f);
        const u32x2f pk = {pack2f(v[0], v[1]), pack2f(v[2], v[3])};
        *reinterpret_cast<u32x2f*>(H1 + h1_off(mt * 16 + r16, n, hid)) = pk;
        if (a.h1save && hglob[mt] >= 0) *reinterpret_cast<u32x2f*>(a.h1save + (long)hglob[mt] * hid + n) = pk;
      }
    };
    bf16x8f w1a[K1M], w1b[K1M];
    load_w1(wid, w1a);
    for (int nt = wid; nt < ntiles; nt += 8) {   // hidden == 512: 32 n-tiles, 8 per wave, two per iteration
      load_w1(nt + 4, w1b);
      conv1_tile(nt, w1a);
      load_w1(min(nt + 8, ntiles - 4 + wid), w1a);
      conv1_tile(nt + 4, w1b);
    }
  }
  __syncthreads();

  stamp(3);
  // ------------------------------------------------------------------ 4./5. conv2 chunks feeding conv3
  const __bf16* W2 = reinterpret_cast<const __bf16*>(s.wt2);
  const __bf16* W3 = reinterpret_cast<const __bf16*>(s.wt3);
  const int Kp2 = (hid + 63) / 64 * 64, Kp3 = (9 * hid + 63) / 64 * 64;
  f32x4 acc3[NT3][MT_O];
#pragma unroll
  for (int i = 0; i < NT3; ++i)
#pragma unroll
    for (int j = 0; j < MT_O; ++j) acc3[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // conv3 B-operand addressing: this lane's output pixel in each m-tile -> (sample, row in h-region, column)
  int o_base[MT_O], o_x[MT_O], o_rr[MT_O];
#pragma unroll
  for (int mo = 0; mo < MT_O; ++mo) {
    const int po = mo * 16 + r16;
    const int sm = po / (R * W), rem = po - sm * R * W, rr = rem / W, xx = rem - rr * W;
    o_base[mo] = sm * g.RH * W;
    o_rr[mo] = rr + halo;
    o_x[mo] = xx;
  }

  // The weight streams are software-pipelined through TWO register sets: while chunk nc computes, every W2 / W3
  // fragment of chunk nc+1 is already in flight (one wave per SIMD owns 512 VGPRs, and nothing else hides the
  // L2 latency here).  K2S <= 16 (hidden <= 512) is guaranteed by the host.
  // conv3 B-operand byte offsets into the h2 chunk image, per (k-step of this wave, output m-tile): lane constants
  int c3_off[5][MT_O];
  unsigned c3_ok = 0u;
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const int ksi = wid + 4 * i;
    const int kc = min(ksi, 17), tap = kc >> 1, hh = kc & 1, dy = tap / 3 - 1, dx = tap % 3 - 1;
#pragma unroll
    for (int mo = 0; mo < MT_O; ++mo) {
      const int rr = o_rr[mo] + dy, xn = o_x[mo] + dx;
      const bool ok = ksi < 18 && (unsigned)rr < (unsigned)g.RH && (unsigned)xn < (unsigned)W;
      const int rc = min(max(rr, 0), g.RH - 1), xc = min(max(xn, 0), W - 1);
      c3_off[i][mo] = h2_off(o_base[mo] + rc * W + xc, hh * 32 + q4 * 8);
      c3_ok |= (ok ? 1u : 0u) << (i * MT_O + mo);
    }
  }
  const int nchunks = hid / 64;
  const int h1_c = r16 >> 2;
  const char* h1_lane = H1 + r16 * 1024 + ((q4 ^ (r16 & 3)) << 4);   // hidden == 512: 1024-byte rows
  constexpr int K3W = 5;  // k-steps of conv3 per wave and chunk: ceil(18 / 4)
  auto load_w2 = [&](int nc, bf16x8f (&fa)[K2S]) {
    const __bf16* wrow = W2 + (long)(nc * 64 + wid * 16 + r16) * Kp2 + q4 * 8;
#pragma unroll
    for (int u = 0; u < K2S; ++u) fa[u] = *reinterpret_cast<const bf16x8f*>(wrow + u * 32);
  };
  auto load_w3 = [&](int nc, bf16x8f (&fa)[K3W][NT3]) {
#pragma unroll
    for (int i = 0; i < K3W; ++i) {
      const int ksi = min(wid + 4 * i, 17), tap = ksi >> 1, hh = ksi & 1;   // clamped: no control flow around loads
#pragma unroll
      for (int nt = 0; nt < NT3; ++nt) {
        const int n = min(nt * 16 + r16, Cout - 1);
        fa[i][nt] = *reinterpret_cast<const bf16x8f*>(W3 + (long)n * Kp3 + tap * hid + nc * 64 + hh * 32 + q4 * 8);
      }
    }
  };
  auto conv2_chunk = [&](int nc, const bf16x8f (&fa)[K2S]) {
    f32x4 acc[MT_H];
#pragma unroll
    for (int mt = 0; mt < MT_H; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8f fb[2][MT_H];
    // h1 image: address(px, slot) = px*1024 + ((slot ^ px) & 15 | slot & ~15) * 16 with slot = 4u + q4.  For this
    // lane px&15 = r16 in every m-tile, so the slot term is ((u ^ (r16>>2)) << 2 | (q4 ^ (r16&3))) : one XOR per
    // k-step; the m-tile stride (16 rows) is a compile-time constant folded into the ds_read offset field.
    auto rd2 = [&](int u, bf16x8f (&f)[MT_H]) {
      const char* p = h1_lane + (((u ^ h1_c) << 2) << 4);
#pragma unroll
      for (int mt = 0; mt < MT_H; ++mt) f[mt] = *reinterpret_cast<const bf16x8f*>(p + mt * 16 * 1024);
    };
    rd2(0, fb[0]);
#pragma unroll
    for (int u = 0; u < K2S; ++u) {
      if (u + 1 < K2S) rd2(u + 1, fb[(u + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mt = 0; mt < MT_H; ++mt)
        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[u], fb[u & 1][mt], acc[mt], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    const int n = nc * 64 + wid * 16 + q4 * 4;
    const f32x4 bb = *reinterpret_cast<const f32x4*>(Pb2 + n), es = *reinterpret_cast<const f32x4*>(Pe2 + n);
    if (nc > 0) __syncthreads();   // every wave is done reading the previous chunk
#pragma unroll
    for (int mt = 0; mt < MT_H; ++mt) {
      f32x4 v;
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = hvalid[mt] ? fmaxf((acc[mt][r] + bb[r]) * es[r], 0.f) : 0.f;  // zero padding of conv3
      const u32x2f pk = {pack2f(v[0], v[1]), pack2f(v[2], v[3])};
      *reinterpret_cast<u32x2f*>(H2c + h2_off(mt * 16 + r16, wid * 16 + q4 * 4)) = pk;
      if (a.h2save && hglob[mt] >= 0) *reinterpret_cast<u32x2f*>(a.h2save + (long)hglob[mt] * hid + n) = pk;
    }
    __syncthreads();
  };
  // rows of W3 beyond Cout were clamped (duplicate data): their accumulators are simply never read back
  auto conv3_chunk = [&](const bf16x8f (&fa)[K3W][NT3]) {
    const bf16x8f zero = __builtin_bit_cast(bf16x8f, u32x4f{0u, 0u, 0u, 0u});
    bf16x8f fb[2][MT_O];
    auto rd3 = [&](int i, bf16x8f (&f)[MT_O]) {
#pragma unroll
      for (int mo = 0; mo < MT_O; ++mo) {
        const bf16x8f v = *reinterpret_cast<const bf16x8f*>(H2c + c3_off[i][mo]);
        f[mo] = ((c3_ok >> (i * MT_O + mo)) & 1u) ? v : zero;
      }
    };
    rd3(0, fb[0]);
#pragma unroll
    for (int i = 0; i < K3W; ++i) {
      if (i + 1 < K3W) rd3(i + 1, fb[(i + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mo = 0; mo < MT_O; ++mo)
#pragma unroll
        for (int nt = 0; nt < NT3; ++nt)
          acc3[nt][mo] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i][nt], fb[i & 1][mo], acc3[nt][mo], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  {
    bf16x8f w2a[K2S], w2b[K2S], w3[K3W][NT3];
    load_w2(0, w2a);
    // nchunks is even (hidden == 512).  Prefetches are UNCONDITIONAL (clamped index): a load under a branch cannot be
    // counted by the compiler's vmcnt bookkeeping and would make every conv2 wait for the prefetch just issued.
    if (a.dbg & 1) { load_w3(0, w3); load_w2(1, w2b); }
    for (int nc = 0; nc < nchunks; nc += 2) {
      if (!(a.dbg & 1)) { load_w3(nc, w3); load_w2(nc + 1, w2b); }   // lands while conv2 of this chunk computes
      conv2_chunk(nc, w2a);
      if (!(a.dbg & 2)) conv3_chunk(w3);
      if (!(a.dbg & 1)) { load_w3(nc + 1, w3); load_w2(min(nc + 2, nchunks - 1), w2a); }
      conv2_chunk(nc + 1, w2b);
      if (!(a.dbg & 2)) conv3_chunk(w3);
    }
  }
  __syncthreads();

  stamp(4);
  // ------------------------------------------------------------------ cross-wave reduction of conv3, then the coupling tail
  float* R3 = reinterpret_cast<float*>(H1);     // [4 waves][64 px][NT3*16]   (h1 is dead now)
  constexpr int NP = NT3 * 16;
#pragma unroll
  for (int nt = 0; nt < NT3; ++nt)
#pragma unroll
    for (int mo = 0; mo < MT_O; ++mo) {
      float* dst = R3 + ((wid * 64 + mo * 16 + r16) * NP + nt * 16 + q4 * 4);
      *reinterpret_cast<f32x4*>(dst) = acc3[nt][mo];
    }
  __syncthreads();
  // one pass per sample of the tile: coupling tail + fixed-order sum of its log-scales
  __shared__ float sh1[32];
  const int per_px = R * W, tile = (TS == 1) ? blockIdx.x % tps : 0;
  for (int sm = 0; sm < TS; ++sm) {
    const int b = b0 + sm;
    float ls = 0.f;
    if (b < s.B) {
      for (int idx = tid; idx < per_px * C2; idx += 256) {
        const int pl = idx / C2, j = idx - pl * C2, rr = pl / W, xx = pl - rr * W;
        const int po = sm * per_px + pl;
        const long m = (long)b * HW + (row0 + rr) * W + xx;
        const float z2 = Zm[((sm * g.RZ + rr + zh) * W + xx) * C + Ch + j];
        auto a3 = [&](int n) {
          const float* p = R3 + po * NP + n;
          return ((p[0] + p[64 * NP]) + (p[2 * 64 * NP] + p[3 * 64 * NP]) + Pb3[n]) * Pe3[n];
        };
        float out;
        if (s.additive) {
          const float h = a3(j);
          if (a.hsave) a.hsave[m * Cout + j] = h;
          out = z2 + h;
        } else {
          const float sh = a3(2 * j), sr = a3(2 * j + 1);
          if (a.hsave) { a.hsave[m * Cout + 2 * j] = sh; a.hsave[m * Cout + 2 * j + 1] = sr; }
          const float t = sr + 2.f;
          out = (z2 + sh) * sigmoidf(t);
          ls += log_sigmoidf(t);
        }
        a.zout[m * C + Ch + j] = out;
      }
      for (int idx = tid; idx < per_px * Ch; idx += 256) {
        const int pl = idx / Ch, j = idx - pl * Ch, rr = pl / W, xx = pl - rr * W;
        a.zout[((long)b * HW + (row0 + rr) * W + xx) * C + j] = Zm[((sm * g.RZ + rr + zh) * W + xx) * C + j];
      }
    }
    ls = block_sum(ls, sh1);
    if (tid == 0 && b < s.B) a.ldpart[(long)b * tps + tile] = ls + (tile == 0 ? s.dld[0] : 0.f);
  }
  stamp(5);
}

// ld_out[b] = ld_in[b] + sum over steps and tiles of ldpart[step][b][tile], in fixed order
__global__ void ld_finish_kernel(const float* __restrict__ ld_in, const float* __restrict__ ldpart, int nsteps, int B,
                                 int tps, float* __restrict__ ld_out) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float v = ld_in[b];
  for (int st = 0; st < nsteps; ++st)
    for (int t = 0; t < tps; ++t) v += ldpart[((long)st * B + b) * tps + t];
  ld_out[b] = v;
}

static unsigned long long* g_stamps = nullptr;
static int g_dbg = 0;
static int g_force_fused = 0;
static int g_no_v2 = 0;
static int grid_of(const nfk_step2d& s) {   // 64 output pixels per workgroup
  const int HW = s.H * s.W;
  return HW <= 64 ? (s.B * HW + 63) / 64 : s.B * (HW / 64);
}
static bool fused_pick(const nfk_step2d& s, int* R, int* TS, int* halo) {
  if (!s.bf16 || s.hidden != 512 || s.Cout > 48 || (s.C & 3) != 0) return false;
  const int HW = s.H * s.W;
  if (HW <= 64 && 64 % HW == 0) { *R = s.H; *TS = 64 / HW; *halo = 0; }
  else if (s.W <= 64 && 64 % s.W == 0 && s.H % (64 / s.W) == 0) { *R = 64 / s.W; *TS = 1; *halo = 1; }
  else return false;
  const FusedGeom g = fused_geom(s, *R, *TS, *halo);
  // measured (CIFAR, B=64): the fused step wins where it fills the chip (level 1: 256 workgroups, 55 vs 77 us);
  // with 64 / 16 workgroups (levels 2 / 3) the five-kernel path is still faster, so it is kept there
  const int grid = *TS == 1 ? s.B * (s.H / *R) : (s.B + *TS - 1) / *TS;
  if (grid < 128 && !g_force_fused) return false;
  if (g.MT_h != 4 && g.MT_h != 6) return false;
  if (g.K1 > 256) return false;   // conv1 keeps at most 8 k-steps of weight fragments per n-tile in registers
  if (g.total > 160 * 1024 - 512) return false;
  if ((size_t)4 * 64 * g.NT3 * 16 * 4 > (size_t)g.MT_h * 16 * s.hidden * 2) return false;  // R3 must fit in the h1 region
  return true;
}

template <int MT_H, int NT3, int K1M>
static int launch_fused(const FusedArgs& a, const FusedGeom& g, int grid, hipStream_t st) {
  static bool attr = false;
  if (!attr) {
    // 160 KiB per workgroup in total; the kernel also owns a few hundred bytes of static LDS
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(fused_step_fwd_kernel<MT_H, NT3, K1M>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512) != hipSuccess)
      return NFK_ERR_LAUNCH;
    attr = true;
  }
  hipLaunchKernelGGL((fused_step_fwd_kernel<MT_H, NT3, K1M>), dim3(grid), dim3(256), g.total, st, a);
  NFK_LAUNCH_CHECK();
  return NFK_OK;
}

template <int MT_H, int NT3>
static int dispatch_k1(const FusedArgs& a, const FusedGeom& g, int grid, hipStream_t st) {
  switch (g.K1p) {
    case 64: return launch_fused<MT_H, NT3, 2>(a, g, grid, st);
    case 128: return launch_fused<MT_H, NT3, 4>(a, g, grid, st);
    default: return launch_fused<MT_H, NT3, 8>(a, g, grid, st);
  }
}
template <int MT_H>
static int dispatch_nt3(const FusedArgs& a, const FusedGeom& g, int grid, hipStream_t st) {
  switch (g.NT3) {
    case 1: return dispatch_k1<MT_H, 1>(a, g, grid, st);
    case 2: return dispatch_k1<MT_H, 2>(a, g, grid, st);
    default: return dispatch_k1<MT_H, 3>(a, g, grid, st);
  }
}

}  // namespace nfk

using namespace nfk;
extern "C" {

void nfk_fused_set_stamps(void* buf) { g_stamps = reinterpret_cast<unsigned long long*>(buf); }
void nfk_fused_set_split(int ways) { fused2_set_split(ways >= 4 ? 1 : (ways >= 2 ? 2 : 0)); }
void nfk_fused_set_debug(int bits) {
  g_dbg = bits & 0xff; g_force_fused = (bits >> 8) & 1; g_no_v2 = (bits >> 9) & 1;
  fused2_set_split(((bits >> 10) & 1) ? 0 : (((bits >> 11) & 1) ? 2 : 1));   // bit 10: no hidden split, bit 11: 2-way instead of 4-way
}

static int v2_tiles(const nfk_step2d& s) {
  if (g_no_v2 || !s.bf16) return 0;
  // measured (CIFAR, B=64, teacher K=32): generation 2 beats the five-kernel path at every level, also where it
  // fills only 64 / 16 CUs (38 / 37 / 49 us per step against 76 / 50 / 54)
  return fused2_tiles(s);
}

int nfk_step2d_fused_tiles(const nfk_step2d* s) {
  if (const int t2 = v2_tiles(*s)) return t2;
  int R, TS, halo;
  if (!fused_pick(*s, &R, &TS, &halo)) return 0;
  return TS == 1 ? s->H / R : 1;   // log-det partial slots per sample
}

static int step2d_fwd_fused_impl(const nfk_step2d* s, const float* x, int ldx, float* zsave, void* h1save,
                                 void* h2save, float* hsave, float* zout, float* ldpart, float* ws, void* masksave,
                                 void* stream);

size_t nfk_step2d_fused_ws_floats(const nfk_step2d* s) { return v2_tiles(*s) ? fused2_ws_floats(*s) : 0; }

int nfk_step2d_fwd_fused(const nfk_step2d* s, const float* x, int ldx, float* zsave, void* h1save, void* h2save,
                         float* hsave, float* zout, float* ldpart, float* ws, void* masksave, void* stream) {
  // algorithmic work of one FlowStep forward (bench.py roofline leg): the three convs' FLOPs; HBM bytes = the step's
  // input and output, its bf16 weights once, and what a training forward keeps for the backward
  const double M = (double)s->B * s->H * s->W, hid = s->hidden, K1 = 9.0 * (s->C / 2);
  const double flops = 2.0 * M * (K1 * hid + hid * hid + 9.0 * hid * s->Cout);
  double bytes = 8.0 * M * s->C + 2.0 * (K1 * hid + hid * hid + 9.0 * hid * s->Cout);
  if (zsave) bytes += 4.0 * M * s->C;
  if (h1save) bytes += 2.0 * M * hid;
  if (h2save) bytes += 2.0 * M * hid;
  if (hsave) bytes += 4.0 * M * s->Cout;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  int cls = NFK_PROF_FUSED_STEP + (s->Cout + 15) / 16 - 1;
  if (cls > NFK_PROF_FUSED_STEP_NT3) cls = NFK_PROF_FUSED_STEP_NT3;
  prof_scope_begin(cls, flops, bytes, st);
  const int rc = step2d_fwd_fused_impl(s, x, ldx, zsave, h1save, h2save, hsave, zout, ldpart, ws, masksave, stream);
  prof_scope_end(cls, st);
  return rc;
}

static int step2d_fwd_fused_impl(const nfk_step2d* s, const float* x, int ldx, float* zsave, void* h1save,
                                 void* h2save, float* hsave, float* zout, float* ldpart, float* ws, void* masksave,
                                 void* stream) {
  if (masksave && !(v2_tiles(*s) && h1save && h2save)) return NFK_ERR_ARG;   // only the generation-2 kernel writes masks
  if (v2_tiles(*s)) {
    FusedCall c{};
    c.s = *s; c.x = x; c.ldx = ldx; c.zsave = zsave; c.h1save = h1save; c.h2save = h2save; c.hsave = hsave;
    c.masksave = reinterpret_cast<unsigned char*>(masksave);
    c.zout = zout; c.ldpart = ldpart; c.ws = ws; c.stamps = g_stamps; c.dbg = g_dbg;
    return fused2_launch(c, reinterpret_cast<hipStream_t>(stream));
  }
  FusedArgs a{};
  if (!fused_pick(*s, &a.R, &a.TS, &a.halo)) return NFK_ERR_ARG;
  a.s = *s; a.x = x; a.ldx = ldx; a.zsave = zsave;
  a.h1save = reinterpret_cast<__bf16*>(h1save); a.h2save = reinterpret_cast<__bf16*>(h2save);
  a.hsave = hsave; a.zout = zout; a.ldpart = ldpart; a.stamps = g_stamps; a.dbg = g_dbg;
  const FusedGeom g = fused_geom(*s, a.R, a.TS, a.halo);
  const int grid = a.TS == 1 ? s->B * (s->H / a.R) : (s->B + a.TS - 1) / a.TS;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  return g.MT_h == 4 ? dispatch_nt3<4>(a, g, grid, st) : dispatch_nt3<6>(a, g, grid, st);
}

int nfk_ld_finish(const float* ld_in, const float* ldpart, int nsteps, int B, int tps, float* ld_out, void* stream) {
  if (B <= 0) return NFK_OK;
  hipLaunchKernelGGL(ld_finish_kernel, dim3((B + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), ld_in,
                     ldpart, nsteps, B, tps, ld_out);
  NFK_LAUNCH_CHECK();
  return NFK_OK;
}
}
