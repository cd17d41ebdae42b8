// nfk fused FlowStep forward (bf16 operands / fp32 accumulate): ONE launch per FlowStep
// (models/flows.py:142-171: ActNorm -> invertible 1x1 conv -> affine/additive coupling with the 3-conv net
// models/flows.py:17-60).
//
//   workgroup = a spatial tile of 64 output pixels (whole image rows, +1 halo row each side when the tile is not
//   a whole image): CIFAR level 1: 4 rows of one image, level 2: one 8x8 image, level 3: four 4x4 images.
//   1. mix      z = Wf x + cf on the tile rows (+2 halo rows)                              -> LDS (fp32)
//   2. im2col   3x3 taps of z1 on the h-region (tile + 1 halo row)                          -> LDS (bf16)
//   3. conv1    3x3 C/2->512, ActNorm, ReLU.  32x32x16 MFMA: every wave owns 4 tiles of 32 output channels and
//               keeps their W1 rows in registers from the start of the kernel               -> h1 in LDS (bf16)
//   4. conv2    1x1 512->512 in ONE pass over k: every wave owns 4 tiles of 32 output channels (12 accumulator tiles
//               of 32x32 at 96 pixels) so that a B fragment (32 pixels x 16 k of h1) read from LDS feeds four MFMAs per
//               pixel tile.  A fragments (W2 rows) stream from L2 through a 32-fragment register ring, each slot
//               reloaded in place 8 k-steps ahead.  h2 = ReLU(ActNorm(.)) then REPLACES h1 in the same LDS image
//   5. conv3    3x3 512->Cout over the h2 image (9 taps = shifted LDS rows): per 128-channel chunk, wave w takes the
//               chunk's w-th 32 channels for all 9 taps; W3 fragments sit in a register ring up to four chunks deep,
//               requested before conv2's epilogue; the waves' sums are reduced through LDS at the end
//   6. tail     bias, exp(3 logs), sigmoid(.+2), (z2+shift)*scale, per-tile partial sum of log-scales
//   h1 / h2 never travel through HBM between the convolutions; when a backward will follow they are copied out
//   once, from LDS, in whole 16-byte pieces of contiguous rows.
// No integer division on the device: W, H and the tile sizes are powers of two (shifts); divisions by C and its
// relatives use a multiply-high with a reciprocal computed once per kernel.
#include "nfk_internal.h"

namespace nfk {

typedef __bf16 bf16x8f __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4f __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2f __attribute__((ext_vector_type(2)));

struct F2Args {
  FusedCall c;           // one step: the call.  A run of steps (MULTI): c.s = the first step (dimensions), c.x / c.ldx = the
                         // run's input, c.stamps; everything else comes from `run`
  int R, TS, halo;       // rows per tile, samples per tile, 1 if the tile needs halo rows
  int lw, lrw, lhw;      // log2(W), log2(R*W), log2(RH*W) (the last only meaningful when TS > 1)
  nfk_fused_run run;     // MULTI only
};

struct F2Geom {
  int RH, RZ, P_h, MT32, K1, K1M, K1p, X1S, NT3;
  size_t oZm, oWf, oPar, oX1, oH1, oZero, total;
};

__host__ __device__ static inline F2Geom f2_geom(const nfk_step2d& s, int R, int TS, int halo) {
  F2Geom g;
  const int W = s.W, C = s.C, Ch = C / 2;
  g.RH = R + 2 * halo;
  g.RZ = R + 4 * halo;
  g.P_h = TS * g.RH * W;
  g.MT32 = (g.P_h + 31) / 32;
  g.K1 = 9 * Ch;
  g.K1M = g.K1 <= 64 ? 4 : (g.K1 <= 128 ? 8 : 16);   // conv1 k-steps of 16 (zero columns pad the im2col tile)
  g.K1p = g.K1M * 16;
  g.X1S = g.K1p * 2 + 16;                            // bytes per im2col row: an odd number of 16-byte slots
  g.NT3 = (s.Cout + 15) / 16;
  size_t o = 0;
  g.oZm = o; o += (size_t)TS * g.RZ * W * C * 4;
  o = (o + 15) & ~size_t(15);
  g.oWf = o; o += (size_t)(C * ((C + 3) & ~3) + C) * 4;
  o = (o + 15) & ~size_t(15);
  g.oPar = o; o += (size_t)(4 * s.hidden + 2 * ((s.Cout + 3) & ~3)) * 4;
  g.oX1 = o; o += (size_t)g.MT32 * 32 * g.X1S;
  o = (o + 15) & ~size_t(15);
  g.oH1 = o; o += (size_t)g.MT32 * 32 * 1024;
  g.oZero = o; o += 3 * 256 + 16;   // sixteen zero bytes per 128-channel chunk, 256 bytes apart (the chunks' pitch in
                                    // the h2 image): conv3's zero padding reads them instead of masking fragments
  g.total = o;
  return g;
}

__device__ __forceinline__ unsigned f2_pack(float lo, float hi) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  typedef float f32x2p __attribute__((ext_vector_type(2)));
  const f32x2p f = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2));
}
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
// two fp32 -> packed bf16 pair with ReLU applied on the packed halves (sign bit set <=> negative as int16)
__device__ __forceinline__ unsigned f2_relu_pack(f32x2 v) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  const bf16x2 b = __builtin_convertvector(v, bf16x2);   // one v_cvt_pk_bf16_f32
  s16x2 h = __builtin_bit_cast(s16x2, b);
  const s16x2 z = {0, 0};
  h = __builtin_elementwise_max(h, z);
  return __builtin_bit_cast(unsigned, h);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t f2_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
// 16-byte weight fragment: lane-constant voff + scalar soff (+ a compile-time constant folded into the instruction)
__device__ __forceinline__ bf16x8f f2_ldw(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  return __builtin_bit_cast(bf16x8f, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
// scale = sigmoid(t) and log(scale) from ONE exponential, one reciprocal and one logarithm (hardware v_exp / v_rcp / v_log):
//   t >= 0: u = e^-t in (0, 1]:  scale = 1 / (1 + u),  log scale = -log(1 + u)
//   t <  0: u = e^t  in (0, 1):  scale = u / (1 + u),  log scale = t - log(1 + u)
// 1 + u lies in [1, 2], so log(1 + u) carries an ABSOLUTE error of ~1e-7 -- against per-sample sums of thousands of such
// terms of size ~0.1 - 1 that is the rounding noise of the sum itself.  (The library log1pf / expf pair this replaces was
// ~10x the instructions and half of the tail's time.)
__device__ __forceinline__ void f2_sigmoid_logsigmoid(float t, float& sg, float& lsg) {
  const float u = __expf(-fabsf(t));
  const float r = __frcp_rn(1.f + u);
  const float l = __logf(1.f + u);
  sg = t >= 0.f ? r : u * r;
  lsg = fminf(t, 0.f) - l;
}
// h1 image: 1024-byte rows (512 channels); the low 4 bits of the 16-byte slot index are XORed with the pixel index
__device__ __forceinline__ int f2_h1_off(int px, int slot) {
  return px * 1024 + ((((slot ^ px) & 15) | (slot & ~15)) << 4);
}

// NS > 1 ("hidden split", for grids that would fill only 64 or 16 CUs): NS workgroups share a pixel tile; each runs
// the whole mix / im2col / conv1 but only its 4/NS chunks of conv2 + conv3, and leaves its partial conv3 sums in slab
// q of a.c.ws; coupling_fwd_kernel (S = NS slabs) then finishes the step.  The conv2 / conv3 weight stream, which is
// what a lone workgroup spends its time on, is divided by NS.
//
// MULTI ("run of steps"): ONE launch walks all K FlowSteps of a level for its pixel tile.  Only for tiles that are whole
// images (no halo rows: CIFAR levels 2 and 3), where a step's output tile is exactly the next step's input tile: z stays
// in LDS between the steps (it is also written to HBM for the steps whose output somebody reads: `run.keep`, and for all
// of them in a training forward), the next step's parameters and conv1 weights are requested before the current step's
// tail, and a level costs one launch instead of K (the teacher's levels 2 / 3: 64 launches -> 2).  Per-step descriptors
// (nfk_step2d) are read from a device array with scalar loads; per-step activation buffers are base + k * stride.
// The arithmetic of every step is the single-step kernel's, instruction for instruction: results are bit-identical.
// MO: output pixels per tile in units of 16 -- 4 (64 pixels, the default) or, for runs of steps on tiny images (CIFAR level
// 3: 4x4 images, 16 tiles of 64 pixels at B = 64), 2: tiles of 32 pixels = two images, twice the workgroups, each with half
// the mix / im2col / epilogue / tail work and half the MFMAs; only the weight stream per workgroup stays.
template <int MT32, int NT3, int K1M, int NS, bool MULTI, int MO = 4>
__global__ void __launch_bounds__(256) fused2_step_fwd_kernel(F2Args a) {
  static_assert(!MULTI || NS == 1, "a run of steps has no hidden split");
  static_assert(MO == 4 || ((MO == 2 || MO == 1) && MULTI && MT32 == 1) || (MO == 2 && MT32 == 3 && !MULTI && NS == 1),
                "small tiles: runs of steps on whole tiny images, or ONE 32-pixel image row (+ a halo row each side) per tile");
  constexpr int NPX = 16 * MO;   // output pixels of a tile
  typedef const __attribute__((address_space(4))) nfk_step2d CStep;   // scalar (constant-memory) loads of a descriptor
#define F2S(p, f) (MULTI ? (p)->f : a.c.s.f)
  constexpr int HID = 512;
  constexpr int MT_O = MO;  // output pixels = MO tiles of 16 (conv3 uses 16x16x32)
  extern __shared__ __attribute__((aligned(16))) char smem_f2[];
  const nfk_step2d& s = a.c.s;   // dimensions (the same for every step of a run); per-step POINTERS go through F2S
  // registers that live from one step of a run to the next: the per-channel parameters of the NEXT step, requested
  // before the current step's tail
  constexpr int W1R = (K1M >= 16) ? 2 : 4;  // conv1's register ring: tiles of 32 output channels held ahead
  constexpr int XMAX = 12, WMAX = 9;   // per-thread slots: TS*RZ*W*C <= 3072, C*C <= 2304 (checked on the host)
  float wreg[WMAX], cfr, preg[2][4], b3r, e3r;
  const int nst = MULTI ? a.run.nsteps : 1;
  for (int st = 0; st < nst; ++st) {
  // Everything below is per step.  In a run of steps the thread index and the tile dimensions are passed through an
  // empty asm so that the compiler computes what derives from them (addresses, masks, descriptors) once per step, where
  // it is used, as in the one-step kernel -- hoisted out of the step loop they would all be live across it (hundreds of
  // registers, spilled)
  int tid_ = threadIdx.x, W_ = s.W, H_ = s.H, C_ = s.C, Cout_ = s.Cout, R_ = a.R, TS_ = a.TS;
  if (MULTI) {
    asm volatile("" : "+v"(tid_));
    asm volatile("" : "+s"(W_), "+s"(H_), "+s"(C_), "+s"(Cout_), "+s"(R_), "+s"(TS_));
  }
  const int W = W_, H = H_, C = C_, Ch = C / 2, C2 = C - Ch, Cout = Cout_, HW = H * W;
  const int R = R_, TS = TS_, halo = a.halo, lw = a.lw, lrw = a.lrw, lhw = a.lhw;
  nfk_step2d sd;   // the dimensions f2_geom reads
  sd.W = W; sd.H = H; sd.C = C; sd.Cout = Cout; sd.hidden = HID; sd.B = s.B;
  const F2Geom g = f2_geom(sd, R, TS, halo);
  float* Zm = reinterpret_cast<float*>(smem_f2 + g.oZm);
  float* Wl = reinterpret_cast<float*>(smem_f2 + g.oWf);
  char* X1 = smem_f2 + g.oX1;
  char* H1 = smem_f2 + g.oH1;
  const int tid = tid_, lane = tid & 63, wid = tid >> 6;
  const int l32 = lane & 31, hi = lane >> 5;      // 32x32x16 operand roles: row / k-half
  const int r16 = lane & 15, q4 = lane >> 4;      // 16x16x32 operand roles (conv3)

  const int tps = (TS == 1) ? H / R : 1;
  const int wg = (NS == 1) ? (int)blockIdx.x : (int)blockIdx.x / NS, qs = (NS == 1) ? 0 : (int)blockIdx.x % NS;
  const int b0 = (TS == 1) ? wg / tps : wg * TS;
  const int tile = (TS == 1) ? wg - b0 * tps : 0;
  const int row0 = tile * R;
  const int zh = 2 * halo;
  auto stamp = [&](int i) {
    if (a.c.stamps && tid == 0 && (!MULTI || st == nst - 1))   // a run: the phases of its last step
      a.c.stamps[(long)blockIdx.x * 16 + i] = __builtin_readcyclecounter();
  };
  stamp(0);

  // Weights come in MFMA-fragment order (nfk_repack_desc.wf): fragment f of an array is the 1 KiB at f*1024, lane l
  // owns bytes [16 l, 16 l + 16): every wave-load is one contiguous KiB (8 whole cache lines) at a scalar offset.
  const int swid = __builtin_amdgcn_readfirstlane(wid);
  const int voffL = lane * 16;
  const int Kp1 = (g.K1 + 63) / 64 * 64;   // padded K of conv1 (== K1M * 16, checked on the host)
  bf16x8f w1[W1R][K1M];
  auto load_w1 = [&](const __amdgpu_buffer_rsrc_t& r, int j, bf16x8f (&f)[K1M]) {   // row tile (32 channels) swid*4 + j
    const int soff = (swid * 4 + j) * K1M * 1024;
#pragma unroll
    for (int ks = 0; ks < K1M; ++ks) f[ks] = f2_ldw(r, voffL, soff + ks * 1024);
  };
  const int Cp = (C + 3) & ~3;
  const unsigned mC = 0xFFFFFFFFu / (unsigned)C + 1u;   // idx / C == umulhi(idx, mC) for the small idx used here
  const int rowlen = W * C, per_s = g.RZ * rowlen, nx = TS * per_s;
  float* Pb1 = reinterpret_cast<float*>(smem_f2 + g.oPar);
  float* Pe1 = Pb1 + HID;
  float* Pb2 = Pe1 + HID;
  float* Pe2 = Pb2 + HID;
  float* Pb3 = Pe2 + HID;
  float* Pe3 = Pb3 + ((Cout + 3) & ~3);
  constexpr int NP = NT3 * 16;
  // raw x rows of the tile (+halo), staged in the h1 region, which is free at that time.  In a run of steps the same
  // place receives the step's OUTPUT tile (= the next step's x) from the coupling tail, while the front of the region
  // still holds the cross-wave sums the tail reads (R3): it then starts behind those.
  float* Xraw = reinterpret_cast<float*>(H1 + (MULTI ? 4 * NPX * NP * 4 : 0));
  // per-channel parameters of a step: global -> registers (load_params) -> LDS (store_params).  A run of steps loads
  // the NEXT step's before the current step's tail and stores them behind it.
  auto load_params = [&](CStep* p) {
    const float *Wfp = F2S(p, Wf), *cfp = F2S(p, cf), *b1p = F2S(p, b1), *e1p = F2S(p, es1), *b2p = F2S(p, b2),
                *e2p = F2S(p, es2), *b3p = F2S(p, bias3), *e3p = F2S(p, es3);
#pragma unroll
    for (int i = 0; i < WMAX; ++i) wreg[i] = Wfp[min(tid + 256 * i, C * C - 1)];
    cfr = cfp[min(tid, C - 1)];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = tid + 256 * i;
      preg[i][0] = b1p[idx]; preg[i][1] = e1p[idx]; preg[i][2] = b2p[idx]; preg[i][3] = e2p[idx];
    }
    b3r = b3p[min(tid, Cout - 1)]; e3r = e3p[min(tid, Cout - 1)];
  };
  auto store_params = [&]() {
#pragma unroll
    for (int i = 0; i < WMAX; ++i) {
      const int idx = tid + 256 * i;
      if (idx < C * C) {
        const int o = (int)__umulhi((unsigned)idx, mC), c = idx - o * C;
        Wl[c * Cp + o] = wreg[i];
      }
    }
    if (tid < C) Wl[C * Cp + tid] = cfr;
#pragma unroll
    for (int i = 0; i < 2; ++i) {   // (a + b) * e is evaluated as fma(a, e, b*e)
      const int idx = tid + 256 * i;
      Pb1[idx] = preg[i][0] * preg[i][1]; Pe1[idx] = preg[i][1];
      Pb2[idx] = preg[i][2] * preg[i][3]; Pe2[idx] = preg[i][3];
    }
    if (tid < Cout) { Pb3[tid] = b3r; Pe3[tid] = e3r; }
  };
  if (tid < 16) reinterpret_cast<unsigned*>(smem_f2 + g.oZero + (tid >> 2) * 256)[tid & 3] = 0u;

  CStep* sp = MULTI ? (CStep*)(a.run.steps + st) : nullptr;
  // this step's buffers: the call's (one step) or base + st * stride (a run)
  float* const c_zsave = MULTI ? (a.run.zsave ? a.run.zsave + (long)st * a.run.zsave_stride : nullptr) : a.c.zsave;
  __bf16* const c_h1save = MULTI ? (a.run.h1save ? reinterpret_cast<__bf16*>(a.run.h1save) + (long)st * a.run.h_stride : nullptr)
                                 : reinterpret_cast<__bf16*>(a.c.h1save);
  __bf16* const c_h2save = MULTI ? (a.run.h2save ? reinterpret_cast<__bf16*>(a.run.h2save) + (long)st * a.run.h_stride : nullptr)
                                 : reinterpret_cast<__bf16*>(a.c.h2save);
  float* const c_hsave = MULTI ? (a.run.hsave ? a.run.hsave + (long)st * a.run.hsave_stride : nullptr) : a.c.hsave;
  unsigned char* const c_masksave =
      MULTI ? (a.run.masksave ? reinterpret_cast<unsigned char*>(a.run.masksave) + (long)st * a.run.mask_stride : nullptr)
            : a.c.masksave;
  float* const c_zout =
      MULTI ? (((a.run.keep >> st) & 1ull)
                   ? a.run.zout + (long)__builtin_popcountll(a.run.keep & ((1ull << st) - 1ull)) * a.run.zout_stride
                   : nullptr)
            : a.c.zout;
  float* const c_ldpart = MULTI ? a.run.ldpart + (long)st * s.B * tps : a.c.ldpart;
  const __amdgpu_buffer_rsrc_t r1 = f2_rsrc(F2S(sp, wf1), (unsigned)HID * Kp1 * 2);

  // ------------------------------------------------------------------ 0./1. conv1 weights -> registers, mix
  // Every global load of the prologue is issued before the first wait (x first: it is needed first and the vector
  // memory returns in order); each of these is an L2 miss, and taken one loop after the other they cost a miss each.
  if (!MULTI || st == 0) {
    // the tile's rows (+halo) of all its samples are one run of pixels in x: element e = (pixel, channel)
    const int lo = (TS == 1) ? max(0, zh - row0) * rowlen : 0;                         // elements inside the image
    const int hi_e = (TS == 1) ? min(g.RZ, H - row0 + zh) * rowlen : min(s.B - b0, TS) * per_s;   // ... and the batch
    const float* xb = a.c.x + ((long)b0 * HW + (long)(row0 - zh) * W) * a.c.ldx;
    float xreg[XMAX];
#pragma unroll
    for (int i = 0; i < XMAX; ++i) {
      const int e = min(max(tid + 256 * i, lo), hi_e - 1);   // clamped: the load itself is unconditional
      const int pix = (int)__umulhi((unsigned)e, mC), ch = e - pix * C;
      xreg[i] = xb[(long)pix * a.c.ldx + ch];
    }
    load_params(sp);
    // conv1's weight fragments travel during the mix and the im2col; their requests (up to 32 KiB per wave: ~2 k cycles
    // of issue for the four waves) go out while x and the parameters are still on their way
#pragma unroll
    for (int j = 0; j < W1R; ++j) load_w1(r1, j, w1[j]);
#pragma unroll
    for (int i = 0; i < XMAX; ++i) {
      const int e = tid + 256 * i;
      if (e < nx) Xraw[e] = (e >= lo && e < hi_e) ? xreg[i] : 0.f;
    }
    store_params();
  } else {
    store_params();   // requested before the previous step's tail
#pragma unroll
    for (int j = 0; j < W1R; ++j) load_w1(r1, j, w1[j]);
  }
  stamp(8);
  __syncthreads();
  stamp(9);
  {
    // z = Wf x + cf per pixel; thread = (2 neighbouring pixels, 4 outputs).  This phase is bound by LDS bandwidth (every
    // thread streams its 4-column strip of Wf: 64 B of weights + 16 B of x per 16 FMAs with one pixel per thread, all four
    // waves on one LDS port), so a thread takes two pixels per strip: 96 B per 32 FMAs.  C % 4 == 0 and W even (host
    // checks): the two pixels share their row and sample.  Four input channels per round (6 LDS reads, then 32 FMAs);
    // the FMA order per output is input channels ascending, as in every other kernel of the step.
    const int G4 = Cp / 4, npp = (TS * g.RZ * W) >> 1;
    const unsigned mG = 0xFFFFFFFFu / (unsigned)G4 + 1u;
    const int lzw = lhw;   // TS > 1 implies halo == 0: RZ == RH
    for (int t = tid; t < npp * G4; t += 256) {
      const int pp = (int)__umulhi((unsigned)t, mG), o = (t - pp * G4) * 4, pz = 2 * pp;
      const int sm = (TS == 1) ? 0 : (pz >> lzw), rem = (TS == 1) ? pz : (pz & ((1 << lzw) - 1));
      const int rz = rem >> lw, xx = rem & (W - 1);
      const int y = row0 + rz - zh, b = b0 + sm;
      f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
      const bool ok = (unsigned)y < (unsigned)H && b < s.B;
      if (ok) {
        const float* xr = Xraw + pz * C;
        const float* wl = Wl + o;
        acc0 = acc1 = *reinterpret_cast<const f32x4*>(wl + C * Cp);
#pragma unroll 2
        for (int c = 0; c < C; c += 4) {
          const f32x4 xa = *reinterpret_cast<const f32x4*>(xr + c), xb = *reinterpret_cast<const f32x4*>(xr + C + c);
          const f32x4 w0 = *reinterpret_cast<const f32x4*>(wl + c * Cp);
          const f32x4 w1 = *reinterpret_cast<const f32x4*>(wl + (c + 1) * Cp);
          const f32x4 w2 = *reinterpret_cast<const f32x4*>(wl + (c + 2) * Cp);
          const f32x4 w3 = *reinterpret_cast<const f32x4*>(wl + (c + 3) * Cp);
          acc0 += w0 * xa[0]; acc1 += w0 * xb[0];
          acc0 += w1 * xa[1]; acc1 += w1 * xb[1];
          acc0 += w2 * xa[2]; acc1 += w2 * xb[2];
          acc0 += w3 * xa[3]; acc1 += w3 * xb[3];
        }
      }
      *reinterpret_cast<f32x4*>(Zm + pz * C + o) = acc0;
      *reinterpret_cast<f32x4*>(Zm + pz * C + C + o) = acc1;
      if (ok && c_zsave && qs == 0 && rz >= zh && rz < zh + R) {
        float* zp = c_zsave + ((long)b * HW + y * W + xx) * C + o;
        *reinterpret_cast<f32x4*>(zp) = acc0;
        *reinterpret_cast<f32x4*>(zp + C) = acc1;
      }
    }
  }
  stamp(10);
  __syncthreads();
  stamp(1);

  // ------------------------------------------------------------------ 2. im2col of z1 on the h-region (bf16 pairs)
  {
    // thread item = (h-region pixel, tap, group of 4 channels of z1): one 16-byte LDS read -> two packed bf16 pairs.
    // Items are taken four at a time per thread, all four reads before the first write, so that a thread waits for the
    // LDS once per four items (the dependent read -> pack -> write chain per item is what this phase costs).  Taps
    // outside the image read the zero slot (conv3's, 16 zero bytes) instead of branching.
    const int npad = (g.K1p - g.K1) >> 1;   // zero pairs behind the last tap (K1 = 9*Ch is even)
    for (int t = tid; t < MT32 * 32 * npad; t += 256) {
      const int ph = (int)__umulhi((unsigned)t, 0xFFFFFFFFu / (unsigned)npad + 1u), e = t - ph * npad;
      reinterpret_cast<unsigned*>(X1 + ph * g.X1S + g.K1 * 2)[e] = 0u;
    }
    const int G = (Ch + 3) >> 2, per_ph = 9 * G, nitems = MT32 * 32 * per_ph;
    const unsigned mPP = 0xFFFFFFFFu / (unsigned)per_ph + 1u, mGG = 0xFFFFFFFFu / (unsigned)G + 1u;
    const float* zslot = reinterpret_cast<const float*>(smem_f2 + g.oZero);
    for (int base = tid; base < nitems; base += 4 * 256) {
      f32x4 v[4];
      unsigned* d[4];
      int np[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int t = min(base + u * 256, nitems - 1);
        const int ph = (int)__umulhi((unsigned)t, mPP), r = t - ph * per_ph;
        const int tap = (int)__umulhi((unsigned)r, mGG), c0 = (r - tap * G) * 4;
        const int sm = (TS == 1) ? 0 : (ph >> lhw), rem = (TS == 1) ? ph : (ph & ((1 << lhw) - 1));
        const int rr = rem >> lw, xx = rem & (W - 1);
        const int t3 = (tap * 11) >> 5;               // tap / 3 for tap < 9
        const int dy = t3 - 1, dx = tap - t3 * 3 - 1;
        const int rz = rr + halo + dy, xn = xx + dx;  // Zm has `zh` halo rows, the h-region `halo`
        const bool ok = ph < g.P_h && (unsigned)rz < (unsigned)g.RZ && (unsigned)xn < (unsigned)W;
        const float* src = ok ? Zm + ((sm * g.RZ + rz) * W + xn) * C + c0 : zslot;   // 16-byte aligned: C % 4 == 0
        v[u] = *reinterpret_cast<const f32x4*>(src);   // may run 2 floats into z2 (C/2 % 4 == 2): not written below
        d[u] = reinterpret_cast<unsigned*>(X1 + ph * g.X1S + (tap * Ch + c0) * 2);
        np[u] = (base + u * 256 < nitems) ? ((c0 + 2 < Ch) ? 2 : 1) : 0;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (np[u] > 0) d[u][0] = f2_pack(v[u][0], v[u][1]);
        if (np[u] > 1) d[u][1] = f2_pack(v[u][2], v[u][3]);
      }
    }
  }
  __syncthreads();
  stamp(2);
  if (c_masksave && qs == 0) {
    // training forward: the im2col rows of the tile's own pixels (bf16 [M][K1p], behind the ReLU masks in `masksave`) are
    // the Q operand of conv1's weight gradient -- a plain pixel-major GEMM operand instead of a 9-tap gather from z
    __bf16* x1g = reinterpret_cast<__bf16*>(c_masksave + (long)2 * s.B * HW * (HID / 8));
    constexpr int PPR = K1M * 2;   // 16-byte pieces per row
    for (int p = tid; p < NPX * PPR; p += 256) {
      const int po = p / PPR, slot = p % PPR;
      const int sm = po >> lrw, rem = po & ((1 << lrw) - 1), rr = rem >> lw, xx = rem & (W - 1);
      const int b = b0 + sm;
      if (b >= s.B) continue;
      const int ph = sm * g.RH * W + (rr + halo) * W + xx;
      const long m = (long)b * HW + (row0 + rr) * W + xx;
      *reinterpret_cast<u32x4f*>(x1g + m * (K1M * 16) + slot * 8) = *reinterpret_cast<const u32x4f*>(X1 + ph * g.X1S + slot * 16);
    }
  }

  // this lane's h-region pixel in each 32-pixel tile: inside the image?  (conv3's zero padding is applied to h2)
  unsigned hmask[MT32];
#pragma unroll
  for (int mt = 0; mt < MT32; ++mt) {
    const int ph = mt * 32 + l32;
    const int sm = (TS == 1) ? 0 : (ph >> lhw), rem = (TS == 1) ? ph : (ph & ((1 << lhw) - 1));
    const int rr = rem >> lw;
    const int y = row0 + rr - halo, b = b0 + sm;
    hmask[mt] = (ph < g.P_h && (unsigned)y < (unsigned)H && b < s.B) ? 0xFFFFFFFFu : 0u;
  }
  // coalesced copy of the 64 output pixels' rows (16-byte slots slot0 .. slot0 + 2^slots_log2 - 1) of the LDS image to
  // columns col0.. of a global [M][512] bf16 array
  // `mask` (optional): [M][HID/8] bytes, bit c%8 of byte c/8 = (h[m][c] > 0) -- what the fused step backward needs of h
  auto save_rows = [&](__bf16* dstg, int col0, const char* img, int slots_log2, int slot0, unsigned char* mask) {
    const int spr = 1 << slots_log2;                 // 16-byte slots per row
    const int slot = tid & (spr - 1);
    for (int po = tid >> slots_log2; po < NPX; po += (256 >> slots_log2)) {
      const int sm = po >> lrw, rem = po & ((1 << lrw) - 1), rr = rem >> lw, xx = rem & (W - 1);
      const int b = b0 + sm;
      if (b >= s.B) continue;
      const int ph = sm * g.RH * W + (rr + halo) * W + xx;
      const long m = (long)b * HW + (row0 + rr) * W + xx;
      const u32x4f v = *reinterpret_cast<const u32x4f*>(img + f2_h1_off(ph, slot0 + slot));
      *reinterpret_cast<u32x4f*>(dstg + m * HID + col0 + slot * 8) = v;
      if (mask) {   // post-ReLU bf16 values are >= +0: positive <=> non-zero bits
        unsigned bits = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          bits |= ((v[e] & 0xFFFFu) ? 1u : 0u) << (2 * e) | ((v[e] >> 16) ? 1u : 0u) << (2 * e + 1);
        mask[m * (HID / 8) + (col0 >> 3) + slot] = (unsigned char)bits;
      }
    }
  };

  // ------------------------------------------------------------------ 3. conv1 -> H1
  constexpr int Kp2 = HID, Kp3 = 9 * HID;    // both multiples of 64: no row padding
  constexpr int NCH = 4 / NS;                // channel tiles (32 channels) of conv2 per wave: tiles (nc0 + i) * 4 + wid
  constexpr int RD = 32 / NCH;               // W2 register ring: RD k-steps x NCH tiles = 32 fragments (128 VGPRs)
  const int nc0 = qs * NCH;
  bf16x8f w2r[RD][NCH];
  const __amdgpu_buffer_rsrc_t r2 = f2_rsrc(F2S(sp, wf2), (unsigned)HID * Kp2 * 2);
  const __amdgpu_buffer_rsrc_t r3 = f2_rsrc(F2S(sp, wf3), (unsigned)NT3 * 16 * Kp3 * 2);
  int w2s[NCH];   // scalar byte offset of k-step 0 of each of this wave's channel tiles
#pragma unroll
  for (int i = 0; i < NCH; ++i) w2s[i] = ((nc0 + i) * 4 + swid) * (Kp2 / 16) * 1024;
  // the ring's first fill travels while conv1 computes: eight 1 KiB loads per channel tile of conv1, between its MFMAs
  // and its epilogue (all 32 at once hold every wave at the vector-memory issue for ~2.5K cycles, the last wave longest)
  auto load_w2_part = [&](int part) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int f = part * 8 + e;   // fragment f of the fill: k-step f / NCH of tile f % NCH
      w2r[f / NCH][f % NCH] = f2_ldw(r2, voffL, w2s[f % NCH] + (f / NCH) * 1024);
    }
  };
  {
    const char* x1_lane = X1 + l32 * g.X1S + hi * 16;
    char* h1w[MT32];
#pragma unroll
    for (int mt = 0; mt < MT32; ++mt) h1w[mt] = H1 + (mt * 32 + l32) * 1024 + 8 * hi;
    // the im2col fragments are the same for all four channel tiles: where they fit they are read from LDS once
    constexpr bool XB_REG = (K1M * MT32 <= 16);
    constexpr int XBN = XB_REG ? K1M : 2;
    bf16x8f xb[XBN][MT32];
    auto rd1 = [&](int ks, bf16x8f (&f)[MT32]) {
#pragma unroll
      for (int mt = 0; mt < MT32; ++mt)
        f[mt] = *reinterpret_cast<const bf16x8f*>(x1_lane + mt * 32 * g.X1S + ks * 32);
    };
    if (XB_REG) {
#pragma unroll
      for (int ks = 0; ks < K1M; ++ks) rd1(ks, xb[ks]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f32x16 acc[MT32];
#pragma unroll
      for (int mt = 0; mt < MT32; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;
      if (!XB_REG) rd1(0, xb[0]);
#pragma unroll
      for (int ks = 0; ks < K1M; ++ks) {
        if (!XB_REG) {
          if (ks + 1 < K1M) rd1(ks + 1, xb[(ks + 1) & 1]);
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int mt = 0; mt < MT32; ++mt)
          acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1[j % W1R][ks], xb[XB_REG ? ks : (ks & 1)][mt], acc[mt],
                                                            0, 0, 0);
        if (!XB_REG) __builtin_amdgcn_sched_barrier(0);
      }
      if (j + W1R < 4) load_w1(r1, j + W1R, w1[j % W1R]);
      load_w2_part(j);
      const int n0 = (wid * 4 + j) * 32 + 4 * hi;
      // h1 write address: px*1024 + (((slot ^ px) & 15) | (slot & 48)) * 16 + 8*hi with slot = wid*16 + j*4 + gi:
      // the low 2 slot bits are gi, so (slot ^ px) & 15 = ((j*4 + gi) ^ (px & 15)); px & 15 = l32 & 15 in every tile
      int h1ws[4];
#pragma unroll
      for (int gi = 0; gi < 4; ++gi) h1ws[gi] = ((((j * 4 + gi) ^ l32) & 15) << 4) + wid * 256;
#pragma unroll
      for (int gi = 0; gi < 4; ++gi) {
        const int n = n0 + 8 * gi;
        const f32x4 bb = *reinterpret_cast<const f32x4*>(Pb1 + n), es = *reinterpret_cast<const f32x4*>(Pe1 + n);
#pragma unroll
        for (int mt = 0; mt < MT32; ++mt) {
          const f32x2 v0 = f32x2{acc[mt][4 * gi], acc[mt][4 * gi + 1]} * f32x2{es[0], es[1]} + f32x2{bb[0], bb[1]};
          const f32x2 v1 = f32x2{acc[mt][4 * gi + 2], acc[mt][4 * gi + 3]} * f32x2{es[2], es[3]} + f32x2{bb[2], bb[3]};
          const u32x2f pk = {f2_relu_pack(v0), f2_relu_pack(v1)};
          *reinterpret_cast<u32x2f*>(h1w[mt] + h1ws[gi]) = pk;
        }
      }
    }
  }
  __syncthreads();
  if (c_h1save && qs == 0) save_rows(c_h1save, 0, H1, 6, 0, c_masksave);
  stamp(3);

  // ------------------------------------------------------------------ 4. conv2: ONE pass over k for all of the wave's channel tiles
  // Every B fragment (32 pixels x 16 k of h1) read from LDS feeds NCH MFMAs per pixel tile instead of one: the LDS
  // array, which a chunk-by-chunk order keeps saturated (each wave re-reads all of h1 once per chunk), runs at 1/NCH of
  // that, and the phase is bounded by the MFMAs.  W2 fragments stream from L2 through the register ring, each slot
  // reloaded IN PLACE right after its last use with the fragment RD k-steps ahead (thousands of cycles: an L2 miss to
  // the Infinity Cache / HBM is covered); the loads sit between the MFMAs, so the vector-memory issue (16 cycles per
  // KiB and CU) stays under them instead of stalling all four waves in a burst.
  // conv3's weight fragments: a register ring D3 chunks deep (all four chunks when conv3 has one row tile).  The first
  // fill is requested as soon as conv2's k loop ends, so that it travels under conv2's epilogue (a stretch of pure VALU
  // work with nothing else on the vector-memory path): one chunk ahead the loop waited on every tap for a load issued
  // only nine taps earlier.
  constexpr int D3 = (NT3 == 1 ? 4 : (NT3 == 2 ? 2 : 1)) < NCH ? (NT3 == 1 ? 4 : (NT3 == 2 ? 2 : 1)) : NCH;
  bf16x8f w3[D3][9][NT3];
  auto load_w3 = [&](int slot, int nc, int tap) {   // k-step (of 32) tap*16 + nc*4 + swid of every row tile; rows behind Cout are zero
#pragma unroll
    for (int nt = 0; nt < NT3; ++nt)
      w3[slot][tap][nt] = f2_ldw(r3, voffL, (nt * (Kp3 / 32) + tap * 16 + nc * 4 + swid) * 1024);
  };
  {
    // h1 B-fragment address: px = mt*32 + l32 (px & 15 == l32 & 15), slot = 2*ks + hi:
    //   px*1024 + ((((2ks) & 15) ^ pc) | ((2ks) & 48)) * 16   with the lane constant pc = (l32 & 15) ^ hi
    // the XOR term takes 8 values per lane (k-step mod 8): 8 lane-constant addresses; m-tile and k-step/8 are immediates
    const int pc = (l32 & 15) ^ hi;
    const char* h1r[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) h1r[e] = H1 + l32 * 1024 + (((2 * e) ^ pc) << 4);
    f32x16 acc2[NCH][MT32];
#pragma unroll
    for (int i = 0; i < NCH; ++i)
#pragma unroll
      for (int mt = 0; mt < MT32; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[i][mt][r] = 0.f;
    bf16x8f fb[2][MT32];
    auto rd2 = [&](int ks, bf16x8f (&f)[MT32]) {
#pragma unroll
      for (int mt = 0; mt < MT32; ++mt)
        f[mt] = *reinterpret_cast<const bf16x8f*>(h1r[ks & 7] + mt * 32768 + (ks >> 3) * 256);
    };
    rd2(0, fb[0]);
#pragma unroll
    for (int ks = 0; ks < 32; ++ks) {
      if (ks + 1 < 32) rd2(ks + 1, fb[(ks + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
#pragma unroll
        for (int mt = 0; mt < MT32; ++mt)
          acc2[i][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2r[ks % RD][i], fb[ks & 1][mt], acc2[i][mt], 0, 0, 0);
        if (ks + RD < 32) w2r[ks % RD][i] = f2_ldw(r2, voffL, w2s[i] + (ks + RD) * 1024);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    stamp(6);
#pragma unroll
    for (int c = 0; c < D3; ++c)
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) load_w3(c, nc0 + c, tap);
    __syncthreads();   // every wave is done reading h1: h2 takes its place in the same image (same layout)
    // h2 write address: px*1024 + ((nc0 + i)*16 + ((wid*4 + gi) ^ px) & 15) * 16 + 8*hi
    char* h2w[MT32];
    int h2ws[4];
#pragma unroll
    for (int mt = 0; mt < MT32; ++mt) h2w[mt] = H1 + (mt * 32 + l32) * 1024 + nc0 * 256 + 8 * hi;
#pragma unroll
    for (int gi = 0; gi < 4; ++gi) h2ws[gi] = (((wid * 4 + gi) ^ l32) & 15) << 4;
    const int nl = nc0 * 128 + wid * 32 + 4 * hi;
#pragma unroll
    for (int i = 0; i < NCH; ++i)
#pragma unroll
      for (int gi = 0; gi < 4; ++gi) {
        const int n = nl + i * 128 + 8 * gi;
        const f32x4 bb = *reinterpret_cast<const f32x4*>(Pb2 + n), es = *reinterpret_cast<const f32x4*>(Pe2 + n);
#pragma unroll
        for (int mt = 0; mt < MT32; ++mt) {
          const f32x2 v0 = f32x2{acc2[i][mt][4 * gi], acc2[i][mt][4 * gi + 1]} * f32x2{es[0], es[1]} + f32x2{bb[0], bb[1]};
          const f32x2 v1 =
              f32x2{acc2[i][mt][4 * gi + 2], acc2[i][mt][4 * gi + 3]} * f32x2{es[2], es[3]} + f32x2{bb[2], bb[3]};
          const u32x2f pk = {f2_relu_pack(v0) & hmask[mt], f2_relu_pack(v1) & hmask[mt]};   // zero padding of conv3
          *reinterpret_cast<u32x2f*>(h2w[mt] + i * 256 + h2ws[gi]) = pk;
        }
      }
  }
  __syncthreads();
  stamp(7);
  if (c_h2save) {
    unsigned char* m2 = c_masksave ? c_masksave + (long)s.B * HW * (HID / 8) : nullptr;
    if (NCH == 4) save_rows(c_h2save, 0, H1, 6, 0, m2);
    else
      for (int i = 0; i < NCH; ++i) save_rows(c_h2save, (nc0 + i) * 128, H1, 4, (nc0 + i) * 16, m2);
  }

  // ------------------------------------------------------------------ 5. conv3 over the wave's K-slices of h2
  // 3x3 512->Cout: 9 taps = shifted rows of the h2 image; for every 128-channel chunk wave w takes the chunk's w-th 32
  // channels (one 16x16x32 k-step) for all 9 taps; the four waves' sums are reduced through LDS in 6.
  f32x4 acc3[NT3][MT_O];
#pragma unroll
  for (int i = 0; i < NT3; ++i)
#pragma unroll
    for (int j = 0; j < MT_O; ++j) acc3[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  {
    // B operand: byte offset into the image per (tap, output m-tile) for chunk nc0; lane constants.  Chunk nc0 + i is
    // 256*i bytes further (an immediate); taps that fall outside the image read the zero slot, which therefore exists
    // once per chunk, at the same 256-byte pitch.
    // (unsigned arithmetic: a tap outside the image gives a meaningless pixel index, which the select discards)
    int c3_off[9][MT_O];
    const unsigned k4 = (unsigned)(wid * 4 + q4), zoff = (unsigned)g.oZero + (unsigned)nc0 * 256u;
    const unsigned ioff = (unsigned)g.oH1 + (unsigned)nc0 * 256u;
#pragma unroll
    for (int mo = 0; mo < MT_O; ++mo) {
      const int po = mo * 16 + r16;
      const int sm = po >> lrw, rem = po & ((1 << lrw) - 1), rr = (rem >> lw) + halo, xx = rem & (W - 1);
      const unsigned px0 = (unsigned)(sm * g.RH * W + rr * W + xx);
      unsigned rok[3], xok[3];   // all-ones / zero masks (selects by mask: conditions here otherwise become 36 branches)
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        rok[d] = 0u - (unsigned)((unsigned)(rr + d - 1) < (unsigned)g.RH);
        xok[d] = 0u - (unsigned)((unsigned)(xx + d - 1) < (unsigned)W);
      }
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int dy = tap / 3 - 1, dx = tap % 3 - 1;
        const unsigned px = px0 + (unsigned)(dy * W + dx);
        const unsigned addr = ioff + (px << 10) + (((k4 ^ px) & 15u) << 4);
        c3_off[tap][mo] = (int)(zoff ^ ((addr ^ zoff) & rok[dy + 1] & xok[dx + 1]));
      }
    }
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      // a tap is only 4*NT3 short MFMAs: the LDS reads run three taps ahead
      bf16x8f fb[4][MT_O];
      auto rd3 = [&](int tap, bf16x8f (&f)[MT_O]) {
#pragma unroll
        for (int mo = 0; mo < MT_O; ++mo)
          f[mo] = *reinterpret_cast<const bf16x8f*>(smem_f2 + c3_off[tap][mo] + i * 256);
      };
      rd3(0, fb[0]);
      rd3(1, fb[1]);
      rd3(2, fb[2]);
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        if (tap + 3 < 9) rd3(tap + 3, fb[(tap + 3) & 3]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mo = 0; mo < MT_O; ++mo)
#pragma unroll
          for (int nt = 0; nt < NT3; ++nt)
            acc3[nt][mo] =
                __builtin_amdgcn_mfma_f32_16x16x32_bf16(w3[i % D3][tap][nt], fb[tap & 3][mo], acc3[nt][mo], 0, 0, 0);
        if (i + D3 < NCH) load_w3(i % D3, nc0 + i + D3, tap);   // in place, D3 chunks ahead
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  __syncthreads();
  stamp(4);

  // ------------------------------------------------------------------ 6. cross-wave reduction of conv3, coupling tail
  float* R3 = reinterpret_cast<float*>(H1);     // [4 waves][64 px][NT3*16]   (h1 is dead now)
#pragma unroll
  for (int nt = 0; nt < NT3; ++nt)
#pragma unroll
    for (int mo = 0; mo < MT_O; ++mo)
      *reinterpret_cast<f32x4*>(R3 + ((wid * NPX + mo * 16 + r16) * NP + nt * 16 + q4 * 4)) = acc3[nt][mo];
  const bool more = MULTI && st + 1 < nst;
  if (more) {
    // a run of steps: the next step's parameters are requested here and travel under the tail
    load_params((CStep*)(a.run.steps + st + 1));
  }
  __syncthreads();
  const int per_px = R * W;
  if (NS > 1) {
    // partial conv3 sums of this workgroup's chunks -> slab qs; bias, scale and the coupling tail follow in
    // coupling_fwd_kernel once all NS slabs are there
    float* slab = a.c.ws + (long)qs * s.B * HW * Cout;
    for (int idx = tid; idx < NPX * Cout; idx += 256) {
      const int po = (int)__umulhi((unsigned)idx, 0xFFFFFFFFu / (unsigned)Cout + 1u), n = idx - po * Cout;
      const int sm = po >> lrw, rem = po & ((1 << lrw) - 1), rr = rem >> lw, xx = rem & (W - 1);
      const int b = b0 + sm;
      if (b >= s.B) continue;
      const float* p = R3 + po * NP + n;
      slab[((long)b * HW + (row0 + rr) * W + xx) * Cout + n] = (p[0] + p[NPX * NP]) + (p[2 * NPX * NP] + p[3 * NPX * NP]);
    }
    stamp(5);
    return;
  }
  // Coupling tail on PAIRS of z2 channels (C/2 is even): the four waves' conv3 partial sums of a pair's (shift, scale)
  // values are four 16-byte LDS reads, the outputs 8- / 16-byte stores.  Per-sample sums of log(scale): one workgroup
  // reduction when the tile is one sample; with several samples per tile every pair leaves its sum in LDS (the im2col
  // tile is dead) and wave w then adds up samples w, w + 4, ... -- one barrier instead of two per sample.
  __shared__ float sh2[32];
  typedef float f32x2t __attribute__((ext_vector_type(2)));
  const int C2h = C2 >> 1, Chh = Ch >> 1;
  const unsigned mC2h = 0xFFFFFFFFu / (unsigned)C2h + 1u, mChh = 0xFFFFFFFFu / (unsigned)Chh + 1u;
  const float dld0 = F2S(sp, dld)[0];
  float* LS = reinterpret_cast<float*>(X1);
  float ls = 0.f;
  for (int idx = tid; idx < NPX * C2h; idx += 256) {
    const int po = (int)__umulhi((unsigned)idx, mC2h), j = (idx - po * C2h) * 2;
    const int sm = po >> lrw, pl = po & ((1 << lrw) - 1), rr = pl >> lw, xx = pl & (W - 1);
    const int b = b0 + sm;
    float lsp = 0.f;
    if (b < s.B) {
      const long m = (long)b * HW + (row0 + rr) * W + xx;
      const int zi = ((sm * g.RZ + rr + zh) * W + xx) * C + Ch + j;
      const f32x2t z2 = *reinterpret_cast<const f32x2t*>(Zm + zi);
      f32x2t out;
      if (s.additive) {
        const float* p = R3 + po * NP + j;
        auto ld2 = [](const float* q) { return *reinterpret_cast<const f32x2t*>(q); };
        const f32x2t h = ((ld2(p) + ld2(p + NPX * NP)) + (ld2(p + 2 * NPX * NP) + ld2(p + 3 * NPX * NP)) + ld2(Pb3 + j)) * ld2(Pe3 + j);
        if (c_hsave) *reinterpret_cast<f32x2t*>(c_hsave + m * Cout + j) = h;
        out = z2 + h;
      } else {
        const float* p = R3 + po * NP + 2 * j;   // (shift, scale) of channel j, then of channel j + 1
        auto ld4 = [](const float* q) { return *reinterpret_cast<const f32x4*>(q); };
        const f32x4 h = ((ld4(p) + ld4(p + NPX * NP)) + (ld4(p + 2 * NPX * NP) + ld4(p + 3 * NPX * NP)) + ld4(Pb3 + 2 * j)) * ld4(Pe3 + 2 * j);
        if (c_hsave) *reinterpret_cast<f32x4*>(c_hsave + m * Cout + 2 * j) = h;
        float s0, s1, l0, l1;
        f2_sigmoid_logsigmoid(h[1] + 2.f, s0, l0);
        f2_sigmoid_logsigmoid(h[3] + 2.f, s1, l1);
        out = f32x2t{(z2[0] + h[0]) * s0, (z2[1] + h[2]) * s1};
        lsp = l0 + l1;
      }
      if (!MULTI || c_zout) *reinterpret_cast<f32x2t*>(c_zout + m * C + Ch + j) = out;
      if (more) *reinterpret_cast<f32x2t*>(Xraw + zi) = out;   // a run's tiles are whole images (zh == 0): Zm's pixel order is the x tile's
    }
    if (TS == 1) ls += lsp;
    else LS[idx] = lsp;
  }
  for (int idx = tid; idx < NPX * Chh; idx += 256) {   // z1 passes through
    const int po = (int)__umulhi((unsigned)idx, mChh), j = (idx - po * Chh) * 2;
    const int sm = po >> lrw, pl = po & ((1 << lrw) - 1), rr = pl >> lw, xx = pl & (W - 1);
    const int b = b0 + sm;
    if (b >= s.B) continue;
    const int zi = ((sm * g.RZ + rr + zh) * W + xx) * C + j;
    const f32x2t z1 = *reinterpret_cast<const f32x2t*>(Zm + zi);
    if (!MULTI || c_zout) *reinterpret_cast<f32x2t*>(c_zout + ((long)b * HW + (row0 + rr) * W + xx) * C + j) = z1;
    if (more) *reinterpret_cast<f32x2t*>(Xraw + zi) = z1;
  }
  stamp(11);
  if (TS == 1) {
    ls = block_sum(ls, sh2);
    if (tid == 0 && b0 < s.B) c_ldpart[(long)b0 * tps + tile] = ls + (tile == 0 ? dld0 : 0.f);
  } else {
    __syncthreads();
    const int pps = per_px * C2h;   // pairs per sample, contiguous in LS
    for (int sm = wid; sm < TS; sm += 4) {
      float v = 0.f;
      for (int e = lane; e < pps; e += 64) v += LS[sm * pps + e];
      v = wave_sum(v);
      if (lane == 0 && b0 + sm < s.B) c_ldpart[(long)(b0 + sm) * tps + tile] = v + dld0;   // several samples per tile: tile == 0
    }
  }
  stamp(5);
  if (more) __syncthreads();   // the tail is done with Zm, R3 and the parameters: the next step may overwrite them
  }   // steps of the run
#undef F2S
}

static int ilog2_exact(int v) {
  if (v <= 0 || (v & (v - 1))) return -1;
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}

static bool f2_pick(const nfk_step2d& s, F2Args* a) {
  if (!s.bf16 || s.hidden != 512 || s.Cout > 48 || s.Cout < 1 || (s.C & 3) != 0 || s.C < 4) return false;
  if (!s.wf1 || !s.wf2 || !s.wf3) return false;   // needs the fragment-ordered weight copies
  const int lw = ilog2_exact(s.W), lh = ilog2_exact(s.H);
  if (lw < 1 || lh < 0) return false;   // W even: the channel mix takes pixels in pairs of one row
  const int HW = s.H * s.W;
  if (HW <= 64) { a->R = s.H; a->TS = 64 / HW; a->halo = 0; }
  else if (s.W <= 64 && s.H % (64 / s.W) == 0 && 64 / s.W < s.H) {
    a->R = 64 / s.W; a->TS = 1; a->halo = 1;
    // rows of 32 pixels (CelebA-64 level 1): two rows + halo would be 128 pixels of h-region (16 accumulator tiles per
    // wave in conv2: no room); ONE row per tile instead -- 32 output pixels, 96 computed (three times the conv1 / conv2
    // work per output pixel instead of 1.5 times: still several times the per-conv launches' speed)
    if (a->R == 2 && s.W == 32 && f2_geom(s, 2, 1, 1).MT32 > 3) a->R = 1;
  }
  else return false;
  const F2Geom g = f2_geom(s, a->R, a->TS, a->halo);
  if (g.MT32 != 2 && g.MT32 != 3) return false;
  if (a->TS == 1 && a->R * s.W != 64 && !(a->R * s.W == 32 && g.MT32 == 3)) return false;   // output pixels per tile: 64, or 32 (MO = 2)
  if (g.K1 > 256 || g.K1M * 16 != (g.K1 + 63) / 64 * 64) return false;   // wf1's k extent must be K1M k-steps
  if (g.total > 160 * 1024 - 512) return false;
  if ((size_t)4 * 64 * g.NT3 * 16 * 4 > (size_t)g.MT32 * 32 * 1024) return false;   // R3 lives in the h1 region
  if ((size_t)a->TS * g.RZ * s.W * s.C * 4 > (size_t)g.MT32 * 32 * 1024) return false;  // so does the raw x staging
  if (a->TS * g.RZ * s.W * s.C > 12 * 256 || s.C * s.C > 9 * 256 || s.Cout > 256) return false;  // prologue register slots
  a->lw = lw;
  a->lrw = ilog2_exact(a->R * s.W);
  a->lhw = ilog2_exact(g.RH * s.W);   // -1 when TS == 1 with halo rows: unused there
  if (a->lrw < 0 || (a->TS > 1 && a->lhw < 0)) return false;
  if (a->lhw < 0) a->lhw = 0;
  return true;
}

template <int MT32, int NT3, int K1M, int NS, bool MULTI = false, int MO = 4>
static int f2_launch_t(const F2Args& a, const F2Geom& g, int grid, hipStream_t st) {
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(fused2_step_fwd_kernel<MT32, NT3, K1M, NS, MULTI, MO>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512) != hipSuccess)
      return NFK_ERR_LAUNCH;
    attr = true;
  }
  hipLaunchKernelGGL((fused2_step_fwd_kernel<MT32, NT3, K1M, NS, MULTI, MO>), dim3(grid * NS), dim3(256), g.total, st, a);
  NFK_LAUNCH_CHECK();
  return NFK_OK;
}

template <int MT32, int NS, bool MULTI = false, int MO = 4>
static int f2_dispatch(const F2Args& a, const F2Geom& g, int grid, hipStream_t st) {
  // instantiated for the shapes Glow produces: conv3 width grows with conv1's K (both follow C)
  if (g.NT3 == 1 && g.K1M == 4) return f2_launch_t<MT32, 1, 4, NS, MULTI, MO>(a, g, grid, st);
  if (MO != 4) return NFK_ERR_ARG;   // (32-pixel rows: level 1 only, C = 12)
  if (g.NT3 == 2 && g.K1M == 8) return f2_launch_t<MT32, 2, 8, NS, MULTI>(a, g, grid, st);
  if (g.NT3 == 3 && g.K1M == 16) return f2_launch_t<MT32, 3, 16, NS, MULTI>(a, g, grid, st);
  return NFK_ERR_ARG;
}

static int f2_grid(const nfk_step2d& s, const F2Args& a) {
  return a.TS == 1 ? s.B * (s.H / a.R) : (s.B + a.TS - 1) / a.TS;
}
// hidden split (the caller asks for `ways` = 4 or 2): only for tiles that need no halo (whole images per workgroup)
// when the grid would leave most CUs idle
static int f2_splits(const nfk_step2d& s, const F2Args& a, const F2Geom& g, int ways) {
  if (ways < 2 || g.MT32 != 2 || f2_grid(s, a) >= 128) return 1;
  return ways >= 4 ? 4 : 2;
}

int fused2_tiles(const nfk_step2d& s) {
  F2Args a{};
  if (!f2_pick(s, &a)) return 0;
  const F2Geom g = f2_geom(s, a.R, a.TS, a.halo);
  const bool inst = (g.NT3 == 1 && g.K1M == 4) || (g.NT3 == 2 && g.K1M == 8) || (g.NT3 == 3 && g.K1M == 16);
  if (!inst) return 0;
  if (a.TS == 1 && a.R * s.W == 32 && !(g.NT3 == 1 && g.K1M == 4)) return 0;   // 32-pixel tiles: the C = 12 shape only
  return a.TS == 1 ? s.H / a.R : 1;
}

int fused2_k1p(const nfk_step2d& s) { return (9 * (s.C / 2) + 63) / 64 * 64; }

size_t fused2_ws_floats(const nfk_step2d& s) {
  F2Args a{};
  if (!f2_pick(s, &a)) return 0;
  const F2Geom g = f2_geom(s, a.R, a.TS, a.halo);
  return f2_splits(s, a, g, 4) > 1 ? (size_t)4 * s.B * s.H * s.W * s.Cout : 0;   // sized for the widest split
}

int fused2_launch(const FusedCall& c, hipStream_t st) {
  F2Args a{};
  if (!f2_pick(c.s, &a)) return NFK_ERR_ARG;
  a.c = c;
  const F2Geom g = f2_geom(c.s, a.R, a.TS, a.halo);
  const int grid = f2_grid(c.s, a);
  const int ns = c.ws ? f2_splits(c.s, a, g, c.split_ways) : 1;   // ws == NULL: the caller does not want the hidden split
  if (ns == 1) {
    if (a.TS == 1 && a.R * c.s.W == 32) return f2_dispatch<3, 1, false, 2>(a, g, grid, st);   // one 32-pixel row per tile
    return g.MT32 == 2 ? f2_dispatch<2, 1>(a, g, grid, st) : f2_dispatch<3, 1>(a, g, grid, st);
  }
  // hidden split: the step's mixed z must be kept (the coupling tail reads it), and the tail is a second launch
  if (!c.zsave) return NFK_ERR_ARG;
  { const int rc = ns == 2 ? f2_dispatch<2, 2>(a, g, grid, st) : f2_dispatch<2, 4>(a, g, grid, st); if (rc != NFK_OK) return rc; }
  const nfk_step2d& s = c.s;
  CouplingArgs cp{};
  cp.a3 = c.ws; cp.S = ns; cp.slab_stride = (long)s.B * s.H * s.W * s.Cout;
  cp.bias3 = s.bias3; cp.es3 = s.es3; cp.z = c.zsave; cp.ldz = s.C; cp.zout = c.zout; cp.ldo = s.C; cp.hsave = c.hsave;
  cp.ld_in = nullptr; cp.dld = s.dld; cp.ld_out = c.ldpart;   // one log-det slot per sample (tiles == 1 here)
  cp.HW = s.H * s.W; cp.C = s.C; cp.Cout = s.Cout; cp.additive = s.additive; cp.has_ld = 1;
  return launch_coupling_fwd(cp, s.B, st);
}

static int g_run_half_tiles = 1;   // experiment knob (nfk_tune NFK_TUNE_RUN_HALF_TILES)
void fused2_set_half_tiles(int on) { g_run_half_tiles = on; }

// a run of steps in one launch: tiles must be whole images (no halo rows) and the next-x tile must fit behind R3
static bool f2_run_ok(const nfk_step2d& s, F2Args* a, F2Geom* gp) {
  if (!f2_pick(s, a) || a->halo != 0) return false;
  const F2Geom g = f2_geom(s, a->R, a->TS, a->halo);
  if (g.MT32 != 2) return false;
  if (!((g.NT3 == 1 && g.K1M == 4) || (g.NT3 == 2 && g.K1M == 8) || (g.NT3 == 3 && g.K1M == 16))) return false;
  if ((size_t)4 * 64 * g.NT3 * 16 * 4 + (size_t)a->TS * g.RZ * s.W * s.C * 4 > (size_t)g.MT32 * 32 * 1024) return false;
  *gp = g;
  return true;
}
int fused2_run_ok(const nfk_step2d& s) {
  F2Args a{};
  F2Geom g;
  return f2_run_ok(s, &a, &g) ? 1 : 0;
}
int fused2_run_launch(const nfk_fused_run& r, const nfk_step2d& first, unsigned long long* stamps, hipStream_t st) {
  F2Args a{};
  F2Geom g;
  if (r.nsteps < 1 || r.nsteps > 64 || !r.steps || !r.x || !r.ldpart || !r.zout) return NFK_ERR_ARG;
  if (!((r.keep >> (r.nsteps - 1)) & 1ull)) return NFK_ERR_ARG;   // the run's last output is always written
  if (r.masksave && !(r.h1save && r.h2save)) return NFK_ERR_ARG;
  if (!f2_run_ok(first, &a, &g)) return NFK_ERR_ARG;
  a.c.s = first; a.c.x = r.x; a.c.ldx = r.ldx; a.c.stamps = stamps;
  a.run = r;
  // Tiny images on a small grid (CIFAR level 3 at B = 64: 16 tiles of four 4x4 images): 32-pixel tiles instead -- twice the
  // workgroups, each with half of everything but the weight stream.  Same arithmetic per pixel and per sample: bit-identical.
  const int HW = first.H * first.W;
  if (g_run_half_tiles && HW <= 16 && f2_grid(first, a) < 64 && g.NT3 == 3 && g.K1M == 16) {
    F2Args h = a;
    // 1 (default): 16-pixel tiles (one 4x4 image per workgroup) when even 32-pixel tiles would leave fewer than 64
    // workgroups, else 32-pixel tiles; 2 / 3: always 16 / 32 (experiments)
    const int npx = (HW == 16 && g_run_half_tiles != 3 && (g_run_half_tiles == 2 || f2_grid(first, a) < 32)) ? 16 : 32;
    h.TS = npx / HW;
    h.lrw = ilog2_exact(h.R * first.W);
    h.lhw = ilog2_exact((h.R + 2 * h.halo) * first.W);
    const F2Geom gh = f2_geom(first, h.R, h.TS, h.halo);
    const bool fits = gh.MT32 == 1 && gh.total <= 160 * 1024 - 512 &&
                      (size_t)4 * npx * gh.NT3 * 16 * 4 + (size_t)h.TS * gh.RZ * first.W * first.C * 4 <= (size_t)gh.MT32 * 32 * 1024;
    if (fits && npx == 32) return f2_launch_t<1, 3, 16, 1, true, 2>(h, gh, (first.B + h.TS - 1) / h.TS, st);
    if (fits && npx == 16) return f2_launch_t<1, 3, 16, 1, true, 1>(h, gh, first.B, st);
  }
  return f2_dispatch<2, 1, true>(a, g, f2_grid(first, a), st);
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

// Several partial buffers in sequence (the FlowSteps of a level, a Split2d, the next level, ...): exactly the arithmetic of
// one ld_finish_kernel launch per buffer, each starting from the previous one's result -- bit-identical, one launch.
struct LdMulti {
  const float* part[NFK_LD_MAX_PARTS];
  int nsteps[NFK_LD_MAX_PARTS], tiles[NFK_LD_MAX_PARTS];
  int n;
};
// One wave per sample: the lanes fetch 64 partials at once, then they are added in the SAME order as the one-thread loop did
// (v += partial 0, 1, 2, ...: bit-identical) by reading the lanes in turn -- the loop was a chain of ~50 dependent loads per
// sample (13 us on the training step's critical chain, between the forward and the loss).
__global__ void __launch_bounds__(64) ld_finish_multi_kernel(const float* __restrict__ ld_in, LdMulti m, int B,
                                                              float* __restrict__ ld_out) {
  const int b = blockIdx.x, lane = threadIdx.x;
  if (b >= B) return;
  float v = ld_in[b];
  for (int k = 0; k < m.n; ++k) {
    const float* __restrict__ p = m.part[k];
    const int tps = m.tiles[k], total = m.nsteps[k] * tps;
    for (int base = 0; base < total; base += 64) {
      const int j = base + lane, st = j / tps, t = j - st * tps;
      const float x = j < total ? p[((long)st * B + b) * tps + t] : 0.f;
      const int cnt = total - base < 64 ? total - base : 64;
      for (int i = 0; i < cnt; ++i) v += __shfl(x, i, 64);
    }
  }
  if (lane == 0) ld_out[b] = v;
}

static unsigned long long* g_stamps = nullptr;   // diagnostic only (nfk_fused_set_stamps)

}  // namespace nfk

using namespace nfk;
extern "C" {

void nfk_fused_set_stamps(void* buf) { g_stamps = reinterpret_cast<unsigned long long*>(buf); }

int nfk_step2d_fused_tiles(const nfk_step2d* s) { return s->bf16 ? fused2_tiles(*s) : 0; }

size_t nfk_step2d_fused_ws_floats(const nfk_step2d* s) { return nfk_step2d_fused_tiles(s) ? fused2_ws_floats(*s) : 0; }

int nfk_step2d_fwd_fused(const nfk_step2d* s, const float* x, int ldx, float* zsave, void* h1save, void* h2save,
                         float* hsave, float* zout, float* ldpart, float* ws, void* masksave, int split_ways,
                         void* stream) {
  if (!nfk_step2d_fused_tiles(s)) return NFK_ERR_ARG;
  if (masksave && !(h1save && h2save)) return NFK_ERR_ARG;
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
  FusedCall c{};
  c.s = *s; c.x = x; c.ldx = ldx; c.zsave = zsave; c.h1save = h1save; c.h2save = h2save; c.hsave = hsave;
  c.masksave = reinterpret_cast<unsigned char*>(masksave);
  c.zout = zout; c.ldpart = ldpart; c.ws = ws; c.stamps = g_stamps; c.split_ways = split_ways;
  prof_scope_begin(cls, flops, bytes, st);
  const int rc = fused2_launch(c, st);
  prof_scope_end(cls, st);
  return rc;
}

int nfk_run2d_fused_ok(const nfk_step2d* s) { return s->bf16 ? fused2_run_ok(*s) : 0; }

int nfk_run2d_fwd_fused(const nfk_fused_run* run, const nfk_step2d* first, void* stream) {
  if (!run || !first || !first->bf16) return NFK_ERR_ARG;
  const nfk_step2d* s = first;
  const double n = run->nsteps, M = (double)s->B * s->H * s->W, hid = s->hidden, K1 = 9.0 * (s->C / 2);
  const double flops = n * 2.0 * M * (K1 * hid + hid * hid + 9.0 * hid * s->Cout);
  double bytes = 4.0 * M * s->C * (1.0 + (double)__builtin_popcountll(run->keep)) + n * 2.0 * (K1 * hid + hid * hid + 9.0 * hid * s->Cout);
  if (run->zsave) bytes += n * 4.0 * M * s->C;
  if (run->h1save) bytes += n * 2.0 * M * hid;
  if (run->h2save) bytes += n * 2.0 * M * hid;
  if (run->hsave) bytes += n * 4.0 * M * s->Cout;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  int cls = NFK_PROF_FUSED_STEP + (s->Cout + 15) / 16 - 1;
  if (cls > NFK_PROF_FUSED_STEP_NT3) cls = NFK_PROF_FUSED_STEP_NT3;
  prof_scope_begin(cls, flops, bytes, st);
  const int rc = fused2_run_launch(*run, *first, g_stamps, st);
  prof_scope_end(cls, st);
  return rc;
}

int nfk_ld_finish_multi(const float* ld_in, const nfk_ld_parts* parts, int n, int B, float* ld_out, void* stream) {
  if (B <= 0) return NFK_OK;
  if (n < 0 || n > NFK_LD_MAX_PARTS || (n > 0 && !parts)) return NFK_ERR_ARG;
  LdMulti m{};
  m.n = n;
  for (int k = 0; k < n; ++k) { m.part[k] = parts[k].part; m.nsteps[k] = parts[k].nsteps; m.tiles[k] = parts[k].tiles; }
  hipLaunchKernelGGL(ld_finish_multi_kernel, dim3(B), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), ld_in, m, B, ld_out);
  NFK_LAUNCH_CHECK();
  return NFK_OK;
}
int nfk_ld_finish(const float* ld_in, const float* ldpart, int nsteps, int B, int tps, float* ld_out, void* stream) {
  if (B <= 0) return NFK_OK;
  hipLaunchKernelGGL(ld_finish_kernel, dim3((B + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), ld_in,
                     ldpart, nsteps, B, tps, ld_out);
  NFK_LAUNCH_CHECK();
  return NFK_OK;
}

}  // extern "C"
