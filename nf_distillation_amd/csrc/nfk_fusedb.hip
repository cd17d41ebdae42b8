// nfk fused FlowStep BACKWARD (bf16 operands / fp32 accumulate): the dgrad chain of a step in ONE launch
// (autograd of models/flows.py:142-171 through the coupling net of models/flows.py:25-34):
//
//     g_a3 [M][Cout] --dgrad3 (3x3, Cout->512)--> g_h2 --ReLU/ActNorm bwd--> g_a2
//          --dgrad2 (1x1, 512->512)--> g_h1 --ReLU/ActNorm bwd--> g_a1 --dgrad1 (3x3, 512->C/2)--> g_z1
//
// It is the mirror image of the fused forward (nfk_fused2.hip), with the same pixel tiles, the same LDS image and the
// same weight streaming; a transposed 3x3 convolution is a 3x3 convolution with flipped taps (the `wd` layouts), so
//   1. stage    the tile's rows (+2 halo rows each side) of g_a3 (fp32), 3x3 im2col -> bf16 tile      (forward: mix + im2col)
//   2. dgrad3   [512 x 9 Cout] x im2col, 32x32x16 MFMA, weights in registers                           (forward: conv1)
//               epilogue: g_a2 = (h2 > 0) ? g * e^logs2 : 0  -> the LDS image (bf16)
//   3. dgrad2   [512 x 512] x image in ONE pass over k, 12 accumulator tiles per wave, W2^T streamed   (forward: conv2)
//               from L2 through a register ring; g_a1 = (h1 > 0) ? g * e^logs1 : 0 then REPLACES g_a2 in the image
//   4. dgrad1   [C/2 x 9*512] x shifted rows of the image, 16x16x32 MFMA, K split over the waves       (forward: conv3)
//   g_a2 / g_a1 leave LDS once, as the rows of the tile's own 64 pixels (bf16): the weight-gradient GEMMs read them.
// Of the saved activations h1 / h2 the chain needs only the ReLU masks: the fused forward leaves them as one bit per
// element (FusedCall.masksave, 1/16 of the bf16 tensors), so this kernel reads g_a3, 2 x 6 KB of mask words and the
// weights -- no 512-wide tensor at all.  The per-channel sums behind the two ActNorm layers' gradients
// (sum g_a h / e^logs, sum g_a) are taken where g_a and h are both streamed anyway: beside the weight-gradient GEMMs
// (actnorm_sum_body in nfk_gemm_bf16.hip).
// What this replaces per step: three GEMM launches whose 512-wide operands (g_a2, g_a1: 16.8 MB each at CIFAR level 1,
// B = 64) made a round trip through HBM each, re-read once per tap by the narrow dgrad, plus the 2 x 16.8 MB of h1 / h2
// their epilogues read.
#include "nfk_internal.h"
#include <type_traits>

namespace nfk {

typedef __bf16 bf16x8f __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4f __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4f __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2f __attribute__((ext_vector_type(2)));

struct FBArgs {
  FusedBwdCall c;        // one step, stand-alone: the call.  Level mode (LV): c.s = the level's first step (dimensions)
  int R, TS, halo;       // rows per tile, samples per tile, 1 if the tile needs halo rows
  int lw, lrw, lhw;      // log2(W), log2(R*W), log2(RH*W) (the last only meaningful when TS > 1)
  // ---- level mode (nfk_level2d_bwd): the launch walks steps k_hi .. k_lo (one step per launch unless MULTI)
  nfk_level_bwd lv;
  int k_hi, k_lo;
  float* gz[2];          // [M][C] rows of dL/dz (mixed tensor), step k in gz[k & 1]: written by the coupling backward
  float* tailpart;       // [nsteps][NFK_LEVEL_ROWS][2][Cout] partial (sum g_a3, 3 sum g_h h) rows of the coupling backward
  float* ga;             // per-step blocks of ga_stride floats: g_a3 fp32 | g_a2 bf16 | g_a1 bf16 | dgrad3's im2col rows bf16
  long ga_stride, o_ga2, o_ga1, o_x3;   // (offsets in floats)
};

struct FBGeom {
  int RH, RZ, P_h, MT32, K3, K3M, K3p, X3S, NTD;
  size_t oG3, oPar, oX3, oImg, oZero, oGZ, total;
};
// level mode: floats of the epilogue's scratch (it lives in the im2col region, dead after dgrad3):
//   GZ' / XIN / GX tiles [npx][C + 1], Wf [C][Cp], the coupling backward's per-group column sums [256 / C2][4][C2]
__host__ __device__ static inline size_t fb_epi_floats(int C, int npx) {
  const int Cp = (C + 3) & ~3, C2 = C - C / 2;
  return (size_t)2 * npx * (C + 1) + (size_t)C * Cp + (size_t)(256 / (C2 > 0 ? C2 : 1)) * 4 * C2;
}

__host__ __device__ static inline FBGeom fb_geom(const nfk_step2d& s, int R, int TS, int halo) {
  FBGeom g;
  const int W = s.W, Ch = s.C / 2;
  g.RH = R + 2 * halo;
  g.RZ = R + 4 * halo;
  g.P_h = TS * g.RH * W;
  g.MT32 = (g.P_h + 31) / 32;
  g.K3 = 9 * s.Cout;
  g.K3p = (g.K3 + 63) / 64 * 64;                     // = the k extent of wd3 / wg3 (nfk_padded_k_bf16)
  g.K3M = g.K3p / 16;                                // dgrad3 k-steps of 16 (zero columns pad the im2col tile)
  g.X3S = g.K3p * 2 + 16;                            // bytes per im2col row: an odd number of 16-byte slots
  g.NTD = (Ch + 15) / 16;
  size_t o = 0;
  g.oG3 = o; o += (size_t)TS * g.RZ * W * s.Cout * 4;
  o = (o + 15) & ~size_t(15);
  g.oPar = o; o += (size_t)2 * s.hidden * 4;
  g.oX3 = o; o += (size_t)g.MT32 * 32 * g.X3S;
  o = (o + 15) & ~size_t(15);
  g.oImg = o; o += (size_t)g.MT32 * 32 * 1024;
  g.oZero = o; o += 3 * 256 + 16;   // sixteen zero bytes per 128-channel chunk, 256 bytes apart (see nfk_fused2.hip)
  g.oGZ = o;                        // level mode only: the tile's rows of dL/dz [TS * R * W][C + 1] fp32 (fb_geom_lv)
  g.total = o;
  return g;
}
__host__ __device__ static inline FBGeom fb_geom_lv(const nfk_step2d& s, int R, int TS, int halo) {
  FBGeom g = fb_geom(s, R, TS, halo);
  g.total = g.oGZ + (size_t)TS * R * s.W * (s.C + 1) * 4;
  return g;
}

__device__ __forceinline__ unsigned fb_pack(float lo, float hi) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  const f32x2 f = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2));
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t fb_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ bf16x8f fb_ldw(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  return __builtin_bit_cast(bf16x8f, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
// image: 1024-byte rows (512 channels); the low 4 bits of the 16-byte slot index are XORed with the pixel index
__device__ __forceinline__ int fb_img_off(int px, int slot) {
  return px * 1024 + ((((slot ^ px) & 15) | (slot & ~15)) << 4);
}

// MO: output pixels per tile in units of 16 -- 4 (64 pixels) or, on tiny images with a small grid (CIFAR level 3), 2 / 1:
// more workgroups, each with a smaller share of everything but the weight stream (as nfk_fused2.hip's runs of steps).
//
// LV ("level mode", nfk_level2d_bwd): the launch belongs to the backward of a whole level.  Per-step pointers come from the
// device arrays lv.steps / lv.grads (scalar loads), per-step buffers are base + k * stride, and the kernel also does what
// the channel-mix launch did for the step: on the tile's own pixels
//     g_z = [dL/dz1 + dgrad1 | dL/dz2]  ->  g_x = Wf^T g_z, partial (g_z x^T, sum g_z) for nfk_prep_bwd
//     -> coupling backward of the step BELOW (k - 1): g_a3, dL/dz of that step, its partial (bias3, logs3) row
// so that a step's backward is ONE launch.  MULTI (tiles that are whole images: no halo rows): the launch walks steps
// k_hi .. k_lo; g_a3 and dL/dz of the next step stay in LDS, only what the weight-gradient GEMMs read goes to HBM.
// The arithmetic per element is that of chanmix_body / coupling_bwd_kernel (nfk_tail.h, nfk_flow.hip).
template <int MT32, int NTD, int K3M, int MO = 4, bool LV = false, bool MULTI = false>
__global__ void __launch_bounds__(256) fusedb_step_bwd_kernel(FBArgs a) {
  static_assert(MO == 4 || ((MO == 2 || MO == 1) && MT32 == 1) || (MO == 2 && (MT32 == 3 || MT32 == 2) && !MULTI),
                "small tiles: whole tiny images, or 32 pixels of image rows (+ halo rows) per tile");
  static_assert(!MULTI || LV, "runs of steps exist in level mode only");
  typedef const __attribute__((address_space(4))) nfk_step2d CStep;        // scalar (constant-memory) loads
  typedef const __attribute__((address_space(4))) nfk_level_grads CGrads;
#define FBS(f) (LV ? sp->f : a.c.s.f)
  constexpr int HID = 512;
  constexpr int NPX = 16 * MO;   // output pixels of a tile
  constexpr int MT_O = MO;       // output m-tiles of 16 (dgrad1 uses 16x16x32)
  extern __shared__ __attribute__((aligned(16))) char smem_fb[];
  const nfk_step2d& s = a.c.s;   // dimensions (the same for every step of a level); per-step POINTERS go through FBS
  const int k_first = LV ? a.k_hi : 0, k_last = LV ? (MULTI ? a.k_lo : a.k_hi) : 0;
  // registers that live from one step of a run to the next: the ReLU mask words, e^logs values and dgrad3's first weight
  // tiles of the NEXT step are requested before the current step's epilogue and travel under it
  constexpr int W3R = (K3M >= 28) ? 1 : ((K3M >= 16) ? 2 : 4);   // register ring: tiles of 32 output channels held ahead
                                                                  // (1: every fragment is reloaded in place right after its use)
  bf16x8f w3[W3R][K3M];
  u32x4f mk2[MT32];
  unsigned mk1[MT32][4];
  float ereg[4];
  for (int k = k_first; k >= k_last; --k) {
  // (a run of steps: thread index and dimensions pass through an empty asm once per step so that what derives from them
  //  is computed where it is used instead of being hoisted out of the step loop and kept live across it: nfk_fused2.hip)
  int tid_ = threadIdx.x, W_ = s.W, H_ = s.H, C_ = s.C, Cout_ = s.Cout, R_ = a.R, TS_ = a.TS;
  if (LV) {
    asm volatile("" : "+v"(tid_));
    asm volatile("" : "+s"(W_), "+s"(H_), "+s"(C_), "+s"(Cout_), "+s"(R_), "+s"(TS_));
  }
  const int W = W_, H = H_, C = C_, Ch = C / 2, Cout = Cout_, HW = H * W;
  const int R = R_, TS = TS_, halo = a.halo, lw = a.lw, lrw = a.lrw, lhw = a.lhw;
  nfk_step2d sd;   // the dimensions fb_geom reads
  sd.W = W; sd.H = H; sd.C = C; sd.Cout = Cout; sd.hidden = HID; sd.B = s.B;
  const FBGeom g = LV ? fb_geom_lv(sd, R, TS, halo) : fb_geom(sd, R, TS, halo);
  CStep* sp = LV ? (CStep*)(a.lv.steps + k) : nullptr;
  // this step's buffers: the call's (stand-alone) or base + k * stride (level mode)
  float* const gablk = LV ? a.ga + (long)k * a.ga_stride : nullptr;
  const float* const c_ga3 = LV ? gablk : a.c.ga3;
  const unsigned char* const c_masks =
      LV ? reinterpret_cast<const unsigned char*>(a.lv.masks) + (long)k * a.lv.mask_stride : a.c.masks;
  void* const c_ga2 = LV ? (void*)(gablk + a.o_ga2) : a.c.ga2;
  void* const c_ga1 = LV ? (void*)(gablk + a.o_ga1) : a.c.ga1;
  void* const c_x3save = LV ? (void*)(gablk + a.o_x3) : a.c.x3save;
  float* G3 = reinterpret_cast<float*>(smem_fb + g.oG3);
  float* Pe2 = reinterpret_cast<float*>(smem_fb + g.oPar);
  float* Pe1 = Pe2 + HID;
  char* X3 = smem_fb + g.oX3;
  char* IMG = smem_fb + g.oImg;
  const int tid = tid_, lane = tid & 63, wid = tid >> 6;
  const int l32 = lane & 31, hi = lane >> 5;      // 32x32x16 operand roles: row / k-half
  const int r16 = lane & 15, q4 = lane >> 4;      // 16x16x32 operand roles (dgrad1)

  const int tps = (TS == 1) ? H / R : 1;
  const int wg = (int)blockIdx.x;
  const int b0 = (TS == 1) ? wg / tps : wg * TS;
  const int tile = (TS == 1) ? wg - b0 * tps : 0;
  const int row0 = tile * R;
  const int zh = 2 * halo;
  auto stamp = [&](int i) {
    if (a.c.stamps && tid == 0 && (!LV || k == k_last))   // level mode: the phases of the last step of the launch
      a.c.stamps[(long)blockIdx.x * 16 + i] = __builtin_readcyclecounter();
  };
  stamp(0);

  // ------------------------------------------------------------------ 0. dgrad3's weights -> registers (in flight during 1.)
  const int swid = __builtin_amdgcn_readfirstlane(wid);
  const int voffL = lane * 16;
  const __amdgpu_buffer_rsrc_t r3 = fb_rsrc(FBS(wg3), (unsigned)HID * g.K3p * 2);
  auto load_w3r = [&](const __amdgpu_buffer_rsrc_t& r, int j, bf16x8f (&f)[K3M]) {   // row tile (32 channels) swid*4 + j
    const int soff = (swid * 4 + j) * K3M * 1024;
#pragma unroll
    for (int ks = 0; ks < K3M; ++ks) f[ks] = fb_ldw(r, voffL, soff + ks * 1024);
  };
  auto load_w3 = [&](int j, bf16x8f (&f)[K3M]) { load_w3r(r3, j, f); };
  // ReLU mask words of this lane's pixel in every 32-pixel tile of the h-region (zero outside the image / the batch:
  // the gradients there are the zero padding of dgrad1): mk2[mt] = the 4 words of the wave's channel tiles in dgrad3
  // (tiles wid*4 .. wid*4+3), mk1[mt][i] = the word of tile i*4 + wid in dgrad2.  Requested first, used in the epilogues.
  auto load_masks = [&](const unsigned char* mp) {
    constexpr unsigned FB_OOB = 0x80000000u;
    const long msz = (long)s.B * HW * (HID / 8);
    const __amdgpu_buffer_rsrc_t rm1 = fb_rsrc(mp, (unsigned)msz), rm2 = fb_rsrc(mp + msz, (unsigned)msz);
#pragma unroll
    for (int mt = 0; mt < MT32; ++mt) {
      const int ph = mt * 32 + l32;
      const int sm = (TS == 1) ? 0 : (ph >> lhw), rem = (TS == 1) ? ph : (ph & ((1 << lhw) - 1));
      const int rr = rem >> lw, xx = rem & (W - 1);
      const int y = row0 + rr - halo, b = b0 + sm;
      const bool ok = ph < g.P_h && (unsigned)y < (unsigned)H && b < s.B;
      const int moff = ((b * HW + y * W + xx) * (HID / 8));
      mk2[mt] = __builtin_amdgcn_raw_buffer_load_b128(rm2, ok ? moff + wid * 16 : (int)FB_OOB, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        mk1[mt][i] = __builtin_amdgcn_raw_buffer_load_b32(rm1, ok ? moff + (i * 4 + wid) * 4 : (int)FB_OOB, 0, 0);
    }
  };
  // e^logs of the two ActNorm layers and dgrad3's first weight tiles of a step
  auto load_es_w3 = [&](const float* es2p, const float* es1p, const void* wg3p) {
    ereg[0] = es2p[tid]; ereg[1] = es2p[tid + 256]; ereg[2] = es1p[tid]; ereg[3] = es1p[tid + 256];
    const __amdgpu_buffer_rsrc_t rk = fb_rsrc(wg3p, (unsigned)HID * g.K3p * 2);
#pragma unroll
    for (int j = 0; j < W3R; ++j) load_w3r(rk, j, w3[j]);
  };

  // rows of the LDS image -> rows of a global [M][512] bf16 array, for the tile's own 64 pixels; 16-byte pieces, a
  // row's 64 pieces by consecutive threads
  auto save_rows = [&](__bf16* dstg) {
#pragma unroll
    for (int it = 0; it < NPX / 4; ++it) {
      const int p = tid + 256 * it, po = p >> 6, slot = p & 63;
      const int sm = po >> lrw, rem = po & ((1 << lrw) - 1), rr = rem >> lw, xx = rem & (W - 1);
      const int b = b0 + sm;
      if (b >= s.B) continue;
      const int ph = sm * g.RH * W + (rr + halo) * W + xx;
      const long m = (long)b * HW + (row0 + rr) * W + xx;
      *reinterpret_cast<u32x4f*>(dstg + m * HID + slot * 8) = *reinterpret_cast<const u32x4f*>(IMG + fb_img_off(ph, slot));
    }
  };

  // ------------------------------------------------------------------ 1. stage g_a3 (+2 halo rows), parameters, ReLU masks
  const bool fresh = !MULTI || k == k_first;   // (later steps of a run: requested before the epilogue of the step above)
  if (fresh) load_masks(c_masks);
  {
    constexpr int GMAX = 12;   // per-thread slots: TS*RZ*W*Cout <= 3072 (checked on the host)
    const int rowlen = W * Cout, per_s = g.RZ * rowlen, nx = TS * per_s;
    const int lo = (TS == 1) ? max(0, zh - row0) * rowlen : 0;                                     // elements inside the image
    const int hi_e = (TS == 1) ? min(g.RZ, H - row0 + zh) * rowlen : min(s.B - b0, TS) * per_s;    // ... and the batch
    const float* gb = c_ga3 + ((long)b0 * HW + (long)(row0 - zh) * W) * Cout;   // rows are contiguous: [pixel][Cout]
    const bool stage = !MULTI || k == k_first;   // (later steps of a run: the epilogue of the step above left both tiles in LDS)
    float greg[GMAX], zreg[GMAX];
    if (stage) {
#pragma unroll
      for (int i = 0; i < GMAX; ++i) greg[i] = gb[min(max(tid + 256 * i, lo), hi_e - 1)];   // clamped: unconditional loads
      if (LV) {
        // the tile's own rows of dL/dz: element e = (pixel of the tile, channel).  Second half: the coupling backward's
        // g_z2 (gz[k & 1]); first half: the gradient wrt the step's output -- g_out (+ g_ext) for the first step walked,
        // else what the launch above stored in the same rows of gz
        const int per_t = R * W * C, nz = min(s.B - b0, TS) * per_t;   // tile pixels are contiguous in [M]: rows row0.. of b0..
        const long e0 = ((long)b0 * HW + (long)row0 * W) * C;
        const float* zsrc = a.gz[k & 1] + e0;
        const bool top = k == a.lv.k_hi;
        const unsigned mC = 0xFFFFFFFFu / (unsigned)C + 1u;
#pragma unroll
        for (int i = 0; i < GMAX; ++i) {
          const int e = min(tid + 256 * i, nz - 1);
          const int c = e - (int)__umulhi((unsigned)e, mC) * C;
          float v = zsrc[e];
          if (top && c < Ch) {
            v = a.lv.g_out[e0 + e];
            if (a.lv.g_ext) v += a.lv.g_ext[e0 + e];
          }
          zreg[i] = (tid + 256 * i < nz) ? v : 0.f;
        }
      }
    }
    if (fresh) load_es_w3(FBS(es2), FBS(es1), FBS(wg3));
    if (stage) {
#pragma unroll
      for (int i = 0; i < GMAX; ++i) {
        const int e = tid + 256 * i;
        if (e < nx) G3[e] = (e >= lo && e < hi_e) ? greg[i] : 0.f;
      }
      if (LV) {
        float* GZl = reinterpret_cast<float*>(smem_fb + g.oGZ);   // [tile pixel][C + 1]
        const unsigned mC = 0xFFFFFFFFu / (unsigned)C + 1u;
#pragma unroll
        for (int i = 0; i < GMAX; ++i) {
          const int e = tid + 256 * i;
          if (e < NPX * C) {
            const int po = (int)__umulhi((unsigned)e, mC);
            GZl[po * (C + 1) + (e - po * C)] = zreg[i];
          }
        }
      }
    }
    Pe2[tid] = ereg[0]; Pe2[tid + 256] = ereg[1]; Pe1[tid] = ereg[2]; Pe1[tid + 256] = ereg[3];
    if (tid < 16) reinterpret_cast<unsigned*>(smem_fb + g.oZero + (tid >> 2) * 256)[tid & 3] = 0u;
  }
  __syncthreads();
  stamp(1);
  // im2col of g_a3 on the h-region (bf16 pairs); k = tap * Cout + co, the order of wd3's columns.  As in the fused forward
  // (nfk_fused2.hip): thread item = (pixel, tap, group of 4 channels), four items per thread with all four 16-byte LDS
  // reads before the first write; taps outside the image read the zero slot.  Cout is even: a group holds 4 or 2 channels.
  {
    const int npad = (g.K3p - g.K3) >> 1;   // zero pairs behind the last tap (K3 = 9 * Cout is even)
    for (int t = tid; t < MT32 * 32 * npad; t += 256) {
      const int ph = (int)__umulhi((unsigned)t, 0xFFFFFFFFu / (unsigned)npad + 1u), e = t - ph * npad;
      reinterpret_cast<unsigned*>(X3 + ph * g.X3S + g.K3 * 2)[e] = 0u;
    }
    const int G = (Cout + 3) >> 2, per_ph = 9 * G, nitems = MT32 * 32 * per_ph;
    const unsigned mPP = 0xFFFFFFFFu / (unsigned)per_ph + 1u, mGG = 0xFFFFFFFFu / (unsigned)G + 1u;
    const float* zslot = reinterpret_cast<const float*>(smem_fb + g.oZero);
    const bool al16 = (Cout & 3) == 0;   // a pixel's channels start on a 16-byte boundary
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
        const int rz = rr + halo + dy, xn = xx + dx;  // G3 has `zh` halo rows, the h-region `halo`
        const bool ok = ph < g.P_h && (unsigned)rz < (unsigned)g.RZ && (unsigned)xn < (unsigned)W;
        const float* src = ok ? G3 + ((sm * g.RZ + rz) * W + xn) * Cout + c0 : zslot;
        if (al16) v[u] = *reinterpret_cast<const f32x4*>(src);
        else {   // Cout % 4 == 2: 8-byte aligned pieces (the second one may belong to the next pixel: not written below)
          const f32x2 lo = *reinterpret_cast<const f32x2*>(src), hi2 = *reinterpret_cast<const f32x2*>(src + 2);
          v[u] = f32x4{lo[0], lo[1], hi2[0], hi2[1]};
        }
        d[u] = reinterpret_cast<unsigned*>(X3 + ph * g.X3S + (tap * Cout + c0) * 2);
        np[u] = (base + u * 256 < nitems) ? ((c0 + 2 < Cout) ? 2 : 1) : 0;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (np[u] > 0) d[u][0] = fb_pack(v[u][0], v[u][1]);
        if (np[u] > 1) d[u][1] = fb_pack(v[u][2], v[u][3]);
      }
    }
  }
  __syncthreads();
  stamp(2);
  if (c_x3save) {
    // the im2col rows of the tile's own pixels, bf16 [M][K3p]: X3[m][tap * Cout + co] = g_a3[m + d(tap)][co] is, with the
    // taps flipped, the P operand of conv3's weight gradient as a plain pixel-major GEMM (sum_m X3[m][(8 - tap) Cout + co]
    // h2[m][ci]) -- instead of nine shifted passes over the 512-wide h2
    __bf16* x3g = reinterpret_cast<__bf16*>(c_x3save);
    constexpr int PPR = K3M * 2;   // 16-byte pieces per row
    for (int p = tid; p < NPX * PPR; p += 256) {
      const int po = p / PPR, slot = p % PPR;
      const int sm = po >> lrw, rem = po & ((1 << lrw) - 1), rr = rem >> lw, xx = rem & (W - 1);
      const int b = b0 + sm;
      if (b >= s.B) continue;
      const int ph = sm * g.RH * W + (rr + halo) * W + xx;
      const long m = (long)b * HW + (row0 + rr) * W + xx;
      *reinterpret_cast<u32x4f*>(x3g + m * g.K3p + slot * 8) = *reinterpret_cast<const u32x4f*>(X3 + ph * g.X3S + slot * 16);
    }
  }

  // ReLU / ActNorm backward of one accumulator tile (32 channels x 32 pixels per m-tile) into the image:
  //   g_a = (h > 0) ? g e^logs : 0, bit (8*gi + 4*hi + r) of `word` = (h > 0) for channel n0 - 4*hi + 8*gi + 4*hi + r
  // `base[mt]`: byte address of this lane's pixel row (+ 8*hi), `sw[gi]`: swizzled offset of the 4-channel group gi,
  // n0: first channel of the tile (+ 4*hi), es: the layer's e^logs
  auto relu_an_bwd = [&](f32x16 (&acc)[MT32], char* (&base)[MT32], const int (&sw)[4], const float* es, int n0,
                         const unsigned (&word)[MT32]) {
#pragma unroll
    for (int gi = 0; gi < 4; ++gi) {
      const f32x4 e = *reinterpret_cast<const f32x4*>(es + n0 + 8 * gi);
#pragma unroll
      for (int mt = 0; mt < MT32; ++mt) {
        const unsigned m4 = word[mt] >> (8 * gi + 4 * hi);
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = ((m4 >> r) & 1u) ? acc[mt][4 * gi + r] * e[r] : 0.f;
        const u32x2f pk = {fb_pack(v[0], v[1]), fb_pack(v[2], v[3])};
        *reinterpret_cast<u32x2f*>(base[mt] + sw[gi]) = pk;
      }
    }
  };

  // ------------------------------------------------------------------ 2. dgrad3 -> g_a2 in the image
  const __amdgpu_buffer_rsrc_t r2 = fb_rsrc(FBS(wg2), (unsigned)HID * HID * 2);
  constexpr int NCH = 4, RD = 8;             // dgrad2: channel tiles per wave, W2^T register ring (RD k-steps x NCH)
  bf16x8f w2r[RD][NCH];
  int w2s[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) w2s[i] = (i * 4 + swid) * (HID / 16) * 1024;
  auto load_w2_part = [&](int part) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int f = part * 8 + e;
      w2r[f / NCH][f % NCH] = fb_ldw(r2, voffL, w2s[f % NCH] + (f / NCH) * 1024);
    }
  };
  {
    const char* x3_lane = X3 + l32 * g.X3S + hi * 16;
    char* imw[MT32];
#pragma unroll
    for (int mt = 0; mt < MT32; ++mt) imw[mt] = IMG + (mt * 32 + l32) * 1024 + 8 * hi;
    bf16x8f xb[2][MT32];
    auto rd1 = [&](int ks, bf16x8f (&f)[MT32]) {
#pragma unroll
      for (int mt = 0; mt < MT32; ++mt)
        f[mt] = *reinterpret_cast<const bf16x8f*>(x3_lane + mt * 32 * g.X3S + ks * 32);
    };
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f32x16 acc[MT32];
#pragma unroll
      for (int mt = 0; mt < MT32; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;
      rd1(0, xb[0]);
#pragma unroll
      for (int ks = 0; ks < K3M; ++ks) {
        if (ks + 1 < K3M) rd1(ks + 1, xb[(ks + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mt = 0; mt < MT32; ++mt)
          acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w3[j % W3R][ks], xb[ks & 1][mt], acc[mt], 0, 0, 0);
        if (W3R == 1 && j + 1 < 4) w3[0][ks] = fb_ldw(r3, voffL, ((swid * 4 + j + 1) * K3M + ks) * 1024);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (W3R > 1 && j + W3R < 4) load_w3(j + W3R, w3[j % W3R]);
      load_w2_part(j);
      // image address of channel group gi of tile (wid*4 + j): slot = wid*16 + j*4 + gi (see nfk_fused2.hip, conv1)
      int sw[4];
#pragma unroll
      for (int gi = 0; gi < 4; ++gi) sw[gi] = ((((j * 4 + gi) ^ l32) & 15) << 4) + wid * 256;
      unsigned word[MT32];
#pragma unroll
      for (int mt = 0; mt < MT32; ++mt) word[mt] = mk2[mt][j];
      relu_an_bwd(acc, imw, sw, Pe2, (wid * 4 + j) * 32 + 4 * hi, word);
    }
  }
  __syncthreads();
  stamp(3);
  save_rows(reinterpret_cast<__bf16*>(c_ga2));
  stamp(4);

  // ------------------------------------------------------------------ 3. dgrad2: ONE pass over k (see nfk_fused2.hip, conv2)
  const __amdgpu_buffer_rsrc_t r1 = fb_rsrc(FBS(wg1), (unsigned)NTD * 16 * 9 * HID * 2);
  constexpr int Kp1 = 9 * HID;
  constexpr int D1 = (NTD == 1 && MT32 == 2) ? 4 : 2;   // dgrad1's weight ring: chunks of 128 channels held ahead (it is
                                                        // requested while dgrad2's 4 x MT32 accumulator tiles are live)
  bf16x8f w1[D1][9][NTD];
  constexpr int EXMAX = LV ? 12 : 1, EWMAX = LV ? 9 : 1;   // level mode: NPX * C <= 3072, C * C <= 2304 (checked on the host)
  float exreg[EXMAX], ewreg[EWMAX];
  auto load_w1 = [&](int slot, int nc, int tap) {   // k-step (of 32) tap*16 + nc*4 + swid of every row tile
#pragma unroll
    for (int nt = 0; nt < NTD; ++nt)
      w1[slot][tap][nt] = fb_ldw(r1, voffL, (nt * (Kp1 / 32) + tap * 16 + nc * 4 + swid) * 1024);
  };
  {
    const int pc = (l32 & 15) ^ hi;
    const char* imr[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) imr[e] = IMG + l32 * 1024 + (((2 * e) ^ pc) << 4);
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
        f[mt] = *reinterpret_cast<const bf16x8f*>(imr[ks & 7] + mt * 32768 + (ks >> 3) * 256);
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
        if (ks + RD < 32) w2r[ks % RD][i] = fb_ldw(r2, voffL, w2s[i] + (ks + RD) * 1024);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    stamp(5);
    // dgrad1's weight ring: requested before the epilogue (a stretch of pure VALU work), it travels under it
#pragma unroll
    for (int c = 0; c < D1; ++c)
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) load_w1(c, c, tap);
    __syncthreads();   // every wave is done reading g_a2: g_a1 takes its place in the image (same layout)
    stamp(6);
    char* imw[MT32];
#pragma unroll
    for (int mt = 0; mt < MT32; ++mt) imw[mt] = IMG + (mt * 32 + l32) * 1024 + 8 * hi;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      // channel tile i*4 + wid: slots i*16 + wid*4 + gi  ->  byte i*256 + (((wid*4 + gi) ^ px) & 15) * 16
      int sw[4];
#pragma unroll
      for (int gi = 0; gi < 4; ++gi) sw[gi] = ((((wid * 4 + gi) ^ l32) & 15) << 4) + i * 256;
      unsigned word[MT32];
#pragma unroll
      for (int mt = 0; mt < MT32; ++mt) word[mt] = mk1[mt][i];
      relu_an_bwd(acc2[i], imw, sw, Pe1, (i * 4 + wid) * 32 + 4 * hi, word);
    }
  }
  __syncthreads();
  stamp(7);
  // level-mode epilogue scratch, in the im2col region (dead since dgrad3)
  const int XS = C + 1, Cp4 = (C + 3) & ~3;
  float* const XIl = reinterpret_cast<float*>(X3);   // [NPX][XS]  x_in tile
  float* const GXl = XIl + NPX * XS;                 // [NPX][XS]  g_x tile
  float* const Wl = GXl + NPX * XS;                  // [C][Cp4]   Wf as [in][out]
  float* const Al = Wl + C * Cp4;                    // [256 / C2][4][C2] column sums of the coupling backward
  float* const GZl = reinterpret_cast<float*>(smem_fb + g.oGZ);   // [NPX][XS] dL/dz on the tile
  // level mode: the epilogue's operands -- the step's input rows on the tile (x_in: the other factor of the partial
  // g_z x^T) and Wf -- are requested before dgrad1 and travel under it (21 more live values; requested at the start of the
  // epilogue they cost the level-1 step a full round trip behind the g_a1 rows' stores: 20 k cycles)
  constexpr bool EARLY_PF = true;
  auto prefetch_epi = [&]() {
    const float* xin = k == 0 ? a.lv.x0 : a.lv.zout + (long)(k - 1) * a.lv.zout_stride;
    const int ldxi = k == 0 ? a.lv.ldx0 : C;
    const int nzq = min(s.B - b0, TS) * R * W;   // valid pixels of the tile
    const long m0 = (long)b0 * HW + (long)row0 * W;
    const unsigned mC = 0xFFFFFFFFu / (unsigned)C + 1u;
    const float* Wfp = FBS(Wf);
#pragma unroll
    for (int i = 0; i < EXMAX; ++i) {
      const int e = min(tid + 256 * i, nzq * C - 1);
      const int po = (int)__umulhi((unsigned)e, mC);
      exreg[i] = xin[(m0 + po) * ldxi + (e - po * C)];
    }
#pragma unroll
    for (int i = 0; i < EWMAX; ++i) ewreg[i] = Wfp[min(tid + 256 * i, C * C - 1)];
  };
  if (LV && EARLY_PF) prefetch_epi();
  save_rows(reinterpret_cast<__bf16*>(c_ga1));
  stamp(8);

  // ------------------------------------------------------------------ 4. dgrad1 over the wave's K-slices of the g_a1 image
  f32x4 acc1[NTD][MT_O];
#pragma unroll
  for (int i = 0; i < NTD; ++i)
#pragma unroll
    for (int j = 0; j < MT_O; ++j) acc1[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  {
    int c_off[9][MT_O];
    const unsigned k4 = (unsigned)(wid * 4 + q4), zoff = (unsigned)g.oZero, ioff = (unsigned)g.oImg;
#pragma unroll
    for (int mo = 0; mo < MT_O; ++mo) {
      const int po = mo * 16 + r16;
      const int sm = po >> lrw, rem = po & ((1 << lrw) - 1), rr = (rem >> lw) + halo, xx = rem & (W - 1);
      const unsigned px0 = (unsigned)(sm * g.RH * W + rr * W + xx);
      unsigned rok[3], xok[3];
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
        c_off[tap][mo] = (int)(zoff ^ ((addr ^ zoff) & rok[dy + 1] & xok[dx + 1]));
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      bf16x8f fb[4][MT_O];
      auto rd3 = [&](int tap, bf16x8f (&f)[MT_O]) {
#pragma unroll
        for (int mo = 0; mo < MT_O; ++mo)
          f[mo] = *reinterpret_cast<const bf16x8f*>(smem_fb + c_off[tap][mo] + i * 256);
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
          for (int nt = 0; nt < NTD; ++nt)
            acc1[nt][mo] =
                __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1[i % D1][tap][nt], fb[tap & 3][mo], acc1[nt][mo], 0, 0, 0);
        if (i + D1 < 4) load_w1(i % D1, i + D1, tap);   // in place, D1 chunks ahead
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  __syncthreads();
  stamp(9);

  // ------------------------------------------------------------------ 5. cross-wave reduction -> g_z1 contribution [M][C/2]
  float* R3 = reinterpret_cast<float*>(IMG);     // [4 waves][64 px][NTD*16]   (the image is dead now)
  constexpr int NP = NTD * 16;
#pragma unroll
  for (int nt = 0; nt < NTD; ++nt)
#pragma unroll
    for (int mo = 0; mo < MT_O; ++mo)
      *reinterpret_cast<f32x4*>(R3 + ((wid * NPX + mo * 16 + r16) * NP + nt * 16 + q4 * 4)) = acc1[nt][mo];
  __syncthreads();
  const unsigned mCh = 0xFFFFFFFFu / (unsigned)Ch + 1u;
  if (!LV) {
    for (int idx = tid; idx < NPX * Ch; idx += 256) {
      const int po = (int)__umulhi((unsigned)idx, mCh), n = idx - po * Ch;
      const int sm = po >> lrw, rem = po & ((1 << lrw) - 1), rr = rem >> lw, xx = rem & (W - 1);
      const int b = b0 + sm;
      if (b >= s.B) continue;
      const float* p = R3 + po * NP + n;
      a.c.d1[((long)b * HW + (row0 + rr) * W + xx) * Ch + n] = (p[0] + p[NPX * NP]) + (p[2 * NPX * NP] + p[3 * NPX * NP]);
    }
    stamp(10);
  } else {
    // ---------------------------------------------------------------- 6. level mode: channel-mix backward on the tile
    // (the thread index passes through an empty asm once more: what the epilogue derives from it -- addresses, pixel and
    //  channel indices -- is then computed HERE; hoisted to the top of the step it was spilled there and came back through
    //  sixty scratch loads, each a round trip of its own behind the weight stream's loads)
    int tq = tid;
    asm volatile("" : "+v"(tq));
    const int C2n = C - Ch, Gn = 256 / C2n, G4 = Cp4 / 4;
    const bool below = k > a.lv.k_lo;   // the coupling backward of step k - 1 rides along
    CStep* spn = (CStep*)(a.lv.steps + (below ? k - 1 : k));
    if (MULTI) {
      // a run: the next step's masks, e^logs and first dgrad3 tiles travel under this epilogue.  Unconditional -- the last
      // step requests its own once more -- so that the carried registers are redefined HERE on every path (conditionally
      // assigned, their old values would stay live through the whole body of the loop)
      CStep* spq = (CStep*)(a.lv.steps + max(k - 1, k_last));
      load_masks(reinterpret_cast<const unsigned char*>(a.lv.masks) + (long)max(k - 1, k_last) * a.lv.mask_stride);
      load_es_w3(spq->es2, spq->es1, spq->wg3);
    }
    const int jn = tq % C2n, gn = tq / C2n;
    const long mt0 = (long)b0 * HW + (long)row0 * W;   // first pixel of the tile (its pixels are contiguous in [M])
    const int nzp = min(s.B - b0, TS) * R * W;         // ... and how many of them exist
    // operands of the coupling backward below: (shift | scale) rows and z2 of step k - 1, requested before the mat-vec
    constexpr int CPF = 8;   // pixels per thread: NPX / Gn <= 7 (checked on the host)
    // (everything the coupling loop reads from global memory is requested HERE: inside the loop, behind its stores, every
    //  load would wait for a full round trip -- the compiler must assume the stores alias it)
    float ch0[CPF], ch1[CPF], cz[CPF], glv[CPF], e3a = 0.f, e3b = 0.f;
    if (below && gn < Gn) {
      const float* __restrict__ hn = a.lv.hsave + (long)(k - 1) * a.lv.hsave_stride + mt0 * Cout;
      const float* __restrict__ zn = a.lv.zsave + (long)(k - 1) * a.lv.zsave_stride + mt0 * C;
      const float* __restrict__ es3n = spn->es3;
#pragma unroll
      for (int e = 0; e < CPF; ++e) {
        const int pp = min(gn + e * Gn, nzp - 1);
        if (s.additive) { ch0[e] = hn[pp * Cout + jn]; ch1[e] = 0.f; cz[e] = 0.f; glv[e] = 0.f; }
        else {
          ch0[e] = hn[pp * Cout + 2 * jn]; ch1[e] = hn[pp * Cout + 2 * jn + 1]; cz[e] = zn[pp * C + Ch + jn];
          glv[e] = a.lv.g_ld ? a.lv.g_ld[b0 + (pp >> lrw)] : 0.f;
        }
      }
      if (s.additive) e3a = es3n[jn];
      else { e3a = es3n[2 * jn]; e3b = es3n[2 * jn + 1]; }
    }
    if (!EARLY_PF) prefetch_epi();
    // g_z1 = (dL/dz1 of the step's output) + dgrad1, the four waves' partial sums in the order the stand-alone kernel adds them
    for (int idx = tq; idx < NPX * Ch; idx += 256) {
      const int po = (int)__umulhi((unsigned)idx, mCh), n = idx - po * Ch;
      const float* p = R3 + po * NP + n;
      const float d = (p[0] + p[NPX * NP]) + (p[2 * NPX * NP] + p[3 * NPX * NP]);
      if (po < nzp) GZl[po * XS + n] += d;
    }
    {   // x_in and Wf into the scratch
      const unsigned mC = 0xFFFFFFFFu / (unsigned)C + 1u;
#pragma unroll
      for (int i = 0; i < EXMAX; ++i) {
        const int e = tq + 256 * i;
        if (e < NPX * C) {
          const int po = (int)__umulhi((unsigned)e, mC);
          XIl[po * XS + (e - po * C)] = (e < nzp * C) ? exreg[i] : 0.f;
        }
      }
#pragma unroll
      for (int i = 0; i < EWMAX; ++i) {
        const int idx = tq + 256 * i;
        if (idx < C * C) {
          const int r = (int)__umulhi((unsigned)idx, mC);
          Wl[r * Cp4 + (idx - r * C)] = ewreg[i];   // Wm[r][q]: in = r, out = q (chanmix_body, trans)
        }
      }
    }
    __syncthreads();
    // g_x = Wf^T g_z: thread (pixel p of a pass, outputs 4 og .. 4 og + 3), input channels ascending (chanmix_body); two
    // passes' pixels side by side, so that a 16-byte row piece of Wf read from LDS feeds both (the phase is bound by LDS reads)
    {
      const int TP = 256 / G4 < NPX ? (256 / G4) : NPX;
      const int p = tq / G4, og = tq - p * G4, o = og * 4;
      float* gxo = (k == a.lv.k_lo) ? a.lv.g_x : nullptr;
      if (p < TP)
        for (int pp = p; pp < NPX; pp += 2 * TP) {
          const int p2 = pp + TP < NPX ? pp + TP : pp;   // (no second pixel: the first once more, result discarded)
          f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
          const float* xr = GZl + pp * XS;
          const float* xr2 = GZl + p2 * XS;
#pragma unroll 4
          for (int c = 0; c < C; ++c) {   // (C % 4 == 0: four channels' LDS reads in flight before their FMAs)
            const float xv = xr[c], xv2 = xr2[c];
            const f32x4 w = *reinterpret_cast<const f32x4*>(Wl + c * Cp4 + o);
#pragma unroll
            for (int r = 0; r < 4; ++r) { acc[r] += w[r] * xv; acc2[r] += w[r] * xv2; }
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) GXl[pp * XS + o + r] = acc[r];
          if (gxo && pp < nzp) *reinterpret_cast<f32x4*>(gxo + (mt0 + pp) * C + o) = acc;   // C % 4 == 0 (host check)
          if (p2 != pp) {
#pragma unroll
            for (int r = 0; r < 4; ++r) GXl[p2 * XS + o + r] = acc2[r];
            if (gxo && p2 < nzp) *reinterpret_cast<f32x4*>(gxo + (mt0 + p2) * C + o) = acc2;
          }
        }
    }
    stamp(10);
    // partial (g_z x_in^T, sum g_z) of this workgroup: one slab of the step's mix_part.  A small register-tiled product:
    // thread (to, tc) owns a T x T block of (o, c) pairs -- 2 T LDS reads per T^2 multiply-adds instead of 2 per 1; pixels
    // in ascending order per element, as the one-element-per-thread loop (and chanmix_body) adds them
    {
      const int E = C * C + C;
      float* part = ((CGrads*)(a.lv.grads + k))->mix_part + (long)blockIdx.x * E;
      auto tiled = [&](auto tt) {
        constexpr int T = decltype(tt)::value;
        const int nT = C / T, to = tq / nT, tc = tq - to * nT;
        if (to >= nT) return;
        float acc[T][T], cs[T];
#pragma unroll
        for (int i = 0; i < T; ++i) {
          cs[i] = 0.f;
#pragma unroll
          for (int j = 0; j < T; ++j) acc[i][j] = 0.f;
        }
        const float* gzp = GZl + to * T;
        const float* xip = XIl + tc * T;
#pragma unroll 4
        for (int q = 0; q < NPX; ++q) {
          float gv[T], xv[T];
#pragma unroll
          for (int i = 0; i < T; ++i) { gv[i] = gzp[q * XS + i]; xv[i] = xip[q * XS + i]; }
#pragma unroll
          for (int i = 0; i < T; ++i) {
            cs[i] += gv[i];
#pragma unroll
            for (int j = 0; j < T; ++j) acc[i][j] += gv[i] * xv[j];
          }
        }
#pragma unroll
        for (int i = 0; i < T; ++i) {
#pragma unroll
          for (int j = 0; j < T; ++j) part[(to * T + i) * C + tc * T + j] = acc[i][j];
          if (tc == 0) part[C * C + to * T + i] = cs[i];
        }
      };
      const int T = (C + 15) / 16;
      if (T == 1 && C * C <= 256) tiled(std::integral_constant<int, 1>{});
      else if (T == 2 && (C & 1) == 0 && (C / 2) * (C / 2) <= 256) tiled(std::integral_constant<int, 2>{});
      else if (T == 3 && C % 3 == 0 && (C / 3) * (C / 3) <= 256) tiled(std::integral_constant<int, 3>{});
      else {
        const unsigned mC = 0xFFFFFFFFu / (unsigned)C + 1u;
        for (int idx = tq; idx < E; idx += 256) {
          float sm_ = 0.f;
          if (idx < C * C) {
            const int o = (int)__umulhi((unsigned)idx, mC), c = idx - o * C;
#pragma unroll 8
            for (int q = 0; q < NPX; ++q) sm_ += GZl[q * XS + o] * XIl[q * XS + c];
          } else {
            const int o = idx - C * C;
#pragma unroll 8
            for (int q = 0; q < NPX; ++q) sm_ += GZl[q * XS + o];
          }
          part[idx] = sm_;
        }
      }
    }
    stamp(11);
    if (below) {
      // -------------------------------------------------------------- 7. coupling backward of step k - 1 on the tile
      __syncthreads();   // g_x is complete; the partial sums above are done with GZl (overwritten below in a run)
      stamp(12);
      float* __restrict__ gan = a.ga + (long)(k - 1) * a.ga_stride + mt0 * Cout;   // g_a3 of step k - 1 (the weight gradients read it too)
      float* __restrict__ gzn = a.gz[(k - 1) & 1] + mt0 * C;
      float sb0 = 0.f, sl0 = 0.f, sb1 = 0.f, sl1 = 0.f;
      float gz2n[CPF];
      if (gn < Gn) {
#pragma unroll
        for (int e = 0; e < CPF; ++e) {
          const int pp = gn + e * Gn;
          gz2n[e] = 0.f;
          if (pp < NPX) {
            float v0 = 0.f, v1 = 0.f;   // g_a3 entries (zero rows beyond the batch: dgrad's zero padding)
            if (pp < nzp) {
              const float g2 = GXl[pp * XS + Ch + jn];
              if (s.additive) {
                gz2n[e] = g2;
                v0 = g2 * e3a;
                sb0 += v0; sl0 += g2 * ch0[e];
              } else {
                const float gl = glv[e];
                const float shf = ch0[e], sr = ch1[e];
                const float sc = sigmoidf(sr + 2.f);
                const float gshift = g2 * sc;
                const float gsr = g2 * (cz[e] + shf) * sc * (1.f - sc) + gl * (1.f - sc);
                gz2n[e] = gshift;
                v0 = gshift * e3a; v1 = gsr * e3b;
                sb0 += v0; sl0 += gshift * shf; sb1 += v1; sl1 += gsr * sr;
              }
              if (s.additive) gan[pp * Cout + jn] = v0;
              else { gan[pp * Cout + 2 * jn] = v0; gan[pp * Cout + 2 * jn + 1] = v1; }
              if (!MULTI) gzn[pp * C + Ch + jn] = gz2n[e];
            }
            if (MULTI) {   // the next step's g_a3 tile (whole images, no halo rows: tile pixel order)
              if (s.additive) G3[pp * Cout + jn] = v0;
              else { G3[pp * Cout + 2 * jn] = v0; G3[pp * Cout + 2 * jn + 1] = v1; }
            }
          }
        }
        float* row = Al + (long)gn * 4 * C2n;
        row[jn] = sb0; row[C2n + jn] = sl0; row[2 * C2n + jn] = sb1; row[3 * C2n + jn] = sl1;
      }
      if (!MULTI) {   // dL/dz1 of step k - 1 = g_x's first half, beside g_z2 in the same rows
        for (int idx = tq; idx < nzp * Ch; idx += 256) {
          const int po = (int)__umulhi((unsigned)idx, mCh), n = idx - po * Ch;
          gzn[po * C + n] = GXl[po * XS + n];
        }
      }
      stamp(13);
      __syncthreads();
      if (MULTI) {
        if (gn < Gn) {
#pragma unroll
          for (int e = 0; e < CPF; ++e) {
            const int pp = gn + e * Gn;
            if (pp < NPX) GZl[pp * XS + Ch + jn] = gz2n[e];
          }
        }
        for (int idx = tq; idx < NPX * Ch; idx += 256) {
          const int po = (int)__umulhi((unsigned)idx, mCh), n = idx - po * Ch;
          GZl[po * XS + n] = GXl[po * XS + n];
        }
      }
      // one partial row per workgroup: (sum g_a3, 3 sum g_h h), groups added in fixed order
      float* prow = a.tailpart + ((long)(k - 1) * NFK_LEVEL_ROWS + blockIdx.x) * 2 * Cout;
      for (int jj = tq; jj < C2n; jj += 256) {
        float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
#pragma unroll 4
        for (int gg = 0; gg < Gn; ++gg) {
          const float* row = Al + (long)gg * 4 * C2n;
          t0 += row[jj]; t1 += row[C2n + jj]; t2 += row[2 * C2n + jj]; t3 += row[3 * C2n + jj];
        }
        if (s.additive) { prow[jj] = t0; prow[Cout + jj] = 3.f * t1; }
        else {
          prow[2 * jj] = t0; prow[2 * jj + 1] = t2;
          prow[Cout + 2 * jj] = 3.f * t1; prow[Cout + 2 * jj + 1] = 3.f * t3;
        }
      }
    }
    stamp(14);
    if (MULTI && k > k_last) __syncthreads();   // the next step's staging overwrites what the epilogue read
  }
  }   // steps of the level
#undef FBS
}

static int fb_ilog2_exact(int v) {
  if (v <= 0 || (v & (v - 1))) return -1;
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}

static bool fb_pick(const nfk_step2d& s, FBArgs* a) {
  if (!s.bf16 || s.hidden != 512 || s.Cout > 48 || s.Cout < 2 || (s.Cout & 1) || (s.C & 3) != 0 || s.C < 4) return false;
  if (!s.wg1 || !s.wg2 || !s.wg3) return false;   // needs the fragment-ordered dgrad weights
  const int lw = fb_ilog2_exact(s.W), lh = fb_ilog2_exact(s.H);
  if (lw < 0 || lh < 0) return false;
  const int HW = s.H * s.W;
  if (HW <= 64) { a->R = s.H; a->TS = 64 / HW; a->halo = 0; }
  else if (s.W <= 64 && s.H % (64 / s.W) == 0 && 64 / s.W < s.H) {
    a->R = 64 / s.W; a->TS = 1; a->halo = 1;
    if (a->R == 2 && s.W == 32 && fb_geom(s, 2, 1, 1).MT32 > 3) a->R = 1;   // rows of 32 pixels: one row per tile (nfk_fused2.hip)
    // 16-pixel rows whose four-row tile (96-pixel h-region) does not fit LDS (16 x 16 x 24, level 2 of 64 x 64 images:
    // 164 KB): two rows per tile, 32 output pixels, 64 computed
    else if (a->R == 4 && s.W == 16 && fb_geom(s, 4, 1, 1).total > 160 * 1024 - 512 - 8 * 1024) a->R = 2;
  }
  else return false;
  const FBGeom g = fb_geom(s, a->R, a->TS, a->halo);
  if (g.MT32 != 2 && g.MT32 != 3) return false;
  const bool row32 = a->TS == 1 && a->R * s.W == 32;
  if (a->TS == 1 && a->R * s.W != 64 &&
      !(row32 && ((g.MT32 == 3 && g.NTD == 1 && g.K3M == 8) || (g.MT32 == 2 && g.NTD == 1 && g.K3M == 16))))
    return false;
  if (g.total > 160 * 1024 - 512) return false;
  if ((size_t)4 * 64 * g.NTD * 16 * 4 > (size_t)g.MT32 * 32 * 1024) return false;   // R3 lives in the image
  if (a->TS * g.RZ * s.W * s.Cout > 12 * 256) return false;                          // prologue register slots
  if ((long)s.B * HW * 64 >= 0x7fffffffL) return false;                              // 32-bit buffer offsets (masks)
  a->lw = lw;
  a->lrw = fb_ilog2_exact(a->R * s.W);
  a->lhw = fb_ilog2_exact(g.RH * s.W);   // -1 when TS == 1 with halo rows: unused there
  if (a->lrw < 0 || (a->TS > 1 && a->lhw < 0)) return false;
  if (a->lhw < 0) a->lhw = 0;
  // instantiated for the shapes Glow produces
  return (g.MT32 == 3 && g.NTD == 1 && g.K3M == 8) || (g.MT32 == 2 && g.NTD == 1 && g.K3M == 16) ||
         (g.MT32 == 3 && g.NTD == 1 && g.K3M == 16) ||   // 16 x 16 x 24: level 2 of 64 x 64 images
         (g.MT32 == 2 && g.NTD == 2 && g.K3M == 28) || (g.MT32 == 2 && g.NTD == 1 && g.K3M == 8) ||
         (g.MT32 == 3 && g.NTD == 1 && g.K3M == 4);
}

template <int MT32, int NTD, int K3M, int MO = 4, bool LV = false, bool MULTI = false>
static int fb_launch_t(const FBArgs& a, const FBGeom& g, int grid, hipStream_t st) {
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(fusedb_step_bwd_kernel<MT32, NTD, K3M, MO, LV, MULTI>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512) != hipSuccess)
      return NFK_ERR_LAUNCH;
    attr = true;
  }
  hipLaunchKernelGGL((fusedb_step_bwd_kernel<MT32, NTD, K3M, MO, LV, MULTI>), dim3(grid), dim3(256), g.total, st, a);
  NFK_LAUNCH_CHECK();
  return NFK_OK;
}

static int g_small_tiles = 1;   // experiment knob (shared with the forward: nfk_tune NFK_TUNE_RUN_HALF_TILES)
void fusedb_set_small_tiles(int on) { g_small_tiles = on; }

int fusedb_grid(const nfk_step2d& s) {
  FBArgs a{};
  if (!fb_pick(s, &a)) return 0;
  return a.TS == 1 ? s.B * (s.H / a.R) : (s.B + a.TS - 1) / a.TS;
}

int fusedb_k3p(const nfk_step2d& s) { return (9 * s.Cout + 63) / 64 * 64; }

int fusedb_launch(const FusedBwdCall& c, hipStream_t st) {
  FBArgs a{};
  if (!c.masks || !fb_pick(c.s, &a)) return NFK_ERR_ARG;
  a.c = c;
  const nfk_step2d& s = c.s;
  const FBGeom g = fb_geom(s, a.R, a.TS, a.halo);
  const int grid = a.TS == 1 ? s.B * (s.H / a.R) : (s.B + a.TS - 1) / a.TS;
  const double M = (double)s.B * s.H * s.W;
  const int cls = NFK_PROF_FUSED_BWD + (M >= 8192 ? 0 : (M >= 2048 ? 1 : 2));
  prof_scope_begin(cls, 2.0 * M * (512.0 * 9 * s.Cout + 512.0 * 512 + 9.0 * 512 * (s.C / 2)),
                   M * (4.0 * s.Cout + 4.0 * (s.C / 2) + 2 * 512 * 2.0 + 2 * 64.0) + 2.0 * 512 * (9.0 * s.Cout + 512 + 9.0 * (s.C / 2)), st);
  int rc = NFK_ERR_ARG;
  // 4x4 images on a small grid (CIFAR level 3 at B = 64: 16 tiles): 16- / 32-pixel tiles, 64 / 32 workgroups
  const int HW = s.H * s.W;
  if (g_small_tiles && HW == 16 && a.halo == 0 && grid < 64 && g.NTD == 2 && g.K3M == 28) {
    FBArgs h = a;
    const int npx = grid < 32 ? 16 : 32;
    h.TS = npx / HW;
    h.lrw = fb_ilog2_exact(h.R * s.W);
    h.lhw = fb_ilog2_exact(h.R * s.W);
    const FBGeom gh = fb_geom(s, h.R, h.TS, h.halo);
    if (gh.MT32 == 1 && gh.total <= 160 * 1024 - 512 && (size_t)4 * npx * gh.NTD * 16 * 4 <= (size_t)32 * 1024) {
      rc = npx == 16 ? fb_launch_t<1, 2, 28, 1>(h, gh, s.B, st) : fb_launch_t<1, 2, 28, 2>(h, gh, (s.B + h.TS - 1) / h.TS, st);
      prof_scope_end(cls, st);
      return rc;
    }
  }
  if (a.TS == 1 && a.R * s.W == 32 && g.MT32 == 2) rc = fb_launch_t<2, 1, 16, 2>(a, g, grid, st);
  else if (a.TS == 1 && a.R * s.W == 32) rc = fb_launch_t<3, 1, 8, 2>(a, g, grid, st);
  else if (g.MT32 == 3 && g.NTD == 1 && g.K3M == 8) rc = fb_launch_t<3, 1, 8>(a, g, grid, st);
  else if (g.MT32 == 3 && g.NTD == 1 && g.K3M == 16) rc = fb_launch_t<3, 1, 16>(a, g, grid, st);
  else if (g.MT32 == 3 && g.NTD == 1 && g.K3M == 4) rc = fb_launch_t<3, 1, 4>(a, g, grid, st);
  else if (g.MT32 == 2 && g.NTD == 1 && g.K3M == 8) rc = fb_launch_t<2, 1, 8>(a, g, grid, st);
  else if (g.MT32 == 2 && g.NTD == 1 && g.K3M == 16) rc = fb_launch_t<2, 1, 16>(a, g, grid, st);
  else if (g.MT32 == 2 && g.NTD == 2 && g.K3M == 28) rc = fb_launch_t<2, 2, 28>(a, g, grid, st);
  prof_scope_end(cls, st);
  return rc;
}


// ------------------------------------------------------------------------------------------ level mode
// tile choice of a level: as fusedb_launch (16- / 32-pixel tiles for 4x4 images on a small grid), the same for every
// launch of the level -- the number of workgroups is the number of partial slabs / rows each step leaves
static bool fb_level_pick(const nfk_step2d& s, FBArgs* a, FBGeom* gp, int* mo, int* grid) {
  if (!fb_pick(s, a) || s.B > 256) return false;
  FBGeom g = fb_geom_lv(s, a->R, a->TS, a->halo);
  int npx = (a->TS == 1) ? a->R * s.W : 64, gr = a->TS == 1 ? s.B * (s.H / a->R) : (s.B + a->TS - 1) / a->TS;
  const int HW = s.H * s.W;
  if (g_small_tiles && HW == 16 && a->halo == 0 && gr < 64 && g.NTD == 2 && g.K3M == 28) {
    FBArgs h = *a;
    const int np = gr < 32 ? 16 : 32;
    h.TS = np / HW;
    h.lrw = fb_ilog2_exact(h.R * s.W);
    h.lhw = fb_ilog2_exact(h.R * s.W);
    const FBGeom gh = fb_geom_lv(s, h.R, h.TS, h.halo);
    if (gh.MT32 == 1 && (size_t)4 * np * gh.NTD * 16 * 4 <= (size_t)32 * 1024) {
      *a = h; g = gh; npx = np; gr = (s.B + h.TS - 1) / h.TS;
    }
  }
  const int C = s.C, C2 = C - C / 2;
  if (g.total > 160 * 1024 - 512 || gr > NFK_LEVEL_ROWS || gr < 1) return false;
  if (fb_epi_floats(C, npx) * 4 > (size_t)g.MT32 * 32 * g.X3S) return false;    // the epilogue's scratch: the im2col region
  if (npx * C > 12 * 256 || C * C > 9 * 256 || C2 < 1 || C2 > 256) return false;   // its register slots
  if ((npx + 256 / C2 - 1) / (256 / C2) > 8) return false;                        // pixels per thread of the coupling backward
  *gp = g; *mo = npx / 16; *grid = gr;
  return true;
}
int fusedb_level_multi_ok(const nfk_step2d& s) { return fusedb_level_ok(s) == 2; }
int fusedb_level_ok(const nfk_step2d& s) {
  FBArgs a{};
  FBGeom g;
  int mo, grid;
  if (!fb_level_pick(s, &a, &g, &mo, &grid)) return 0;
  return a.halo == 0 ? 2 : 1;
}
int fusedb_level_grid(const nfk_step2d& s) {
  FBArgs a{};
  FBGeom g;
  int mo, grid;
  return fb_level_pick(s, &a, &g, &mo, &grid) ? grid : 0;
}
template <bool MULTI>
static int fb_level_dispatch(const FBArgs& a, const FBGeom& g, int mo, int grid, hipStream_t st) {
  if (mo == 1 && g.MT32 == 1 && g.NTD == 2 && g.K3M == 28) return fb_launch_t<1, 2, 28, 1, true, MULTI>(a, g, grid, st);
  if (mo == 2 && g.MT32 == 1 && g.NTD == 2 && g.K3M == 28) return fb_launch_t<1, 2, 28, 2, true, MULTI>(a, g, grid, st);
  if (mo == 2 && g.MT32 == 3 && g.NTD == 1 && g.K3M == 8 && !MULTI) return fb_launch_t<3, 1, 8, 2, true, false>(a, g, grid, st);
  if (mo == 2 && g.MT32 == 2 && g.NTD == 1 && g.K3M == 16 && !MULTI) return fb_launch_t<2, 1, 16, 2, true, false>(a, g, grid, st);
  if (mo != 4) return NFK_ERR_ARG;
  if (g.MT32 == 2 && g.NTD == 1 && g.K3M == 8) return fb_launch_t<2, 1, 8, 4, true, MULTI>(a, g, grid, st);
  if (g.MT32 == 2 && g.NTD == 1 && g.K3M == 16) return fb_launch_t<2, 1, 16, 4, true, MULTI>(a, g, grid, st);
  if (g.MT32 == 2 && g.NTD == 2 && g.K3M == 28) return fb_launch_t<2, 2, 28, 4, true, MULTI>(a, g, grid, st);
  if (MULTI) return NFK_ERR_ARG;   // tiles with halo rows: one launch per step
  if (g.MT32 == 3 && g.NTD == 1 && g.K3M == 8) return fb_launch_t<3, 1, 8, 4, true, false>(a, g, grid, st);
  if (g.MT32 == 3 && g.NTD == 1 && g.K3M == 16) return fb_launch_t<3, 1, 16, 4, true, false>(a, g, grid, st);
  if (g.MT32 == 3 && g.NTD == 1 && g.K3M == 4) return fb_launch_t<3, 1, 4, 4, true, false>(a, g, grid, st);
  return NFK_ERR_ARG;
}
int fusedb_level_launch(const FusedLevelCall& c, hipStream_t st, int* grid_used) {
  FBArgs a{};
  FBGeom g;
  int mo, grid;
  const nfk_step2d& s = c.first;
  if (!fb_level_pick(s, &a, &g, &mo, &grid)) return NFK_ERR_ARG;
  a.c.s = s;
  a.lv = c.lv;
  a.gz[0] = c.gz[0]; a.gz[1] = c.gz[1]; a.tailpart = c.tailpart;
  a.ga = c.ga; a.ga_stride = c.ga_stride; a.o_ga2 = c.o_ga2; a.o_ga1 = c.o_ga1; a.o_x3 = c.o_x3;
  a.c.stamps = c.stamps;
  if (grid_used) *grid_used = grid;
  const double M = (double)s.B * s.H * s.W;
  const int cls = NFK_PROF_FUSED_BWD + (M >= 8192 ? 0 : (M >= 2048 ? 1 : 2));
  const double fl = 2.0 * M * (512.0 * 9 * s.Cout + 512.0 * 512 + 9.0 * 512 * (s.C / 2));
  const double by = M * (4.0 * s.Cout + 4.0 * (s.C / 2) + 2 * 512 * 2.0 + 2 * 64.0) + 2.0 * 512 * (9.0 * s.Cout + 512 + 9.0 * (s.C / 2));
  const int nwalk = c.run_hi - c.run_lo + 1;
  if (c.run_hi > c.lv.k_hi || c.run_lo < c.lv.k_lo || nwalk < 1) return NFK_ERR_ARG;
  if (c.lv.multi && a.halo == 0 && nwalk > 1) {
    if (c.run_hi != c.lv.k_hi || c.run_lo != c.lv.k_lo) return NFK_ERR_ARG;   // a run is the whole walk
    a.k_hi = c.lv.k_hi; a.k_lo = c.lv.k_lo;
    prof_scope_begin(cls, nwalk * fl, nwalk * by, st);
    const int rc = fb_level_dispatch<true>(a, g, mo, grid, st);
    prof_scope_end(cls, st);
    return rc;
  }
  for (int k = c.run_hi; k >= c.run_lo; --k) {
    a.k_hi = a.k_lo = k;
    prof_scope_begin(cls, fl, by, st);
    const int rc = fb_level_dispatch<false>(a, g, mo, grid, st);
    prof_scope_end(cls, st);
    if (rc != NFK_OK) return rc;
  }
  return NFK_OK;
}

}  // namespace nfk
