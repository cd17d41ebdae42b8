// Student optimiser on flat fp32 buffers: global grad-norm clip + Adam in two launches.
// Restates torch.nn.utils.clip_grad_norm_(30) (PL gradient_clip_val, train.py:46) and
// torch.optim.Adam (pl_module.py:350-354) for one contiguous parameter arena.
#include "nfk_internal.h"

namespace nfk {

constexpr int SUMSQ_BLOCKS = 1024;

__global__ void __launch_bounds__(256) sumsq_stage1(const float* __restrict__ g, size_t n, float* __restrict__ partial) {
  __shared__ float sh[32];
  float s = 0.f;
  const size_t n4 = n / 4;
  const f32x4* g4 = reinterpret_cast<const f32x4*>(g);
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const f32x4 v = g4[i];
    s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
  }
  if (blockIdx.x == 0)
    for (size_t i = n4 * 4 + threadIdx.x; i < n; i += blockDim.x) s += g[i] * g[i];
  s = block_sum(s, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}
__global__ void __launch_bounds__(256) sumsq_stage2(const float* __restrict__ partial, int nblocks, float* __restrict__ out) {
  __shared__ float sh[32];
  float s = 0.f;
  for (int i = threadIdx.x; i < nblocks; i += blockDim.x) s += partial[i];
  s = block_sum(s, sh);
  if (threadIdx.x == 0) out[0] += s;
}

__global__ void step_inc_kernel(int* step) { step[0] += 1; }

// step_dev (optional): device-resident count of COMPLETED steps; when given, the bias corrections are
// computed on the device from it (so the launch can be replayed from a hipGraph); lr_host is then the raw lr.
__global__ void __launch_bounds__(256) adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, size_t n, const float* __restrict__ sumsq,
                                                   float clip, float gscale, float step_size, float beta1,
                                                   float beta2, float inv_sqrt_bc2, float eps, float wd,
                                                   const int* __restrict__ step_dev) {
  if (step_dev) {
    const int t = step_dev[0] + 1;
    const double bc1 = 1.0 - pow((double)beta1, (double)t), bc2 = 1.0 - pow((double)beta2, (double)t);
    step_size = (float)((double)step_size / bc1);
    inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
  }
  float coef = gscale;
  if (sumsq && clip > 0.f) {
    const float norm = sqrtf(sumsq[0]) * gscale;
    const float c = clip / (norm + 1e-6f);
    coef *= c < 1.f ? c : 1.f;
  }
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float gi = g[i] * coef;
    const float pi = p[i];
    if (wd != 0.f) gi += wd * pi;
    const float mi = beta1 * m[i] + (1.f - beta1) * gi;
    const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    p[i] = pi - step_size * mi / (sqrtf(vi) * inv_sqrt_bc2 + eps);
  }
}

}  // namespace nfk

using namespace nfk;
// ------------------------------------------------------------------------------------------
// Distillation loss (NFModel.loss, pl_module.py:246-300 for the hint-MSE + NLL branch):
//   kd[b] = 1/D * sum_{p<P} mean_i (s_p[b][i] - t_p[b][i])^2 ,  result = mean_b (w_nll * nll[b] + w_kd * kd[b]) [* weights[b]]
//   (P zipped pairs, D = len(student_kd_indices) >= P)
// two launches forward (per (sample, pair) sums, then one workgroup combines in fixed order), one backward.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) kd_pair_sums_kernel(nfk_kd_pairs a, float* __restrict__ part) {
  __shared__ float sh[32];
  const int b = blockIdx.x, p = blockIdx.y;
  const long n = a.per_sample[p];
  const float* s = a.s[p] + (long)b * n;
  const float* t = a.t[p] + (long)b * n;
  float acc = 0.f;
  if ((n & 3) == 0 && ((reinterpret_cast<size_t>(s) | reinterpret_cast<size_t>(t)) & 15) == 0) {
    // 16-byte loads, four per operand in flight (a 3072-element hint is three passes of the block: one round trip)
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4* s4 = reinterpret_cast<const f4*>(s);
    const f4* t4 = reinterpret_cast<const f4*>(t);
    const long n4 = n >> 2;
    for (long i0 = threadIdx.x; i0 < n4; i0 += 4 * 256) {
      f4 sv[4], tv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long i = i0 + u * 256;
        sv[u] = i < n4 ? s4[i] : f4{0.f, 0.f, 0.f, 0.f};
        tv[u] = i < n4 ? t4[i] : f4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const f4 d = sv[u] - tv[u];
        acc += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
      }
    }
  } else {
    for (long i = threadIdx.x; i < n; i += 256) {
      const float d = s[i] - t[i];
      acc += d * d;
    }
  }
  acc = block_sum(acc, sh);
  if (threadIdx.x == 0) part[p * a.B + b] = acc / (float)n;
}
__global__ void __launch_bounds__(256) kd_combine_kernel(nfk_kd_pairs a, const float* __restrict__ part,
                                                         const float* __restrict__ nll, float w_nll, float w_kd,
                                                         float* __restrict__ kd_out, float* __restrict__ out3) {
  __shared__ float sh[32];
  float r = 0.f, sn = 0.f, sk = 0.f;
  for (int b = threadIdx.x; b < a.B; b += 256) {
    float kd = 0.f;
    for (int p = 0; p < a.n; ++p) kd += part[p * a.B + b];
    kd /= (float)(a.divisor > 0 ? a.divisor : (a.n > 0 ? a.n : 1));
    kd_out[b] = kd;
    const float per = w_nll * nll[b] + w_kd * kd;
    r += a.weights ? per * a.weights[b] : per;
    sn += nll[b];
    sk += kd;
  }
  r = block_sum(r, sh);
  sn = block_sum(sn, sh);
  sk = block_sum(sk, sh);
  if (threadIdx.x == 0) { out3[0] = r / a.B; out3[1] = sn / a.B; out3[2] = sk / a.B; }
}
// g_s[p][b][i] = g_result * w_kd * 2 / (B * P * n_p) * (s - t) ;  g_nll[b] = g_result * w_nll / B
__global__ void __launch_bounds__(256) kd_bwd_kernel(nfk_kd_pairs a, const float* __restrict__ g_result, float w_nll,
                                                     float w_kd, nfk_kd_grads g) {
  const int b = blockIdx.x, p = blockIdx.y;
  const float gr = a.weights ? g_result[0] * a.weights[b] : g_result[0];
  if (p == 0 && threadIdx.x == 0 && g.g_nll) g.g_nll[b] = gr * w_nll / (float)a.B;
  const long n = a.per_sample[p];
  const float coef = gr * w_kd * 2.f / ((float)a.B * (float)(a.divisor > 0 ? a.divisor : a.n) * (float)n);
  const float* s = a.s[p] + (long)b * n;
  const float* t = a.t[p] + (long)b * n;
  float* o = g.g_s[p] + (long)b * n;
  if ((n & 3) == 0 && ((reinterpret_cast<size_t>(s) | reinterpret_cast<size_t>(t) | reinterpret_cast<size_t>(o)) & 15) == 0) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4* s4 = reinterpret_cast<const f4*>(s);
    const f4* t4 = reinterpret_cast<const f4*>(t);
    f4* o4 = reinterpret_cast<f4*>(o);
    const long n4 = n >> 2;
    for (long i0 = threadIdx.x; i0 < n4; i0 += 4 * 256) {
      f4 sv[4], tv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long i = i0 + u * 256;
        sv[u] = i < n4 ? s4[i] : f4{0.f, 0.f, 0.f, 0.f};
        tv[u] = i < n4 ? t4[i] : f4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long i = i0 + u * 256;
        if (i < n4) o4[i] = (sv[u] - tv[u]) * coef;
      }
    }
    return;
  }
  for (long i = threadIdx.x; i < n; i += 256) o[i] = coef * (s[i] - t[i]);
}

// ------------------------------------------------------------------------------------------
// Input / output image pipeline on the device (data/src/utils.py:7-25 after torchvision's ToTensor):
//   preprocess : uint8 HWC -> fp32 NCHW, x = ((u / 255) * 255) / 256 - 0.5 (the reference's own sequence of fp32
//                operations, so the result is bit-identical), optional per-sample horizontal flip
//                (transforms.RandomHorizontalFlip, data/src/cifar10.py:17-20)
//   postprocess: fp32 NCHW -> uint8 HWC, clamp(x, -0.5, 0.5) + 0.5, * 255, clamp to [0, 255], truncate
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) preprocess_u8_kernel(const unsigned char* __restrict__ u,
                                                            const unsigned char* __restrict__ flip,
                                                            float* __restrict__ out, int B, int H, int W, int C) {
  const long total = (long)B * C * H * W;
  for (long t = blockIdx.x * 256L + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
    const int x = t % W;
    long r = t / W;
    const int y = r % H; r /= H;
    const int c = r % C;
    const int b = r / C;
    const int xs = (flip && flip[b]) ? W - 1 - x : x;
    float v = (float)u[(((long)b * H + y) * W + xs) * C + c];
    v = v / 255.f;
    v = v * 255.f;
    out[t] = v / 256.f - 0.5f;
  }
}
__global__ void __launch_bounds__(256) postprocess_u8_kernel(const float* __restrict__ x, unsigned char* __restrict__ u,
                                                             int B, int H, int W, int C) {
  const long total = (long)B * C * H * W;
  for (long t = blockIdx.x * 256L + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
    const int c = t % C;
    long r = t / C;
    const int xx = r % W; r /= W;
    const int y = r % H;
    const int b = r / H;
    float v = x[(((long)b * C + c) * H + y) * W + xx];
    v = fminf(fmaxf(v, -0.5f), 0.5f) + 0.5f;
    v = fminf(fmaxf(v * 255.f, 0.f), 255.f);
    u[t] = (unsigned char)v;
  }
}

extern "C" {
int nfk_sumsq(const float* g, size_t n, float* partial, float* sumsq, void* stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if ((reinterpret_cast<uintptr_t>(g) & 15) != 0) return NFK_ERR_ARG;
  size_t want = (n / 4 + 255) / 256;
  const int blocks = (int)(want < 1 ? 1 : (want > SUMSQ_BLOCKS ? SUMSQ_BLOCKS : want));
  hipLaunchKernelGGL(sumsq_stage1, dim3(blocks), dim3(256), 0, st, g, n, partial);
  NFK_LAUNCH_CHECK();
  hipLaunchKernelGGL(sumsq_stage2, dim3(1), dim3(256), 0, st, partial, blocks, sumsq);
  NFK_LAUNCH_CHECK();
  return NFK_OK;
}
int nfk_adam_step(float* p, const float* g, float* m, float* v, size_t n, const float* sumsq, float clip,
                  float gscale, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                  void* stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (step < 1) return NFK_ERR_ARG;
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  size_t want = (n + 255) / 256;
  const int blocks = (int)(want < 1 ? 1 : (want > 4096 ? 4096 : want));
  hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, st, p, g, m, v, n, sumsq, clip, gscale,
                     (float)(lr / bc1), beta1, beta2, (float)(1.0 / sqrt(bc2)), eps, weight_decay,
                     (const int*)nullptr);
  NFK_LAUNCH_CHECK();
  return NFK_OK;
}
int nfk_adam_step_dev(float* p, const float* g, float* m, float* v, size_t n, const float* sumsq, float clip,
                      float gscale, float lr, float beta1, float beta2, float eps, float weight_decay,
                      int* step_dev, void* stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (!step_dev) return NFK_ERR_ARG;
  size_t want = (n + 255) / 256;
  const int blocks = (int)(want < 1 ? 1 : (want > 4096 ? 4096 : want));
  hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, st, p, g, m, v, n, sumsq, clip, gscale, lr, beta1,
                     beta2, 1.f, eps, weight_decay, (const int*)step_dev);
  NFK_LAUNCH_CHECK();
  hipLaunchKernelGGL(step_inc_kernel, dim3(1), dim3(1), 0, st, step_dev);
  NFK_LAUNCH_CHECK();
  return NFK_OK;
}
int nfk_kd_loss_fwd(const nfk_kd_pairs* pairs, const float* nll, float w_nll, float w_kd, float* part, float* kd_out,
                    float* out3, void* stream) {
  if (pairs->B <= 0 || pairs->n < 0 || pairs->n > NFK_KD_MAX_PAIRS || (pairs->divisor && pairs->divisor < pairs->n))
    return NFK_ERR_ARG;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (pairs->n > 0) {
    hipLaunchKernelGGL(kd_pair_sums_kernel, dim3(pairs->B, pairs->n), dim3(256), 0, st, *pairs, part);
    NFK_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(kd_combine_kernel, dim3(1), dim3(256), 0, st, *pairs, part, nll, w_nll, w_kd, kd_out, out3);
  NFK_LAUNCH_CHECK();
  return NFK_OK;
}
int nfk_kd_loss_bwd(const nfk_kd_pairs* pairs, const float* g_result, float w_nll, float w_kd,
                    const nfk_kd_grads* grads, void* stream) {
  if (pairs->B <= 0 || pairs->n <= 0 || pairs->n > NFK_KD_MAX_PAIRS) return NFK_ERR_ARG;
  hipLaunchKernelGGL(kd_bwd_kernel, dim3(pairs->B, pairs->n), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     *pairs, g_result, w_nll, w_kd, *grads);
  NFK_LAUNCH_CHECK();
  return NFK_OK;
}
int nfk_preprocess_u8(const unsigned char* u8_hwc, const unsigned char* flip, float* out_nchw, int B, int H, int W, int C,
                      void* stream) {
  const long total = (long)B * C * H * W;
  if (total <= 0) return NFK_OK;
  long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(preprocess_u8_kernel, dim3((int)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), u8_hwc,
                     flip, out_nchw, B, H, W, C);
  NFK_LAUNCH_CHECK();
  return NFK_OK;
}
int nfk_postprocess_u8(const float* x_nchw, unsigned char* u8_hwc, int B, int H, int W, int C, void* stream) {
  const long total = (long)B * C * H * W;
  if (total <= 0) return NFK_OK;
  long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(postprocess_u8_kernel, dim3((int)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x_nchw,
                     u8_hwc, B, H, W, C);
  NFK_LAUNCH_CHECK();
  return NFK_OK;
}
}
