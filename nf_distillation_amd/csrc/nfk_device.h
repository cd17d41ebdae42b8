// Device-side helpers shared by the nfk kernels (gfx950 / CDNA4 only, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/nfk.h"





typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace nfk {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// Sum over the whole workgroup (blockDim.x multiple of 64, <= 1024).  Result valid in every
// thread.  `scratch` must hold >= 16 floats.  Deterministic: fixed tree order.
__device__ __forceinline__ float block_sum(float v, float* scratch) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
  v = wave_sum(v);
  __syncthreads();
  if (lane == 0) scratch[wid] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < nw; ++i) t += scratch[i];
  return t;
}

__device__ __forceinline__ float sigmoidf(float t) { return 1.f / (1.f + __expf(-t)); }
// log(sigmoid(t)) without cancellation: -log1p(exp(-t)) for t>=0, t - log1p(exp(t)) otherwise.
__device__ __forceinline__ float log_sigmoidf(float t) {
  return t >= 0.f ? -log1pf(expf(-t)) : t - log1pf(expf(t));
}

// Sum of S split-K slabs at one element, fixed order.  Loads are issued 16 (then 8, then 4) at a time so their
// latencies overlap instead of forming an S-deep dependent chain.
__device__ __forceinline__ float slab_sum(const float* __restrict__ a3, int S, long stride, long off) {
  const float* p = a3 + off;
  float v = 0.f;
  int s = 0;
  for (; s + 16 <= S; s += 16) {   // two groups of 8 in flight; the adds keep the order of the one-group loop below
    float t[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) t[e] = p[e * stride];
    v += ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
    v += ((t[8] + t[9]) + (t[10] + t[11])) + ((t[12] + t[13]) + (t[14] + t[15]));
    p += 16 * stride;
  }
  for (; s + 8 <= S; s += 8) {
    const float t0 = p[0], t1 = p[stride], t2 = p[2 * stride], t3 = p[3 * stride];
    const float t4 = p[4 * stride], t5 = p[5 * stride], t6 = p[6 * stride], t7 = p[7 * stride];
    v += ((t0 + t1) + (t2 + t3)) + ((t4 + t5) + (t6 + t7));
    p += 8 * stride;
  }
  if (s + 4 <= S) {
    const float t0 = p[0], t1 = p[stride], t2 = p[2 * stride], t3 = p[3 * stride];
    v += (t0 + t1) + (t2 + t3);
    p += 4 * stride;
    s += 4;
  }
  for (; s < S; ++s) { v += p[0]; p += stride; }
  return v;
}

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

#define NFK_LAUNCH_CHECK()                                   \
  do {                                                       \
    if (hipGetLastError() != hipSuccess) return NFK_ERR_LAUNCH; \
  } while (0)

}  // namespace nfk
