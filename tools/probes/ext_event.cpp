// Probe: does this HIP runtime accept an EXTERNAL event-record node inside a stream capture, and does a wait issued on
// another stream after hipGraphLaunch see that launch's record?   build: hipcc -O2 ext_event.cpp -o ext_event
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void spin(int* p, int n) { for (int i = 0; i < n; ++i) atomicAdd(p, 1); }
__global__ void copy(const int* p, int* q) { *q = *p; }
int main() {
  hipStream_t s, c;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&c, hipStreamNonBlocking));
  int *a, *b;
  CK(hipMalloc(&a, 4)); CK(hipMalloc(&b, 4));
  hipEvent_t ev;
  CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  for (int mode = 0; mode < 3; ++mode) {
    hipStreamCaptureMode m = mode == 0 ? hipStreamCaptureModeGlobal : (mode == 1 ? hipStreamCaptureModeThreadLocal : hipStreamCaptureModeRelaxed);
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, m));
    hipLaunchKernelGGL(spin, dim3(1), dim3(1), 0, s, a, 2000000);
    hipError_t r = hipEventRecordWithFlags(ev, s, hipEventRecordExternal);
    printf("capture mode %d: hipEventRecordWithFlags(external) -> %s\n", mode, hipGetErrorString(r));
    hipLaunchKernelGGL(spin, dim3(1), dim3(1), 0, s, a, 10);
    hipError_t e2 = hipStreamEndCapture(s, &g);
    printf("  end capture -> %s\n", hipGetErrorString(e2));
    (void)hipGetLastError();
    if (r != hipSuccess || e2 != hipSuccess) continue;
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int it = 0; it < 3; ++it) {
      CK(hipMemsetAsync(a, 0, 4, s));
      CK(hipGraphLaunch(ge, s));
      CK(hipStreamWaitEvent(c, ev, 0));
      hipLaunchKernelGGL(copy, dim3(1), dim3(1), 0, c, a, b);
      CK(hipDeviceSynchronize());
      int hb; CK(hipMemcpy(&hb, b, 4, hipMemcpyDeviceToHost));
      printf("  replay %d: value seen by the waiting stream = %d (>= 2000000 means the wait saw this launch's record)\n", it, hb);
    }
  }
  return 0;
}
