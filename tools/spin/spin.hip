// Measurement aid (not part of libodic_hip.so): resident blocks that idle for a given time, to price what a
// persistent decoder kernel's residency alone would cost the encode stream.  tools/persistent_probe.py builds
// and launches it.
#include <hip/hip_runtime.h>
#include <stdint.h>

extern "C" __global__ __launch_bounds__(768) void spin_kernel(long long cycles, int mode, float* sink) {
  const long long t0 = wall_clock64();
  float acc = threadIdx.x;
  while (wall_clock64() - t0 < cycles) {
    if (mode == 0) {
      __builtin_amdgcn_s_sleep(32);                 // parked: almost no issue slots
    } else {
      for (int i = 0; i < 64; ++i) acc = acc * 1.0001f + 0.5f;   // busy VALU
    }
  }
  if (acc == 12345.678f) sink[0] = acc;
}

extern "C" int spin_launch(int blocks, int threads, long long cycles, int mode, float* sink, void* stream) {
  hipLaunchKernelGGL(spin_kernel, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, cycles, mode, sink);
  return (int)hipGetLastError();
}
