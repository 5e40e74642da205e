// Co-execution probe kernels (scripts/coexec_probe.py): what can run UNDER a chain of GEMM kernels from a second stream?
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC scripts/probe/coexec_probe.hip -o scripts/probe/libcoexec_probe.so
#include <hip/hip_runtime.h>
#include <stdint.h>

// pure ALU: `iters` dependent FMAs per thread, no memory traffic (8 VGPRs)
__global__ void spin_alu_kernel(float* out, long iters) {
  float a = threadIdx.x * 1e-3f, b = 1.0001f;
  for (long i = 0; i < iters; ++i) a = a * b + 1e-7f;
  if (a == 123.456f) out[0] = a;
}
// pure stream: grid-stride read of n float4 (non-temporal), sum kept alive
__global__ void stream_read_kernel(const float* __restrict__ p, long n4, float* out) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  f4 acc = {0, 0, 0, 0};
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) acc += __builtin_nontemporal_load(reinterpret_cast<const f4*>(p) + i);
  if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) out[0] = acc[0];
}
// read-modify-write stream (what an optimizer pass does): p[i] = p[i] * 1.0000001f
__global__ void stream_rmw_kernel(float* __restrict__ p, long n4) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    f4 v = __builtin_nontemporal_load(reinterpret_cast<const f4*>(p) + i);
    v *= 1.0000001f;
    __builtin_nontemporal_store(v, reinterpret_cast<f4*>(p) + i);
  }
}
extern "C" void spin_alu(int blocks, int threads, long iters, float* out, void* stream) {
  hipLaunchKernelGGL(spin_alu_kernel, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, out, iters);
}
extern "C" void stream_read(const float* p, long n4, int blocks, float* out, void* stream) {
  hipLaunchKernelGGL(stream_read_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, n4, out);
}
extern "C" void stream_rmw(float* p, long n4, int blocks, void* stream) {
  hipLaunchKernelGGL(stream_rmw_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, n4);
}
