// Round-6 probe behind DESIGN.md section 9 item 4 ("B <= 2 as one persistent launch per layer"): what does an all-to-all SEAM cost inside one launch on this
// part, against the kernel boundary it would replace?  A B = 1 layer has six dependent phases (QKV, attention, c_proj, up-projection, down-projection,
// combine); every seam is all-to-all (each output column needs the whole input row).  The probe runs P phases of a weight-streaming GEMV-shaped body
// (every workgroup streams `bytes_per_wg` of its own weight slice with non-temporal loads, reduces it and publishes 64 bytes that the next phase of every
// workgroup reads) in two forms:
//
//   launches   P plain kernel launches, one per phase (captured into a hipGraph by the host script)
//   persistent ONE launch, one workgroup per CU, phases separated by a grid barrier: write-through (sc1) publication, one monotonic counter per XCD class
//              + one top counter (the guide's "barrier-xcd" shape: XCD leaders meet on the top counter), one relaxed poll per workgroup with s_sleep, ONE
//              agent-scope acquire after the match.  Every spin is bounded (give-up code in `status`); all polled words are zeroed by a memset node first.
//
// extern "C" entry points for ctypes: grid_phase_launches / grid_phase_persistent.  Build + run: scripts/grid_phase_probe.py.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((address_space(1))) unsigned gu32;
typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float stream_slice(const float* __restrict__ w, long n4, int tid) {
  // n4 float4 per workgroup, 256 threads, 8 loads in flight per thread
  const f4* p = reinterpret_cast<const f4*>(w);
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  for (long i = tid; i < n4; i += 256 * 8) {
    f4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = (i + u * 256 < n4) ? __builtin_nontemporal_load(p + i + u * 256) : f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v[u];
  }
  return acc[0] + acc[1] + acc[2] + acc[3];
}

__device__ __forceinline__ float block_sum(float v, float* red, int tid) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  const float s = (red[0] + red[1]) + (red[2] + red[3]);
  __syncthreads();
  return s;
}

// one phase of the body: x_in[nwg * 16] (the previous phase's publication of every workgroup) -> this workgroup's 16 floats of x_out
__device__ __forceinline__ void phase_body(const float* __restrict__ w, long n4, const float* x_in, float* x_out, int nwg, int wg, int tid, float* red) {
  float s = stream_slice(w + (long)wg * n4 * 4, n4, tid);
  float xs = 0.f;
  for (int i = tid; i < nwg * 16; i += 256) xs += __hip_atomic_load(reinterpret_cast<const float*>(x_in) + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  s = block_sum(s + 1e-9f * xs, red, tid);
  if (tid < 16) __hip_atomic_store(x_out + wg * 16 + tid, s + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // write-through (sc1) publication
}

__global__ __launch_bounds__(256) void phase_kernel(const float* __restrict__ w, long n4, const float* x_in, float* x_out, int nwg) {
  __shared__ float red[4];
  phase_body(w, n4, x_in, x_out, nwg, blockIdx.x, threadIdx.x, red);
}

// counters: [0..7] per-XCD-class arrivals, [8] top arrivals, [9] generation; status[0] = give-up code
__global__ __launch_bounds__(256) void persistent_kernel(const float* __restrict__ w, long n4, float* xa, float* xb, int nwg, int phases, unsigned* counters,
                                                        unsigned* status) {
  __shared__ float red[4];
  const int tid = threadIdx.x, wg = blockIdx.x;
  const int cls = wg & 7;                                        // observed placement: block b runs on XCD b % 8 (speed only, never correctness)
  const unsigned per_cls = (unsigned)((nwg - cls + 7) / 8);
  float* bufs[2] = {xa, xb};
  for (int ph = 0; ph < phases; ++ph) {
    phase_body(w + (long)ph * nwg * n4 * 4, n4, bufs[ph & 1], bufs[(ph + 1) & 1], nwg, wg, tid, red);
    if (ph + 1 == phases) break;
    // ---- grid barrier (epoch ph + 1)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // this wave's sc1 stores have left
    __syncthreads();
    if (tid == 0) {
      const unsigned epoch = (unsigned)ph + 1;
      const unsigned a = __hip_atomic_fetch_add(counters + cls, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
      if (a == per_cls * epoch) {                                // last arriver of its class: meets the other classes on the top counter
        const unsigned t = __hip_atomic_fetch_add(counters + 8, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
        const unsigned ncls = (unsigned)(nwg < 8 ? nwg : 8);
        if (t == ncls * epoch) __hip_atomic_store(counters + 9, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      unsigned spins = 0;
      while (__hip_atomic_load(counters + 9, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > (1u << 22)) { __hip_atomic_store(status, 0xdead0000u + epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
  }
}

extern "C" int grid_phase_launches(const float* w, long n4, float* xa, float* xb, int nwg, int phases, void* stream) {
  float* bufs[2] = {xa, xb};
  for (int ph = 0; ph < phases; ++ph)
    hipLaunchKernelGGL(phase_kernel, dim3(nwg), dim3(256), 0, (hipStream_t)stream, w + (long)ph * nwg * n4 * 4, n4, bufs[ph & 1], bufs[(ph + 1) & 1], nwg);
  return (int)hipGetLastError();
}

extern "C" int grid_phase_persistent(const float* w, long n4, float* xa, float* xb, int nwg, int phases, unsigned* counters, unsigned* status, void* stream) {
  if (hipMemsetAsync(counters, 0, 16 * sizeof(unsigned), (hipStream_t)stream) != hipSuccess) return -1;      // re-initialised every call (a memset node under capture)
  hipLaunchKernelGGL(persistent_kernel, dim3(nwg), dim3(256), 0, (hipStream_t)stream, w, n4, xa, xb, nwg, phases, counters, status);
  return (int)hipGetLastError();
}
