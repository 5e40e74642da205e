// What does one device-side producer -> consumer hop cost when consecutive kernels of a dependent chain run on two streams and the consumer WAITS ON A
// FLAG instead of on the kernel boundary?  (LABNOTES.md section 8: the legal form of cross-kernel overlap for the small-batch sampler chain.)
//   chain of NK kernels; kernel i (NWG workgroups of 256 threads): [optional: stream `wbytes` of private "weights" into registers first - the part a
//   real consumer could prefetch], spin until counter[i-1] == NWG, acquire, read the previous kernel's 64-KiB output, add, write its own, release,
//   counter[i] += 1 per workgroup.
//   MODE 0: one stream, plain kernel boundaries (no flags)      MODE 1: two alternating streams + flags, agent-scope fences (__threadfence)
//   MODE 2: two streams + flags, sc1 stores / sc1 loads of the data instead of fences
// Prints microseconds per hop.   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/chain_hop_probe scripts/probe/chain_hop_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f4v __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void hop(const float* __restrict__ in, float* __restrict__ out, const int* __restrict__ wait_ctr, int target,
                                           int* __restrict__ sig_ctr, const f4v* __restrict__ weights, long wvec_per_wg, float* __restrict__ sink) {
  // (1) the input-independent part: this workgroup's slice of the weights
  f4v acc = {0.f, 0.f, 0.f, 0.f};
  const f4v* w = weights + (long)blockIdx.x * wvec_per_wg;
  for (long i = threadIdx.x; i < wvec_per_wg; i += 256) { const f4v v = __builtin_nontemporal_load(w + i); acc += v; }
  // (2) wait for the producer
  if (MODE != 0 && wait_ctr) {
    if (threadIdx.x == 0) {
      int it = 0;
      while (__hip_atomic_load(wait_ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && ++it < (1 << 22)) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
    if (MODE == 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  // (3) the dependent part: every workgroup reads the WHOLE 64-KiB activation (like a skinny GEMM reads all of A) and writes its 1/NWG share
  float s = acc.x + acc.y + acc.z + acc.w;
  const int n = 16384;                                              // floats
  for (int i = threadIdx.x; i < n; i += 256) {
    float v;
    if (MODE == 2) asm volatile("global_load_dword %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(in + i) : "memory");
    else v = in[i];
    s += v * 1e-6f;
  }
  const int per = n / gridDim.x;
  for (int i = threadIdx.x; i < per; i += 256) {
    const float v = in[blockIdx.x * per + i] + 1.0f + s * 0.f;
    if (MODE == 2) asm volatile("global_store_dword %0, %1, off sc1" ::"v"(out + blockIdx.x * per + i), "v"(v) : "memory");
    else out[blockIdx.x * per + i] = v;
  }
  if (s == 12345.678f) sink[0] = s;
  // (4) signal
  if (MODE != 0 && sig_ctr) {
    if (MODE == 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(sig_ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

template <int MODE>
static double run(int NK, int NWG, long wbytes_per_kernel, int reps) {
  float *a, *b, *sink; int* ctr; f4v* w;
  CK(hipMalloc(&a, 65536)); CK(hipMalloc(&b, 65536)); CK(hipMalloc(&sink, 256)); CK(hipMalloc(&ctr, (NK + 1) * 4));
  const long wvec = wbytes_per_kernel / 16 / NWG;
  const long wtot = (long)NK * NWG * wvec;
  CK(hipMalloc(&w, (wtot > 0 ? wtot : 1) * 16)); CK(hipMemset(w, 0, (wtot > 0 ? wtot : 1) * 16));
  CK(hipMemset(a, 0, 65536)); CK(hipMemset(b, 0, 65536));
  hipStream_t s0, s1; CK(hipStreamCreate(&s0)); CK(hipStreamCreate(&s1));
  hipEvent_t e0, e1, fork, join; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&fork)); CK(hipEventCreate(&join));
  double best = 1e30;
  for (int r = 0; r < reps + 1; ++r) {
    CK(hipMemsetAsync(ctr, 0, (NK + 1) * 4, s0));
    CK(hipStreamSynchronize(s0));
    CK(hipEventRecord(e0, s0));
    if (MODE != 0) { CK(hipEventRecord(fork, s0)); CK(hipStreamWaitEvent(s1, fork, 0)); }
    for (int i = 0; i < NK; ++i) {
      hipStream_t st = (MODE != 0 && (i & 1)) ? s1 : s0;
      const float* in = (i & 1) ? b : a; float* out = (i & 1) ? a : b;
      hipLaunchKernelGGL(hop<MODE>, dim3(NWG), dim3(256), 0, st, in, out, i ? ctr + i - 1 : nullptr, NWG, ctr + i, w + (long)i * NWG * wvec, wvec, sink);
    }
    if (MODE != 0) { CK(hipEventRecord(join, s1)); CK(hipStreamWaitEvent(s0, join, 0)); }
    CK(hipEventRecord(e1, s0));
    CK(hipStreamSynchronize(s0));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (r > 0 && ms < best) best = ms;
  }
  // correctness of the chain: every element went through NK increments
  std::vector<float> h(16384);
  CK(hipMemcpy(h.data(), (NK & 1) ? b : a, 65536, hipMemcpyDeviceToHost));
  int bad = 0;
  for (float v : h) bad += (v != (float)NK);
  if (bad) printf("   (MODE %d: %d of 16384 outputs WRONG: %.1f instead of %d)\n", MODE, bad, h[0], NK);
  return best * 1e3 / NK;
}

int main() {
  const int NK = 200, reps = 5;
  for (int NWG : {64, 256}) {
    for (long wb : {0L, 4L << 20, 16L << 20}) {
      printf("NWG %3d  weights/kernel %5.1f MB :  boundaries %6.2f us/hop   flags+fences %6.2f   flags+sc1 %6.2f\n", NWG, wb / 1048576.0,
             run<0>(NK, NWG, wb, reps), run<1>(NK, NWG, wb, reps), run<2>(NK, NWG, wb, reps));
    }
  }
  return 0;
}
