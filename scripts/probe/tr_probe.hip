// Probe: semantics of ds_read_b64_tr_b16 on gfx950.  LDS holds element index as value (u16); every lane supplies its own address.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void probe(const int* lane_addr_elems, uint16_t* out) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  const uint32_t addr = (uint32_t)(uintptr_t)(&lds[0]) + lane_addr_elems[threadIdx.x] * 2;
  uint2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  out[threadIdx.x * 4 + 0] = v.x & 0xffff; out[threadIdx.x * 4 + 1] = v.x >> 16;
  out[threadIdx.x * 4 + 2] = v.y & 0xffff; out[threadIdx.x * 4 + 3] = v.y >> 16;
}
int main() {
  int h[64]; uint16_t r[256];
  int* d; uint16_t* o;
  hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(r));
  for (int variant = 0; variant < 2; ++variant) {
    // variant 0: lane l supplies elements [l*4, l*4+4) (fully linear).  variant 1: row-major [k][128] tile: lane i of group g -> row g*8+(i>>2), col (i&3)*4
    for (int l = 0; l < 64; ++l) h[l] = variant == 0 ? l * 4 : ((l >> 4) * 8 + ((l & 15) >> 2)) * 128 + (l & 3) * 4;
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, o);
    hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
    printf("variant %d\n", variant);
    for (int l = 0; l < 64; ++l) printf("lane %2d addr %4d -> %4d %4d %4d %4d\n", l, h[l], r[l * 4], r[l * 4 + 1], r[l * 4 + 2], r[l * 4 + 3]);
  }
  return 0;
}
