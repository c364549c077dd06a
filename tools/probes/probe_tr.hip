// Lab probe: semantics of ds_read_b64_tr_b16 on gfx950.  LDS holds element index i at lds[i] (u16); every lane supplies
// address base + 8*lane bytes; prints what each lane receives.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void probe(uint16_t* out) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    const uint32_t addr = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) void*)lds) + 8 * threadIdx.x;
    uint2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    out[threadIdx.x * 4 + 0] = v.x & 0xffff;
    out[threadIdx.x * 4 + 1] = v.x >> 16;
    out[threadIdx.x * 4 + 2] = v.y & 0xffff;
    out[threadIdx.x * 4 + 3] = v.y >> 16;
}
int main() {
    uint16_t* d; hipMalloc(&d, 64 * 4 * 2);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    uint16_t h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
    return 0;
}
