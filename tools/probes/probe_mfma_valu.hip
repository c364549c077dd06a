// Probe: do MFMA and VALU work overlap on ONE SIMD of gfx950, (a) between two waves, (b) inside one wave's stream?
// Kernels (one workgroup per CU, 256 workgroups; times are per launch):
//   mode 0: 4 waves/WG (1 per SIMD), MFMA-only loop (4 independent accumulators)
//   mode 1: 4 waves/WG, VALU-only loop (independent v_fma_f32 chains) sized to ~the same duration
//   mode 2: 8 waves/WG (2 per SIMD): waves 0-3 MFMA loop, waves 4-7 VALU loop        -> max(0,1) if the pipes overlap
//   mode 3: 8 waves/WG: all MFMA                                                       -> 2 x mode 0 (shared matrix pipe)
//   mode 4: 8 waves/WG: all VALU                                                       -> 2 x mode 1
//   mode 5: 4 waves/WG: ONE stream, each MFMA followed by VPM plain VALU ops          -> mode 0 if VALU hides under MFMA
//   mode 6: like 2 but the VALU waves run v_exp_f32 (transcendental) instead of fma
//   mode 7: like 5 with v_exp_f32
// build: hipcc --offload-arch=gfx950 -O3 -o build/probes/probe_mfma_valu tools/probes/probe_mfma_valu.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#ifndef VPM
#define VPM 8
#endif

__device__ __forceinline__ void mfma_loop(int iters, float* out, int lane) {
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    bf16x8 x, y;
    for (int i = 0; i < 8; ++i) { x[i] = (__bf16)(0.001f * (lane + i)); y[i] = (__bf16)(0.002f * (lane - i)); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int a = 0; a < 4; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc[a], 0, 0, 0);
    }
    float s = 0.f;
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[0] = s;
}

template <bool EXP>
__device__ __forceinline__ void valu_loop(int iters, float* out, int lane) {
    float v[16];
    for (int r = 0; r < 16; ++r) v[r] = 0.001f * (lane + r);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (EXP) v[r] = __builtin_amdgcn_exp2f(v[r]) - 1.0f;
            else v[r] = __builtin_fmaf(v[r], 1.0001f, 0.0001f);
        }
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += v[r];
    out[0] = s;
}

template <bool EXP>
__device__ __forceinline__ void mixed_loop(int iters, float* out, int lane) {
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    bf16x8 x, y;
    for (int i = 0; i < 8; ++i) { x[i] = (__bf16)(0.001f * (lane + i)); y[i] = (__bf16)(0.002f * (lane - i)); }
    float v[VPM];
    for (int r = 0; r < VPM; ++r) v[r] = 0.001f * (lane + r);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc[a], 0, 0, 0);
#pragma unroll
            for (int r = 0; r < VPM; ++r) {
                if (EXP) v[r] = __builtin_amdgcn_exp2f(v[r]) - 1.0f;
                else v[r] = __builtin_fmaf(v[r], 1.0001f, 0.0001f);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    for (int r = 0; r < VPM; ++r) s += v[r];
    out[0] = s;
}

__global__ void __launch_bounds__(512) probe(int mode, int mi, int vi, float* out) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float* o = out + (size_t)blockIdx.x * 512 + threadIdx.x;
    switch (mode) {
        case 0: case 3: mfma_loop(mi, o, lane); break;
        case 1: case 4: valu_loop<false>(vi, o, lane); break;
        case 2: if (wv < 4) mfma_loop(mi, o, lane); else valu_loop<false>(vi, o, lane); break;
        case 6: if (wv < 4) mfma_loop(mi, o, lane); else valu_loop<true>(vi, o, lane); break;
        case 5: mixed_loop<false>(mi, o, lane); break;
        case 7: mixed_loop<true>(mi, o, lane); break;
        case 8: valu_loop<true>(vi, o, lane); break;
    }
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 512 * sizeof(float));
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const int mi = 2000;                 // 8000 MFMAs per wave = 256 000 matrix-pipe cycles
    auto run = [&](int mode, int threads, int vi, const char* what) {
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(a);
            hipLaunchKernelGGL(probe, dim3(256), dim3(threads), 0, 0, mode, mi, vi, out);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (rep && ms < best) best = ms;
        }
        printf("mode %d  %-62s %8.3f ms\n", mode, what, best);
        return best;
    };
    printf("VPM (VALU per MFMA in the mixed stream) = %d\n", VPM);
    run(0, 256, 0, "1 wave/SIMD, 8000 MFMA 32x32x16");
    const int vi = 8000;                 // 128 000 dependent-free fma per wave
    run(1, 256, vi, "1 wave/SIMD, 128000 v_fma_f32 (16 chains)");
    run(8, 256, vi / 4, "1 wave/SIMD, 32000 (v_exp_f32 + v_add) (16 chains)");
    run(2, 512, vi, "2 waves/SIMD: one MFMA loop + one v_fma loop");
    run(6, 512, vi / 4, "2 waves/SIMD: one MFMA loop + one v_exp loop");
    run(3, 512, 0, "2 waves/SIMD: both MFMA");
    run(4, 512, vi, "2 waves/SIMD: both v_fma");
    run(5, 256, 0, "1 wave/SIMD: ONE stream, each MFMA followed by VPM v_fma");
    run(7, 256, 0, "1 wave/SIMD: ONE stream, each MFMA followed by VPM (v_exp + v_add)");
    return 0;
}
