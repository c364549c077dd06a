// Probe: what does staging 64 KiB per K-tile cost the ISSUING wave on gfx950, as LDS-DMA (buffer_load_dwordx4 ... lds) and as
// global -> VGPR -> ds_write_b128, alone and inside an MFMA stream (one wave per SIMD, the wide GEMM form's situation)?
// One 256- or 512-thread workgroup per CU (128 KiB of LDS), 256 workgroups; the source is a 64 KiB region per workgroup that
// stays in L2 (32 workgroups x 64 KiB per XCD), so the numbers are the CU-side path, not HBM.
//   mode 0: 4 waves, LDS-DMA only: 16 pieces (1 KiB each) per wave per iteration, s_waitcnt vmcnt(0) per iteration
//   mode 1: 8 waves, LDS-DMA only: 8 pieces per wave per iteration (same 64 KiB per workgroup iteration)
//   mode 2: 4 waves, buffer_load_dwordx4 into VGPRs only (16 per iteration)
//   mode 3: 4 waves, mode 2 + ds_write_b128 of what was loaded
//   mode 4: 4 waves, 64 MFMAs per iteration, nothing else                      (2 048 matrix-pipe cycles per iteration)
//   mode 5: 4 waves, 64 MFMAs + 16 LDS-DMA pieces, one after every 4th MFMA
//   mode 6: 4 waves, 64 MFMAs + 16 (VGPR load, ds_write of the previous one), one after every 4th MFMA
//   mode 7: 4 waves, 64 MFMAs + 16 LDS-DMA pieces, vmcnt never waited inside the loop (issue cost only)
// Output: ms per launch, cycles per iteration at the measured rate of mode 4 (= 2 048 cycles), staged bytes per cycle and CU.
// build: hipcc --offload-arch=gfx950 -O3 -o build/probes/probe_staging tools/probes/probe_staging.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int v4i_t __attribute__((ext_vector_type(4)));
typedef uint32_t u4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ v4i_t make_rsrc(const void* base) {
    const uint64_t b = (uint64_t)base;
    v4i_t r;
    r.x = __builtin_amdgcn_readfirstlane((uint32_t)b);
    r.y = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32) & 0xffffu);
    r.z = (int)0xffffffffu;
    r.w = 0x00020000;
    return r;
}
__device__ __forceinline__ void dma16(v4i_t rsrc, uint32_t voff, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_dst));
}
__device__ __forceinline__ u4v vload16(v4i_t rsrc, uint32_t voff) {
    u4v r;
    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(r) : "v"(voff), "s"(rsrc));
    return r;
}
#define MFMA(ACC) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(ACC) : "v"(x), "v"(y))

template <int MODE>
__global__ void __launch_bounds__(MODE == 1 ? 512 : 256) probe(const char* src, float* out, int iters) {
    __shared__ __attribute__((aligned(16))) char lds[131072];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const v4i_t rs = make_rsrc(src + (size_t)blockIdx.x * 65536);
    const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)((__attribute__((address_space(3))) void*)lds));
    constexpr int NW = MODE == 1 ? 8 : 4, PW = 64 / NW;          // pieces per wave per iteration
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    bf16x8 x, y;
    for (int i = 0; i < 8; ++i) { x[i] = (__bf16)(0.001f * (lane + i)); y[i] = (__bf16)(0.002f * (lane - i)); }
    u4v held = {0, 0, 0, 0};
    uint32_t sink = 0;
    for (int it = 0; it < iters; ++it) {
        const uint32_t stage = (it & 1) * 65536;
        if (MODE == 0 || MODE == 1) {
#pragma unroll
            for (int i = 0; i < PW; ++i) dma16(rs, (uint32_t)((i * NW + w) * 1024 + lane * 16), lds_base + stage + (i * NW + w) * 1024);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if (MODE == 2 || MODE == 3) {
            u4v v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = vload16(rs, (uint32_t)((i * 4 + w) * 1024 + lane * 16));
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (MODE == 3) *reinterpret_cast<u4v*>(lds + stage + (i * 4 + w) * 1024 + lane * 16) = v[i];
                else sink += v[i].x;
            }
        } else {
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                MFMA(acc[0]); MFMA(acc[1]); MFMA(acc[2]); MFMA(acc[3]);
                if (MODE == 5 || MODE == 7) dma16(rs, (uint32_t)((g * 4 + w) * 1024 + lane * 16), lds_base + stage + (g * 4 + w) * 1024);
                if (MODE == 6) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the load issued one group (128 cycles) ago
                    *reinterpret_cast<u4v*>(lds + stage + (g * 4 + w) * 1024 + lane * 16) = held;
                    held = vload16(rs, (uint32_t)((g * 4 + w) * 1024 + lane * 16));
                }
            }
            if (MODE == 5) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    float s = (float)sink + (float)held.x + (float)lds[(tid * 16) & 131071];
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    if (s == 12345.678f) out[blockIdx.x] = s;
}

template <int MODE>
static float run(const char* src, float* out, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int threads = MODE == 1 ? 512 : 256;
    hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(threads), 0, 0, src, out, 16);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(threads), 0, 0, src, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best;
}

// ---- ring probe: does a deeper LDS-DMA prefetch ring hide the L2-miss latency that a 2-stage pipeline exposes?
// Slices of 32 KiB (BK = 32 of a 256x256 tile: 16 KiB of A shared by the 4 workgroups of an XCD's tile-window column, 16 KiB of W
// shared by the 8 of a window row -- the GEMM's sharing pattern, so ~80 % of the requests hit L2 and the rest come over the
// fabric), 8 LDS-DMA pieces + 32 MFMAs per wave and slice (1 024 matrix-pipe cycles).  DEPTH = slices in flight ahead of the one
// being consumed: before consuming slice i the wave waits until at most 8 (DEPTH - 1) of its pieces are outstanding.
template <int DEPTH>
__global__ void __launch_bounds__(256) ring_probe(const char* src, float* out, int iters, size_t stream_bytes) {
    __shared__ __attribute__((aligned(16))) char lds[4 * 32768];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xcd = blockIdx.x & 7, l = blockIdx.x >> 3;                   // 32 workgroups per XCD
    const v4i_t rsA = make_rsrc(src + (size_t)(xcd * 12 + (l & 7)) * stream_bytes);          // 8 A streams per XCD, 4 sharers each
    const v4i_t rsW = make_rsrc(src + (size_t)(xcd * 12 + 8 + (l >> 3)) * stream_bytes);     // 4 W streams per XCD, 8 sharers each
    const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)((__attribute__((address_space(3))) void*)lds));
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    bf16x8 x, y;
    for (int i = 0; i < 8; ++i) { x[i] = (__bf16)(0.001f * (lane + i)); y[i] = (__bf16)(0.002f * (lane - i)); }
    auto issue = [&](int slice) {                                          // this wave's 8 pieces of a slice: 4 of A, 4 of W
        const uint32_t so = (uint32_t)slice * 16384u, dst = lds_base + (slice & 3) * 32768;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            dma16(rsA, so + (uint32_t)((i * 4 + w) * 1024 + lane * 16), dst + (i * 4 + w) * 1024);
            dma16(rsW, so + (uint32_t)((i * 4 + w) * 1024 + lane * 16), dst + 16384 + (i * 4 + w) * 1024);
        }
    };
    for (int d = 0; d < DEPTH; ++d) issue(d);
    for (int it = 0; it < iters; ++it) {
        if (DEPTH == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (DEPTH == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (DEPTH == 3) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            MFMA(acc[0]); MFMA(acc[1]); MFMA(acc[2]); MFMA(acc[3]);
            if (g == 0 && it + DEPTH < iters) {                            // refill the slot consumed one iteration ago (DEPTH < 4) / just now
                const uint32_t so = (uint32_t)(it + DEPTH) * 16384u, dst = lds_base + ((it + DEPTH) & 3) * 32768;
                (void)so; (void)dst;
            }
            if (it + DEPTH < iters) {
                const uint32_t so = (uint32_t)(it + DEPTH) * 16384u, dst = lds_base + ((it + DEPTH) & 3) * 32768;
                const int i = g >> 1;
                if ((g & 1) == 0) dma16(rsA, so + (uint32_t)((i * 4 + w) * 1024 + lane * 16), dst + (i * 4 + w) * 1024);
                else dma16(rsW, so + (uint32_t)((i * 4 + w) * 1024 + lane * 16), dst + 16384 + (i * 4 + w) * 1024);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    float s = (float)lds[(tid * 16) & 131071];
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    if (s == 12345.678f) out[blockIdx.x] = s;
}

template <int DEPTH>
static float run_ring(const char* src, float* out, int iters, size_t stream_bytes) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(ring_probe<DEPTH>, dim3(256), dim3(256), 0, 0, src, out, 16, stream_bytes);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(ring_probe<DEPTH>, dim3(256), dim3(256), 0, 0, src, out, iters, stream_bytes);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best;
}

int main() {
    char* src; float* out;
    hipMalloc(&src, (size_t)256 * 65536); hipMemset(src, 1, (size_t)256 * 65536);
    hipMalloc(&out, 4096);
    const int iters = 2000;
    float ms[8];
    ms[4] = run<4>(src, out, iters);
    ms[0] = run<0>(src, out, iters); ms[1] = run<1>(src, out, iters); ms[2] = run<2>(src, out, iters); ms[3] = run<3>(src, out, iters);
    ms[5] = run<5>(src, out, iters); ms[6] = run<6>(src, out, iters); ms[7] = run<7>(src, out, iters);
    const double cyc_per_ms = 2048.0 * iters / ms[4];            // mode 4 = 64 MFMAs x 32 cycles per iteration
    const char* what[8] = {"LDS-DMA only, 4 waves", "LDS-DMA only, 8 waves", "VGPR loads only", "VGPR loads + ds_write", "64 MFMA only",
                           "64 MFMA + 16 LDS-DMA (wait per iteration)", "64 MFMA + 16 (VGPR load, ds_write)", "64 MFMA + 16 LDS-DMA (no wait)"};
    for (int m = 0; m < 8; ++m) {
        const double cyc = ms[m] * cyc_per_ms / iters;
        printf("{\"mode\": %d, \"what\": \"%s\", \"ms\": %.4f, \"cycles_per_iteration\": %.0f, \"staged_bytes_per_cycle_per_cu\": %.1f}\n", m, what[m], ms[m],
               cyc, m == 4 ? 0.0 : 65536.0 / cyc);
    }
    printf("{\"clock_ghz_under_mfma_load\": %.3f}\n", cyc_per_ms / 1e6);
    {
        const int ri = 4000;                                   // slices per workgroup; a stream advances 16 KiB per slice
        const size_t stream_bytes = (size_t)ri * 16384 + 65536;
        char* big;
        if (hipMalloc(&big, 96 * stream_bytes) != hipSuccess) { printf("{\"ring\": \"alloc failed\"}\n"); return 0; }
        hipMemset(big, 1, 96 * stream_bytes);
        const float r1 = run_ring<1>(big, out, ri, stream_bytes), r2 = run_ring<2>(big, out, ri, stream_bytes),
                    r3 = run_ring<3>(big, out, ri, stream_bytes), r4 = run_ring<4>(big, out, ri, stream_bytes);
        const float rr[4] = {r1, r2, r3, r4};
        for (int d = 0; d < 4; ++d)
            printf("{\"ring_depth\": %d, \"ms\": %.4f, \"cycles_per_slice\": %.0f, \"mfma_cycles_per_slice\": 1024, \"fabric_side_tb_per_s\": %.2f}\n", d + 1, rr[d],
                   rr[d] * cyc_per_ms / ri, 96.0 * ri * 16384.0 / (rr[d] * 1e-3) / 1e12);
    }
    return 0;
}
