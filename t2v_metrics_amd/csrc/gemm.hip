// bf16 MFMA GEMM for gfx950 (MI355X):  C = epilogue(A[M,K] . W[N,K]^T), fp32 accumulate.
//
// This is kernel K1 of SURVEY.md §8(a-bis): every nn.Linear on the CLIP-FlanT5 path
// (HF models/clip/modeling_clip.py:303-350, HF models/t5/modeling_t5.py:97-127,206-209,1047).
//
// Structure (cdna_hip_programming.md §5, "glds vs register staging" table, 256² row):
//   * 256x256 output tile per 512-thread workgroup (8 waves as 2(M) x 4(N); 128x64 per wave),
//     BK = 64, v_mfma_f32_32x32x16_bf16, 128 fp32 accumulators per lane;
//   * both operands are K-contiguous, staged by direct-to-LDS DMA (global_load_lds_dwordx4) into a
//     double-buffered 2 x (32 KiB A + 32 KiB W) LDS image, one barrier per K-tile;
//   * LDS rows are 128 B; the 16-B chunk index is XOR-swizzled with (row>>1)&7 so that every
//     ds_read_b128 lane group touches 16 distinct 16-B slots of the 256-B bank row (conflict-free);
//     with DMA staging the swizzle is applied to the per-lane SOURCE address (LDS image stays
//     lane-linear) and again on the read (rule 21 of the guide);
//   * operands are fed to the MFMA swapped (a = W fragment, b = A fragment) so that a lane owns one
//     activation row and 4 consecutive output columns per register quad: 8-byte (bf16) / 16-byte
//     (fp32) epilogue stores, bias/activation/residual/gating fused;
//   * workgroup -> tile map is XCD-aware (block b runs on XCD b%8: each XCD gets a contiguous run
//     of tiles, walked in groups of 8 M-tiles x all N-tiles so neighbours share panels in that L2).
// M and N edges are handled by clamping source rows and predicating stores; K must be a multiple of 64.
#include "vqs_kernels.h"

namespace vqs {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

static constexpr int BM = 256, BN = 256, BK = 64;
static constexpr int STAGE_BYTES = 65536;   // A 32 KiB + W 32 KiB
static constexpr int W_OFF = 32768;

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {   // round-to-nearest-even
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}
__device__ __forceinline__ uint32_t pack2(float a, float b) { return (uint32_t)f2bf(a) | ((uint32_t)f2bf(b) << 16); }

__device__ __forceinline__ float act_quick_gelu(float x) { return x / (1.0f + __expf(-1.702f * x)); }
__device__ __forceinline__ float act_gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float act_gelu_new(float x) {
    const float k = 0.7978845608028654f;   // sqrt(2/pi)
    return 0.5f * x * (1.0f + tanhf(k * (x + 0.044715f * x * x * x)));
}

// One 1-KiB direct-to-LDS piece: lane l's 16 bytes from `src` land at LDS byte address dst + 16*l.
// M0 carries the wave-uniform LDS base; it is compiler-reserved, so it is saved/restored inside the
// same statement (cdna_hip_programming.md §5.7).
__device__ __forceinline__ void glds16(const void* src, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(src), "s"(lds_dst)
        : "memory");
}

template <int EPI, bool GLDS>
__global__ void __launch_bounds__(512) gemm_bf16_kernel(const GemmParams p) {
    __shared__ __attribute__((aligned(16))) char lds[2 * STAGE_BYTES];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = w >> 2, wc = w & 3;

    // ---- XCD-aware, grouped tile map (bijective for any tile count)
    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
    const int nwg = tiles_m * tiles_n;
    int t_lin;
    {
        const int pid = blockIdx.x;
        const int xcd = pid & 7, local = pid >> 3;
        const int q = nwg >> 3, r = nwg & 7;
        t_lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
    }
    const int GM = 8;
    const int width = GM * tiles_n;
    const int group = t_lin / width;
    const int first_m = group * GM;
    const int gsz = min(tiles_m - first_m, GM);
    const int tm = first_m + (t_lin % width) % gsz;
    const int tn = (t_lin % width) / gsz;
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- staging addresses: wave w, instruction i covers LDS rows (i*8+w)*8 .. +7 (1 KiB, lane-linear)
    const int sw = ((w & 1) << 2) + (lane >> 4);   // = (row>>1)&7 of the row this lane stages
    const int gchunk = (lane & 7) ^ sw;
    const bf16_t* pa[4];
    const bf16_t* pb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (i * 8 + w) * 8 + (lane >> 3);
        const int ra = min(m0 + row, p.M - 1);
        const int rb = min(n0 + row, p.N - 1);
        pa[i] = p.A + (size_t)ra * p.lda + gchunk * 8;
        pb[i] = p.W + (size_t)rb * p.ldw + gchunk * 8;
    }

    // ---- fragment read offsets (bytes within a stage)
    const int swr = (lane >> 1) & 7;              // (row>>1)&7 for row = 32-aligned base + (lane&31)
    int koff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) koff[ks] = (((ks * 2 + (lane >> 5)) ^ swr) << 4);
    const int a_row = (wr * 128 + (lane & 31)) * 128;
    const int b_row = W_OFF + (wc * 64 + (lane & 31)) * 128;

    f32x16 acc[4][2];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.0f;

    const int nt = p.K / BK;
    const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)LDS_PTR(lds));

    auto stage = [&](int s, int t) {
        char* base = lds + s * STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bf16_t* ga = pa[i] + (size_t)t * BK;
            const bf16_t* gb = pb[i] + (size_t)t * BK;
            if constexpr (GLDS) {
                // LDS-DMA issued from inline asm so that hipcc does not drain it (vmcnt(0)) in front of
                // the ds_reads of the *other* stage; completion is waited for explicitly before the barrier.
                const uint32_t da = lds_base + s * STAGE_BYTES + (i * 8 + w) * 1024;
                glds16(ga, da);
                glds16(gb, da + W_OFF);
            } else {
                const uint4 va = *reinterpret_cast<const uint4*>(ga);
                const uint4 vb = *reinterpret_cast<const uint4*>(gb);
                *reinterpret_cast<uint4*>(base + (i * 8 + w) * 1024 + lane * 16) = va;
                *reinterpret_cast<uint4*>(base + W_OFF + (i * 8 + w) * 1024 + lane * 16) = vb;
            }
        }
    };

    stage(0, 0);
    for (int t = 0; t < nt; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                       // tile t landed everywhere; buffer (t+1)&1 no longer read
        if (t + 1 < nt) stage((t + 1) & 1, t + 1);
        const char* sb = lds + (t & 1) * STAGE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            uint4 af[4], wf[2];
#pragma unroll
            for (int m = 0; m < 4; ++m) af[m] = *reinterpret_cast<const uint4*>(sb + a_row + m * 4096 + koff[ks]);
#pragma unroll
            for (int n = 0; n < 2; ++n) wf[n] = *reinterpret_cast<const uint4*>(sb + b_row + n * 4096 + koff[ks]);
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                        __builtin_bit_cast(bf16x8, wf[n]), __builtin_bit_cast(bf16x8, af[m]), acc[m][n], 0, 0, 0);
        }
    }

    // ---- epilogue.  acc[m][n][r]: row = m0 + wr*128 + m*32 + (lane&31)
    //                               col = n0 + wc*64 + n*32 + 8*(r>>2) + 4*(lane>>5) + (r&3)
    const int hhalf = lane >> 5;
    const int row_base = m0 + wr * 128 + (lane & 31);
    const int col_base = n0 + wc * 64 + 4 * hhalf;

    if constexpr (EPI == EPI_GATED) {
        // fragment n=0 holds wi_0 (gate) and n=1 holds wi_1 (linear) for the same 32 output columns
        const int oc_base = ((n0 + wc * 64) >> 1) + 4 * hhalf;
        const int NO = p.N >> 1;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int row = row_base + m * 32;
            if (row >= p.M) continue;
            bf16_t* crow = reinterpret_cast<bf16_t*>(p.C) + (size_t)row * p.ldc;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int oc = oc_base + 8 * g;
                if (oc >= NO) continue;
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = act_gelu_new(acc[m][0][4 * g + e]) * acc[m][1][4 * g + e];
                uint2 v;
                v.x = pack2(o[0], o[1]);
                v.y = pack2(o[2], o[3]);
                *reinterpret_cast<uint2*>(crow + oc) = v;
            }
        }
        return;
    } else {
        // bias for this lane's 2 x 4 column quads
        float bia[2][4][4];
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = col_base + n * 32 + 8 * g;
                if (p.bias != nullptr && c < p.N) {
                    const uint2 bv = *reinterpret_cast<const uint2*>(p.bias + c);
                    bia[n][g][0] = bf2f((bf16_t)(bv.x & 0xffff));
                    bia[n][g][1] = bf2f((bf16_t)(bv.x >> 16));
                    bia[n][g][2] = bf2f((bf16_t)(bv.y & 0xffff));
                    bia[n][g][3] = bf2f((bf16_t)(bv.y >> 16));
                } else {
                    bia[n][g][0] = bia[n][g][1] = bia[n][g][2] = bia[n][g][3] = 0.0f;
                }
            }
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int row = row_base + m * 32;
            if (row >= p.M) continue;
            int hb = 0, hs = 0;
            if constexpr (EPI == EPI_HEADS) {
                hb = row / p.S;
                hs = row - hb * p.S;
            }
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c = col_base + n * 32 + 8 * g;
                    if (c >= p.N) continue;
                    float o[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = acc[m][n][4 * g + e] + bia[n][g][e];
                    if constexpr (EPI == EPI_BF16_QGELU) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = act_quick_gelu(o[e]);
                    }
                    if constexpr (EPI == EPI_BF16_GELU) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = act_gelu_erf(o[e]);
                    }
                    if constexpr (EPI == EPI_F32 || EPI == EPI_F32_RESID) {
                        float* cp = reinterpret_cast<float*>(p.C) + (size_t)row * p.ldc + c;
                        if constexpr (EPI == EPI_F32_RESID) {
                            const float4 rv = *reinterpret_cast<const float4*>(p.resid + (size_t)row * p.ldc + c);
                            o[0] += rv.x; o[1] += rv.y; o[2] += rv.z; o[3] += rv.w;
                        }
                        *reinterpret_cast<float4*>(cp) = make_float4(o[0], o[1], o[2], o[3]);
                    } else {
                        uint2 v;
                        v.x = pack2(o[0], o[1]);
                        v.y = pack2(o[2], o[3]);
                        if constexpr (EPI == EPI_HEADS) {
                            const int which = c / p.inner;
                            const int ci = c - which * p.inner;
                            const int hh = ci >> 6, d = ci & 63;
                            bf16_t* dst = p.heads_out[which] + (((size_t)hb * p.H + hh) * p.S + hs) * 64 + d;
                            *reinterpret_cast<uint2*>(dst) = v;
                        } else {
                            *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.C) + (size_t)row * p.ldc + c) = v;
                        }
                    }
                }
        }
    }
}

template <int EPI>
static hipError_t launch_epi(const GemmParams& p, int variant, hipStream_t stream) {
    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
    dim3 grid(tiles_m * tiles_n), block(512);
    if (variant == 0)
        hipLaunchKernelGGL((gemm_bf16_kernel<EPI, true>), grid, block, 0, stream, p);
    else
        hipLaunchKernelGGL((gemm_bf16_kernel<EPI, false>), grid, block, 0, stream, p);
    return hipGetLastError();
}

hipError_t launch_gemm(const GemmParams& p, int epilogue, int variant, hipStream_t stream) {
    if (p.M <= 0 || p.N <= 0 || p.K <= 0 || (p.K % BK) != 0 || (p.N % 8) != 0) return hipErrorInvalidValue;
    if ((p.lda % 8) != 0 || (p.ldw % 8) != 0) return hipErrorInvalidValue;
    switch (epilogue) {
        case EPI_BF16: return launch_epi<EPI_BF16>(p, variant, stream);
        case EPI_BF16_QGELU: return launch_epi<EPI_BF16_QGELU>(p, variant, stream);
        case EPI_BF16_GELU: return launch_epi<EPI_BF16_GELU>(p, variant, stream);
        case EPI_F32: return launch_epi<EPI_F32>(p, variant, stream);
        case EPI_F32_RESID: return launch_epi<EPI_F32_RESID>(p, variant, stream);
        case EPI_GATED: return launch_epi<EPI_GATED>(p, variant, stream);
        case EPI_HEADS: return launch_epi<EPI_HEADS>(p, variant, stream);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace vqs
