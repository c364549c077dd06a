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
#include <cstdlib>
#include <type_traits>

// Kernels of this file: variant 0 one tile per workgroup, 2 / 5 the ping-pong schedule, 3 / 11 persistent with the schedule
// chosen by shape (3 hands the big bf16-result launches to the quad form of gemm_quad.inc).  The A/B forms of rounds 1-3
// (register-staged, wave-specialised, wide, ring) and the ablation builds are in the git history only.

namespace vqs {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

static constexpr int BM = GEMM_BM, BN = GEMM_BN, BK = 64;
static constexpr int STAGE_BYTES = 65536;   // A 32 KiB + W 32 KiB
static constexpr int W_OFF = 32768;
static constexpr int PERSISTENT_WGS = 256;   // one 128-KiB-LDS workgroup per CU
#ifndef VQS_QUAD_WGS                         // lab (make variant VFLAGS=-DVQS_QUAD_WGS=1073741824): one workgroup per TILE, the vendor kernel's launch shape -- the
#define VQS_QUAD_WGS PERSISTENT_WGS          // hardware dispatcher starts a CU's next tile when the previous one exits, so the chip's epilogues drift out of phase
#endif

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ bf16_t f2bf(float f) {   // round-to-nearest-even (v_cvt_pk_bf16_f32 on gfx950)
    return __builtin_bit_cast(bf16_t, (__bf16)f);
}
__device__ __forceinline__ uint32_t pack2(float a, float b) {
    bf16x2_t v;
    v[0] = (__bf16)a;
    v[1] = (__bf16)b;
    return __builtin_bit_cast(uint32_t, v);
}

typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
__device__ __forceinline__ uint32_t pack2h(float a, float b) {   // v_cvt_pk_f16_f32 on gfx950 (RNE, overflow -> inf)
    f16x2_t v;
    v[0] = (_Float16)a;
    v[1] = (_Float16)b;
    return __builtin_bit_cast(uint32_t, v);
}

// x * sigmoid(1.702 x)                                               HF activations.py:117-123
// sigmoid's reciprocal is the hardware v_rcp_f32 (1 ulp): __frcp_rn is a correctly rounded division -- two v_div_scale, v_rcp,
// five FMAs, v_div_fmas, v_div_fixup per element (ISA), 9 extra VALU instructions per output element in an epilogue that no
// MFMA work overlaps -- for a result that is rounded to bf16 next.  The argument lies in [1, +inf]; rcp(+inf) = 0.
__device__ __forceinline__ float fast_rcp(float y) { return __builtin_amdgcn_rcpf(y); }
__device__ __forceinline__ float act_quick_gelu(float x) { return x * fast_rcp(1.0f + __expf(-1.702f * x)); }
__device__ __forceinline__ float act_gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
// 0.5 x (1 + tanh(u)), u = sqrt(2/pi)(x + 0.044715 x^3)  ==  x * sigmoid(2u)      HF activations.py:59-66
__device__ __forceinline__ float act_gelu_new(float x) {
    const float u2 = 1.5957691216057308f * (x + 0.044715f * x * x * x);   // 2u
    return x * fast_rcp(1.0f + __expf(-u2));
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
        : "v"(src), "s"(lds_dst));
}

template <int EPI, int VAR>
__global__ void __launch_bounds__(512) gemm_bf16_kernel(const GemmParams p) {
    __shared__ __attribute__((aligned(16))) char lds[2 * STAGE_BYTES];

    constexpr bool GLDS = (VAR != 1);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = w >> 2, wc = w & 3;

    // ---- XCD-aware, grouped tile map (bijective for any tile count)
    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
    int m0, n0, bz_unused;
    tile_of_slot(blockIdx.x, tiles_m * tiles_n, tiles_m, tiles_n, p.tile_gm, p.tile_ns, m0, n0, bz_unused);

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

    if constexpr (VAR == 2) {
        // ---- ping-pong schedule.  The two waves that share a SIMD (w and w+4, i.e. wr = 0 / 1) run one barrier
        // apart: while one group is in an MFMA segment (8 MFMAs = one k-step of 16) the other is in a load segment
        // (its 6 ds_read_b128 of the next k-step, plus the LDS-DMA issue for the next K-tile), then they swap.
        // Slot k (between barriers k and k+1): group 0 runs L(k/2) | M((k-1)/2), group 1 the other kind.
        //   - tile t+1 is staged into the stage tile t-1 occupied; its DMA is issued in phases 1 and 2 of tile t,
        //     i.e. at least one barrier after every wave has waited (lgkmcnt) for its last read of tile t-1;
        //   - every wave drains its own DMA (vmcnt(0)) in slot 8t+7, one barrier before the first read of tile
        //     t+1 (group 0: end of its last MFMA segment; group 1: end of its last load segment).
        auto stage_part = [&](int s, int t, int i0) {
#pragma unroll
            for (int i = i0; i < i0 + 2; ++i) {
                const uint32_t da = lds_base + s * STAGE_BYTES + (i * 8 + w) * 1024;
                glds16(pa[i] + (size_t)t * BK, da);
                glds16(pb[i] + (size_t)t * BK, da + W_OFF);
            }
        };
        const bool g1 = (wr == 1);
        stage_part(0, 0, 0);
        stage_part(0, 0, 2);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (g1) __builtin_amdgcn_s_barrier();
        for (int t = 0; t < nt; ++t) {
            const char* sb = lds + (t & 1) * STAGE_BYTES;
            const bool more = (t + 1 < nt);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                // ---- load segment
                uint4 af[4], wf[2];
#pragma unroll
                for (int n = 0; n < 2; ++n) wf[n] = *reinterpret_cast<const uint4*>(sb + b_row + n * 4096 + koff[ks]);
#pragma unroll
                for (int m = 0; m < 4; ++m) af[m] = *reinterpret_cast<const uint4*>(sb + a_row + m * 4096 + koff[ks]);
                if (more && ks == 1) stage_part((t + 1) & 1, t + 1, 0);
                if (more && ks == 2) stage_part((t + 1) & 1, t + 1, 2);
                if (ks == 3 && g1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                // ---- MFMA segment
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                            __builtin_bit_cast(bf16x8, wf[n]), __builtin_bit_cast(bf16x8, af[m]), acc[m][n], 0, 0, 0);
                __builtin_amdgcn_s_setprio(0);
                if (ks == 3 && !g1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (!g1) __builtin_amdgcn_s_barrier();
    } else {
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
        // EPI_HEADS: a wave's 64 columns are exactly one head of one of the q/k/v tensors (inner % 64 == 0)
        bf16_t* head_base = nullptr;
        if constexpr (EPI == EPI_HEADS) {
            const int cw = min(n0 + wc * 64, p.N - 64);
            const int which = cw / p.inner;
            bf16_t* hp = which == 0 ? p.heads_out[0] : (which == 1 ? p.heads_out[1] : p.heads_out[2]);
            head_base = hp + (size_t)((cw - which * p.inner) >> 6) * p.S * 64;
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
                            bf16_t* dst = head_base + ((size_t)hb * p.H * p.S + hs) * 64 + (c - (n0 + wc * 64));
                            *reinterpret_cast<uint2*>(dst) = v;
                        } else {
                            *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.C) + (size_t)row * p.ldc + c) = v;
                        }
                    }
                }
        }
    }
}


// =====================================================================================================
// Persistent variant (VAR 3): one workgroup per CU walks the tile list; the 2-stage K-tile pipeline runs
// straight across tile boundaries (the first K-tile of the next tile is in flight during the last K-tile of
// the current one) and the epilogue stores drain underneath the next tile's first K-tile: the main loop uses
// raw s_barrier (no fence, so hipcc adds no vmcnt(0) for the stores) and, because gfx950 retires VMEM
// operations in order on one counter, `s_waitcnt vmcnt(<stores per wave>)` after a full tile's epilogue means
// "everything older than those stores (= the prefetched K-tile) has landed" without waiting for the stores.
// The fp32 residual read-modify-write is software-pipelined (loads of row block m+1 are issued before the
// stores of block m) so that no load ever has to wait behind a just-issued store.
// Measured motivation (tools/gemm_lab.sh ablations, 155648x10240x2048): stores cost 19 %, staging+prologue 26 %.
// =====================================================================================================
// Row stores of the staged epilogue.  nt (GemmParams::nt_store): the result leaves with the non-temporal hint -- a multi-GB
// output that the next kernel streams once should not push the A / W panels out of the Infinity Cache (the tile order is chosen
// so that they stay there).  Same bytes to the same addresses: results do not depend on it.
typedef uint32_t g_u4v __attribute__((ext_vector_type(4)));
typedef float g_f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st_row16(void* p, uint4 v, bool nt) {
    if (nt) {
        g_u4v u = {v.x, v.y, v.z, v.w};
        __builtin_nontemporal_store(u, reinterpret_cast<g_u4v*>(p));
    } else {
        *reinterpret_cast<uint4*>(p) = v;
    }
}
__device__ __forceinline__ void st_row16f(void* p, float4 v, bool nt) {
    if (nt) {
        g_f4v u = {v.x, v.y, v.z, v.w};
        __builtin_nontemporal_store(u, reinterpret_cast<g_f4v*>(p));
    } else {
        *reinterpret_cast<float4*>(p) = v;
    }
}
__device__ __forceinline__ void st8(void* p, uint2 v) { *reinterpret_cast<uint2*>(p) = v; }
__device__ __forceinline__ void st16(void* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }



// ----------------------------------------------------------------------------------------------------
// Staged epilogue: accumulators -> (bias / activation / gate) -> wave-private 8-KiB LDS region -> row-contiguous
// 16-B-per-lane global stores.  `reg` is this wave's region inside an LDS stage nobody reads any more.
//   bf16 outputs : two passes of 64 rows x 128 B (one row = the wave's 64 columns);   gated: 64 rows x 64 B
//   fp32 outputs : four passes of 32 rows x 256 B
// LDS image: 16-B chunk index XOR-ed with the row (conflict-free b128 reads, <= 2-way writes).
// ----------------------------------------------------------------------------------------------------
template <int EPI, bool OUT_F16 = false>      // OUT_F16: the 16-bit result is IEEE fp16 (GemmParams::f16 == 1; the persistent kernel's FT)
__device__ __forceinline__ void staged_epilogue(const GemmParams& p, f32x16 (&acc)[4][2], char* reg, int m0, int n0, int bz,
                                                int wr, int wc, int lane, bool full, float* rowred) {
    const int hh = lane >> 5, lr = lane & 31;
    const bool nt = p.nt_store != 0;
    const int row_w = m0 + wr * 128;                 // first row of this wave's tile
    const int col_w = n0 + wc * 64;                  // first column
    if (p.rowss_in != nullptr) {
        // Consumer of a fused residual + RMSNorm producer: the A operand was x*lnw WITHOUT the per-row 1/rms factor
        // (a scalar per row commutes with the contraction); apply it to the accumulators.  rowss_parts == 0: rowss_in
        // already holds 1/rms per row (the engine's form, one load per row); otherwise it holds the producer's per-tile
        // partial sums of squares, added here in index order (deterministic).
        float rsv[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int row = min(row_w + m * 32 + lr, p.M - 1);
            if (p.rowss_parts == 0) {
                rsv[m] = p.rowss_in[row];
            } else {
                float ss = 0.0f;
                for (int q = 0; q < p.rowss_parts; ++q) ss += p.rowss_in[(size_t)q * p.M + row];
                rsv[m] = rsqrtf(ss * p.rs_invd + p.rs_eps);
            }
        }
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][n][r] *= rsv[m];
    }
    if constexpr (EPI == EPI_RESID_RMS) {
        // [Lab path, VQS_FUSED_NORM=1; measured +1 % end to end, see DESIGN.md §3: a read-modify-write epilogue pays one store-
        //  acknowledgement latency per batch of loads because VMEM returns in order, ~30 us per tile, and starting the
        //  workgroups in staggered phase groups did not change that.]
        // hres += acc (fp32 stream, written back); C = bf16(hres * lnw[col]) -- the next RMSNorm's operand without its
        // 1/rms factor; per-row sum of squares of this tile's 256 columns -> rowss_out[n0/256][row].
        // Four passes of 32 rows x 256 B through the wave's LDS region (the fp32 image below); after the transposing
        // LDS round trip a lane holds 4 consecutive columns of one row and 16 lanes share a row.
        const int c = lane & 15;
        const int col = col_w + c * 4;
        float w4[4] = {0.f, 0.f, 0.f, 0.f};
        if (col < p.N) {
            const uint2 wv = *reinterpret_cast<const uint2*>(p.lnw + col);
            w4[0] = bf2f((bf16_t)(wv.x & 0xffff)); w4[1] = bf2f((bf16_t)(wv.x >> 16));
            w4[2] = bf2f((bf16_t)(wv.y & 0xffff)); w4[3] = bf2f((bf16_t)(wv.y >> 16));
        }
        // The 8 loads of the stream for pass m+1 are issued before the stores of pass m (VMEM retires in order: a load
        // queued behind stores waits for their acknowledgements), and pass 0's before anything else; the main loop has
        // touched every line of the tile into L2 during the last K-tile (see the kernel), so these are L2 hits.
        // Half-passes of 4 x (4 rows x 256 B): the 4 stream loads of half-pass i+1 are issued before the stores of
        // half-pass i (VMEM retires in order: a load queued behind stores waits for their acknowledgements), the first
        // ones before anything else; the main loop has touched every line of the tile into L2 during the last K-tile.
        float4 hv[4];
        const bool col_ok = col < p.N;
        auto load_h = [&](int i) {                 // i = 2*m + half
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = row_w + (i >> 1) * 32 + ((i & 1) * 4 + q) * 4 + (lane >> 4);
                hv[q] = (full || (row < p.M && col_ok)) ? *reinterpret_cast<const float4*>(p.hres + (size_t)row * p.ldh + col)
                                                         : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        };
        load_h(0);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int ch = n * 8 + 2 * g + hh;
                    const float4 v = make_float4(acc[m][n][4 * g + 0], acc[m][n][4 * g + 1], acc[m][n][4 * g + 2], acc[m][n][4 * g + 3]);
                    *reinterpret_cast<float4*>(reg + lr * 256 + ((ch ^ (lr & 15)) << 4)) = v;
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                float4 v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int r = (half * 4 + q) * 4 + (lane >> 4);
                    v[q] = *reinterpret_cast<const float4*>(reg + r * 256 + ((c ^ (r & 15)) << 4));
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    v[q].x += hv[q].x; v[q].y += hv[q].y; v[q].z += hv[q].z; v[q].w += hv[q].w;
                }
                if (2 * m + half + 1 < 8) load_h(2 * m + half + 1);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int r = (half * 4 + q) * 4 + (lane >> 4);
                    const int row = row_w + m * 32 + r;
                    float ss = 0.0f;
                    if (full || (row < p.M && col_ok)) {
                        *reinterpret_cast<float4*>(p.hres + (size_t)row * p.ldh + col) = v[q];
                        uint2 o;
                        o.x = pack2(v[q].x * w4[0], v[q].y * w4[1]);
                        o.y = pack2(v[q].z * w4[2], v[q].w * w4[3]);
                        *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.C) + (size_t)row * p.ldc + col) = o;
                        ss = (v[q].x * v[q].x + v[q].y * v[q].y) + (v[q].z * v[q].z + v[q].w * v[q].w);
                    }
                    ss += __shfl_xor(ss, 1);
                    ss += __shfl_xor(ss, 2);
                    ss += __shfl_xor(ss, 4);
                    ss += __shfl_xor(ss, 8);
                    if (c == 0) rowred[wc * 256 + wr * 128 + m * 32 + r] = ss;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        __syncthreads();                                       // all four column quarters of every row are in rowred
        {
            const int t = threadIdx.x;
            if (t < 256 && m0 + t < p.M) {
                const float tot = ((rowred[t] + rowred[256 + t]) + rowred[512 + t]) + rowred[768 + t];
                p.rowss_out[(size_t)(n0 >> 8) * p.M + m0 + t] = tot;
            }
        }
        return;
    }
    if constexpr (EPI == EPI_GATED) {
        const int NO = p.N >> 1;
        const int ocol_w = col_w >> 1;               // 32 output columns per wave
        if (p.bias != nullptr) {                     // packed (interleaved) order, like the weight rows
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c = col_w + n * 32 + 8 * g + 4 * hh;
                    if (c < p.N) {
                        const uint2 bv = *reinterpret_cast<const uint2*>(p.bias + c);
                        const float b0 = bf2f((bf16_t)(bv.x & 0xffff)), b1 = bf2f((bf16_t)(bv.x >> 16));
                        const float b2 = bf2f((bf16_t)(bv.y & 0xffff)), b3 = bf2f((bf16_t)(bv.y >> 16));
#pragma unroll
                        for (int m = 0; m < 4; ++m) {
                            acc[m][n][4 * g + 0] += b0; acc[m][n][4 * g + 1] += b1;
                            acc[m][n][4 * g + 2] += b2; acc[m][n][4 * g + 3] += b3;
                        }
                    }
                }
        }
        const bool silu = p.gate_act == 1;           // 0: gelu_new (T5 gated-gelu), 1: SiLU (Qwen SwiGLU)
#pragma unroll
        for (int hm = 0; hm < 2; ++hm) {
#pragma unroll
            for (int m2 = 0; m2 < 2; ++m2) {
                const int m = 2 * hm + m2;
                const int r = m2 * 32 + lr;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float o[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float gv = acc[m][0][4 * g + e];
                        const float a = silu ? gv * fast_rcp(1.0f + __expf(-gv)) : act_gelu_new(gv);
                        o[e] = a * acc[m][1][4 * g + e];
                    }
                    uint2 v;
                    v.x = pack2(o[0], o[1]);
                    v.y = pack2(o[2], o[3]);
                    *reinterpret_cast<uint2*>(reg + r * 64 + ((g ^ ((r >> 1) & 3)) << 4) + hh * 8) = v;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int q = 0; q < 4; ++q) {                     // 16 rows x 64 B per instruction
                const int r = q * 16 + (lane >> 2), c = lane & 3;
                const uint4 v = *reinterpret_cast<const uint4*>(reg + r * 64 + ((c ^ ((r >> 1) & 3)) << 4));
                const int row = row_w + hm * 64 + r, oc = ocol_w + c * 8;
                if (full || (row < p.M && oc < NO))
                    st_row16(reinterpret_cast<bf16_t*>(p.C) + (size_t)bz * p.sC + (size_t)row * p.ldc + oc, v, nt);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    } else {
        // bias of this lane's 2 x 4 column quads, added into the accumulators -- under a wave-uniform branch: no T5 linear
        // has a bias, and adding 128 zeros per lane and tile was ~1 % of the o / wo / qkv GEMMs (nothing overlaps an epilogue)
        if (p.bias != nullptr) {
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c = col_w + n * 32 + 8 * g + 4 * hh;
                    float b0 = 0.0f, b1 = 0.0f, b2 = 0.0f, b3 = 0.0f;
                    if (c < p.N) {
                        const uint2 bv = *reinterpret_cast<const uint2*>(p.bias + c);
                        b0 = bf2f((bf16_t)(bv.x & 0xffff));
                        b1 = bf2f((bf16_t)(bv.x >> 16));
                        b2 = bf2f((bf16_t)(bv.y & 0xffff));
                        b3 = bf2f((bf16_t)(bv.y >> 16));
                    }
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        acc[m][n][4 * g + 0] += b0; acc[m][n][4 * g + 1] += b1;
                        acc[m][n][4 * g + 2] += b2; acc[m][n][4 * g + 3] += b3;
                    }
                }
        }
        if constexpr (EPI == EPI_F32 || EPI == EPI_F32_RESID) {
#pragma unroll
            for (int m = 0; m < 4; ++m) {                     // 32 rows x 256 B per pass
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int ch = n * 8 + 2 * g + hh;    // 16-B chunk (4 fp32) within the 256-B row
                        const float4 v = make_float4(acc[m][n][4 * g + 0], acc[m][n][4 * g + 1], acc[m][n][4 * g + 2], acc[m][n][4 * g + 3]);
                        *reinterpret_cast<float4*>(reg + lr * 256 + ((ch ^ (lr & 15)) << 4)) = v;
                    }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int q = 0; q < 8; ++q) {                 // 4 rows x 256 B per instruction
                    const int r = q * 4 + (lane >> 4), c = lane & 15;
                    float4 v = *reinterpret_cast<const float4*>(reg + r * 256 + ((c ^ (r & 15)) << 4));
                    const int row = row_w + m * 32 + r, col = col_w + c * 4;
                    if (full || (row < p.M && col < p.N)) {
                        const size_t off = (size_t)bz * p.sC + (size_t)row * p.ldc + col;
                        if constexpr (EPI == EPI_F32_RESID) {
                            const float4 rv = *reinterpret_cast<const float4*>(p.resid + off);
                            v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
                        }
                        st_row16f(reinterpret_cast<float*>(p.C) + off, v, nt);
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
        } else {
            bf16_t* head_base = nullptr;
            int hx = p.H, hdim = 64;                           // heads of the target tensor, head width
            if constexpr (EPI == EPI_HEADS) {
                // columns [0, inner) -> q heads, [inner, inner + inner_kv) -> k heads, rest -> v heads; head width hd
                // (64, or 128: a wave's 64 columns are then one half of a head); grouped-query models have fewer k/v heads
                const int cw = min(col_w, p.N - 64);
                hdim = p.hd > 0 ? p.hd : 64;
                const int ikv = p.inner_kv > 0 ? p.inner_kv : p.inner;
                const int which = cw < p.inner ? 0 : (cw < p.inner + ikv ? 1 : 2);
                const int base = which == 0 ? 0 : (which == 1 ? p.inner : p.inner + ikv);
                hx = which == 0 ? p.H : (p.Hkv > 0 ? p.Hkv : p.H);
                bf16_t* hp = which == 0 ? p.heads_out[0] : (which == 1 ? p.heads_out[1] : p.heads_out[2]);
                const int head = (cw - base) / hdim;
                head_base = hp + (size_t)head * p.S * hdim + ((cw - base) - head * hdim);
            }
            // EPI_HEADS: this lane's 16 stores go to rows row_w + (lane >> 3) + 8k, k = 0..15, in that order: split the first
            // row into (sample, position) once, then step (needs S >= 8: the launcher sends shorter sequences to variant 0)
            int hs_run = 0;
            long long off_run = 0, off_wrap = 0;              // element offset of the current row; what a sample boundary adds
            if constexpr (EPI == EPI_HEADS) {
                heads_off_first(row_w + (lane >> 3), p.S, hx, hdim, hs_run, off_run);
                off_wrap = (long long)(hx - 1) * p.S * hdim;
            }
#pragma unroll
            for (int hm = 0; hm < 2; ++hm) {                  // 64 rows x 128 B per pass
#pragma unroll
                for (int m2 = 0; m2 < 2; ++m2) {
                    const int m = 2 * hm + m2;
                    const int r = m2 * 32 + lr;
#pragma unroll
                    for (int n = 0; n < 2; ++n)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            float o[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) o[e] = acc[m][n][4 * g + e];
                            if constexpr (EPI == EPI_BF16_QGELU) {
#pragma unroll
                                for (int e = 0; e < 4; ++e) o[e] = act_quick_gelu(o[e]);
                            }
                            if constexpr (EPI == EPI_BF16_GELU) {
#pragma unroll
                                for (int e = 0; e < 4; ++e) o[e] = act_gelu_erf(o[e]);
                            }
                            uint2 v;
                            v.x = OUT_F16 ? pack2h(o[0], o[1]) : pack2(o[0], o[1]);
                            v.y = OUT_F16 ? pack2h(o[2], o[3]) : pack2(o[2], o[3]);
                            const int ch = n * 4 + g;         // 16-B chunk (8 bf16) within the 128-B row; hh picks its half
                            *reinterpret_cast<uint2*>(reg + r * 128 + ((ch ^ (r & 7)) << 4) + hh * 8) = v;
                        }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int q = 0; q < 8; ++q) {                 // 8 rows x 128 B per instruction
                    const int r = q * 8 + (lane >> 3), c = lane & 7;
                    const uint4 v = *reinterpret_cast<const uint4*>(reg + r * 128 + ((c ^ (r & 7)) << 4));
                    const int row = row_w + hm * 64 + r, col = col_w + c * 8;
                    if (full || (row < p.M && col < p.N)) {
                        bf16_t* dst;
                        if constexpr (EPI == EPI_HEADS) {
                            dst = head_base + off_run + c * 8;
                        } else {
                            dst = reinterpret_cast<bf16_t*>(p.C) + (size_t)bz * p.sC + (size_t)row * p.ldc + col;
                        }
                        st_row16(dst, v, nt);
                    }
                    if constexpr (EPI == EPI_HEADS) heads_off_step8(p.S, hdim, off_wrap, hs_run, off_run);   // next row of this lane: + 8 (also across the hm passes)
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if constexpr (EPI == EPI_BF16) {
                    // split-bf16 result (GemmParams::split_off, wave-uniform): the same 64 rows once more, lo = bf16(acc - hi), to
                    // the lo plane.  Twice the stores of the counted waits' constant (EpiStores): a vmcnt(n) with MORE stores
                    // behind the staging DMA than n only waits longer (VMEM retires in order) -- never shorter.
                    if (p.split_off != 0) {
#pragma unroll
                        for (int m2 = 0; m2 < 2; ++m2) {
                            const int m = 2 * hm + m2;
                            const int r = m2 * 32 + lr;
#pragma unroll
                            for (int n = 0; n < 2; ++n)
#pragma unroll
                                for (int g = 0; g < 4; ++g) {
                                    float o[4];
#pragma unroll
                                    for (int e = 0; e < 4; ++e) o[e] = acc[m][n][4 * g + e];
                                    const uint32_t h0 = pack2(o[0], o[1]), h1 = pack2(o[2], o[3]);
                                    uint2 v;
                                    v.x = pack2(o[0] - __uint_as_float(h0 << 16), o[1] - __uint_as_float(h0 & 0xffff0000u));
                                    v.y = pack2(o[2] - __uint_as_float(h1 << 16), o[3] - __uint_as_float(h1 & 0xffff0000u));
                                    const int ch = n * 4 + g;
                                    *reinterpret_cast<uint2*>(reg + r * 128 + ((ch ^ (r & 7)) << 4) + hh * 8) = v;
                                }
                        }
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                        for (int q = 0; q < 8; ++q) {
                            const int r = q * 8 + (lane >> 3), c = lane & 7;
                            const uint4 v = *reinterpret_cast<const uint4*>(reg + r * 128 + ((c ^ (r & 7)) << 4));
                            const int row = row_w + hm * 64 + r, col = col_w + c * 8;
                            if (full || (row < p.M && col < p.N))
                                st_row16(reinterpret_cast<bf16_t*>(p.C) + p.split_off + (size_t)bz * p.sC + (size_t)row * p.ldc + col, v, nt);
                        }
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    }
                }
            }
        }
    }
}

template <int EPI>
struct EpiStores {   // VMEM store instructions per wave for a full tile (staged epilogue): 16-B per lane each
    static constexpr int value = (EPI == EPI_GATED) ? 8 : ((EPI == EPI_F32 || EPI == EPI_F32_RESID) ? 32 : (EPI == EPI_RESID_RMS ? 64 : 16));
};

typedef int v4i_t __attribute__((ext_vector_type(4)));
// MUBUF form of the LDS-DMA: SGPR descriptor + SGPR byte offset (the K-tile) + one 32-bit VGPR offset per lane, so the
// per-lane address never has to be recomputed or moved as 64 bits.
__device__ __forceinline__ void bglds16(v4i_t rsrc, uint32_t voff, uint32_t soff, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %1, %2, %4 offen lds\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(rsrc), "s"(lds_dst), "s"(__builtin_amdgcn_readfirstlane(soff)));
}
// dword form, used only to pull lines into L2: the 4 bytes per lane land in a scratch LDS area nobody reads
__device__ __forceinline__ void bglds4(v4i_t rsrc, uint32_t voff, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "buffer_load_dword %1, %2, 0 offen lds\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(rsrc), "s"(lds_dst));
}
// the same with an SGPR byte offset (the K-tile): L2 prefetch of a K-tile two ahead (TOUCH variant of the persistent kernel)
__device__ __forceinline__ void bglds4s(v4i_t rsrc, uint32_t voff, uint32_t soff, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "buffer_load_dword %1, %2, %4 offen lds\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(rsrc), "s"(lds_dst), "s"(__builtin_amdgcn_readfirstlane(soff)));
}
__device__ __forceinline__ v4i_t make_rsrc(const void* base) {
    const uint64_t b = (uint64_t)base;
    v4i_t r;
    r.x = __builtin_amdgcn_readfirstlane((uint32_t)b);
    r.y = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32) & 0xffffu);
    r.z = (int)0xffffffffu;
    r.w = 0x00020000;
    return r;
}

// TOUCH (lab, VQS_L2_TOUCH): waves 0 and 4 pull the lines of the K-tile TWO ahead into L2 with one dword LDS-DMA per
// K-tile (64 A rows / 32 W rows each: the 4 / 8 workgroups of an XCD's 8x4 tile window that share a panel split it), so
// that the real staging DMA of an operand streamed from HBM finds it in L2.  A hint only: results are unaffected.
// FT (round 5, GemmParams::f16; plain / fp32-result epilogues, TOUCH 0 only): 0 bf16 operands; 1 fp16 operands and fp16 result; 2 fp16 operands,
// bf16 (or split-bf16) result -- the batched launches of the decoder's cross-attention score path under option dec_fp16.
template <int EPI, int TOUCH = 0, int FT = 0>   // TOUCH: 0 off, 1 A and W panels, 2 A panel only
__global__ void __launch_bounds__(512) gemm_bf16_persistent(const GemmParams p) {
    __shared__ __attribute__((aligned(16))) char lds[2 * STAGE_BYTES];
    __shared__ float rowred[EPI == EPI_RESID_RMS ? 1024 : 1];   // per-row partial sums of squares of the 4 wave columns
    __shared__ __attribute__((aligned(16))) char touch_sink[(EPI == EPI_RESID_RMS || TOUCH != 0) ? 2048 : 16];   // landing area of the L2-touch DMA

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = w >> 2, wc = w & 3;
    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
    const int tiles_pb = tiles_m * tiles_n;                 // tiles per batch entry
    const int nbatch = p.batch > 0 ? p.batch : 1;
    const int nwg = tiles_pb * nbatch;
    const int nt = p.K / BK;
    const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)LDS_PTR(lds));

    auto tile_coords = [&](int pid, int& m0, int& n0, int& bz) {
        tile_of_slot(pid, nwg, tiles_m, tiles_n, p.tile_gm, p.tile_ns, m0, n0, bz);
    };

    const int sw = ((w & 1) << 2) + (lane >> 4);
    const int gchunk = (lane & 7) ^ sw;
    uint32_t pa[4], pb[4];          // byte offsets from the batch entry's base (every operand is < 4 GiB)
    v4i_t rsA, rsW;
    auto set_ptrs = [&](int m0, int n0, int bz) {
        rsA = make_rsrc(p.A + (size_t)bz * p.sA);
        rsW = make_rsrc(p.W + (size_t)bz * p.sW);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = (i * 8 + w) * 8 + (lane >> 3);
            pa[i] = (uint32_t)(((size_t)min(m0 + row, p.M - 1) * p.lda + gchunk * 8) * 2);
            pb[i] = (uint32_t)(((size_t)min(n0 + row, p.N - 1) * p.ldw + gchunk * 8) * 2);
        }
    };
#define PGLDS_A(i, koffs, dst) bglds16(rsA, pa[i], (uint32_t)((koffs) * 2), (dst))
#define PGLDS_W(i, koffs, dst) bglds16(rsW, pb[i], (uint32_t)((koffs) * 2), (dst))
    auto stage = [&](int s, int t) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t da = lds_base + s * STAGE_BYTES + (i * 8 + w) * 1024;
            PGLDS_A(i, (size_t)t * BK, da);
            PGLDS_W(i, (size_t)t * BK, da + W_OFF);
        }
    };

    const int swr = (lane >> 1) & 7;
    int koff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) koff[ks] = (((ks * 2 + (lane >> 5)) ^ swr) << 4);
    const int a_row = (wr * 128 + (lane & 31)) * 128;
    const int b_row = W_OFF + (wc * 64 + (lane & 31)) * 128;

    int pid = blockIdx.x;
    if (pid >= nwg) return;
    int m0, n0, bz;
    tile_coords(pid, m0, n0, bz);
    set_ptrs(m0, n0, bz);
    stage(0, 0);
    int buf = 0;
    bool counted = false;     // true: the only VMEM ops younger than the prefetched K-tile are a full epilogue's stores
    // TOUCH state (batch <= 1 only, enforced by the launcher)
    constexpr int TDIST = 2;            // K-tiles ahead (3 measured the same)
    const bool touch_wave = TOUCH != 0 && (w == 0 || (w == 4 && TOUCH == 1)) && nt >= TDIST;
    const v4i_t rsT = make_rsrc(w < 4 ? (const void*)p.A : (const void*)p.W);
    const uint32_t touch_dst = (uint32_t)(uintptr_t)LDS_PTR(touch_sink) + w * 256;
    auto touch_off = [&](int tm0, int tn0) -> uint32_t {
        if (w < 4) {
            const int row = min(tm0 + 64 * ((tn0 / BN) & 3) + lane, p.M - 1);
            return (uint32_t)((size_t)row * p.lda * 2);
        }
        const int row = min(tn0 + 32 * ((tm0 / BM) & 7) + (lane & 31), p.N - 1);
        return (uint32_t)((size_t)row * p.ldw * 2);
    };
    uint32_t t_cur = 0, t_nxt = 0;
    bool did_touch = false;

    while (true) {
        f32x16 acc[4][2];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.0f;

        const int next_pid = pid + gridDim.x;
        const bool has_next = next_pid < nwg;
        int nm0 = 0, nn0 = 0, nbz = 0;
        if constexpr (TOUCH) {
            if (touch_wave) {
                t_cur = touch_off(m0, n0);
                if (has_next) {
                    int xm0, xn0, xbz;
                    tile_coords(next_pid, xm0, xn0, xbz);
                    t_nxt = touch_off(xm0, xn0);
                }
            }
        }

        for (int t = 0; t < nt; ++t) {
            if (t == 0 && counted) {
                if constexpr (EpiStores<EPI>::value == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else if constexpr (EpiStores<EPI>::value == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                else if constexpr (EpiStores<EPI>::value == 64) asm volatile("s_waitcnt vmcnt(63)" ::: "memory");   // counter is 6 bits
                else asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
            } else if (TOUCH != 0 && t > 0 && did_touch) {
                asm volatile("s_waitcnt vmcnt(1)" ::: "memory");     // the touch issued after this K-tile's DMA may still fly
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            did_touch = false;
            __builtin_amdgcn_s_barrier();
            // source of the stage that is filled during this K-tile: the next K-tile of this tile, or the first
            // K-tile of the next tile.  The 8 LDS-DMA pieces are issued two per k-step BETWEEN the MFMA groups:
            // issued back to back they cost 600-1700 cycles per wave (measured) during which the wave's matrix
            // pipe idles and the early waves then sit at the barrier.
            int kn = t + 1;
            bool do_stage = true;
            if (t + 1 >= nt) {
                kn = 0;
                do_stage = has_next;
                if (has_next) {
                    tile_coords(next_pid, nm0, nn0, nbz);
                    set_ptrs(nm0, nn0, nbz);
                }
            }
            const size_t koffs = (size_t)kn * BK;
            const uint32_t dst0 = lds_base + (buf ^ 1) * STAGE_BYTES + w * 1024;
            const char* sb = lds + buf * STAGE_BYTES;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                uint4 af[4], wf[2];
#pragma unroll
                for (int m = 0; m < 4; ++m) af[m] = *reinterpret_cast<const uint4*>(sb + a_row + m * 4096 + koff[ks]);
#pragma unroll
                for (int n = 0; n < 2; ++n) wf[n] = *reinterpret_cast<const uint4*>(sb + b_row + n * 4096 + koff[ks]);
                if constexpr (EPI == EPI_RESID_RMS) {
                    // last K-tile of the tile: touch every 128-B line of the tile's slice of the residual stream (256 rows
                    // x 1 KiB = 2048 lines, 4 per thread) so that the epilogue's read-modify-write finds it in L2
                    if (t == nt - 1 && ks == 2) {
                        const v4i_t rsH = make_rsrc(p.hres);
                        const uint32_t sink = (uint32_t)(uintptr_t)LDS_PTR(touch_sink) + w * 256;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int l = j * 512 + w * 64 + lane;
                            const int row = min(m0 + (l >> 3), p.M - 1);
                            const int colf = min(n0 + (l & 7) * 32, p.N - 1);
                            bglds4(rsH, (uint32_t)(((size_t)row * p.ldh + colf) * 4), sink);
                        }
                    }
                }
                if constexpr (TOUCH) {
                    if (ks == 2 && touch_wave) {
                        int kt2 = t + TDIST;
                        uint32_t voff = t_cur;
                        bool ok = true;
                        if (kt2 >= nt) {
                            kt2 -= nt;
                            voff = t_nxt;
                            ok = has_next;
                        }
                        if (ok) {
                            bglds4s(rsT, voff, (uint32_t)(kt2 * BK * 2), touch_dst);
                            did_touch = true;
                        }
                    }
                }
                // the older wave of a SIMD wins issue arbitration all the time (measured: its 32 MFMAs take ~2000
                // cycles, the younger wave's ~2600, and the older half then idles at the barrier): hand the
                // younger half (waves 4-7) priority for the first two k-steps of every K-tile
                if (ks == 0 && wr == 1) __builtin_amdgcn_s_setprio(1);
                if (ks == 2 && wr == 1) __builtin_amdgcn_s_setprio(0);
                // all eight pieces in the first two k-steps: the last piece has >= 2 k-steps (~1 300 cycles) to land
                // instead of one -- for operands streamed from HBM (the K = 1024 ViT shapes wait 350-900 cycles per K-tile)
                if (do_stage && ks < 2) {
                    const int i0 = ks * 2;
                    PGLDS_A(i0, koffs, dst0 + i0 * 8192);
                    PGLDS_W(i0, koffs, dst0 + i0 * 8192 + W_OFF);
                    PGLDS_A(i0 + 1, koffs, dst0 + (i0 + 1) * 8192);
                    PGLDS_W(i0 + 1, koffs, dst0 + (i0 + 1) * 8192 + W_OFF);
                }
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n) {
                        if constexpr (FT != 0)
                            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                                __builtin_bit_cast(f16x8_t, wf[n]), __builtin_bit_cast(f16x8_t, af[m]), acc[m][n], 0, 0, 0);
                        else
                            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                                __builtin_bit_cast(bf16x8, wf[n]), __builtin_bit_cast(bf16x8, af[m]), acc[m][n], 0, 0, 0);
                    }
            }
            buf ^= 1;
        }

        // ---------------- epilogue: C tile staged through the LDS stage that was just consumed, stored as whole rows.
        // (Storing straight from the accumulator layout touches 32 rows x 16 B per instruction and measured 10-20 K
        // cycles per tile; staged, every store instruction writes 8 full 128-B rows / 4 full 256-B rows.)
        const bool full = (m0 + BM <= p.M) && (n0 + BN <= p.N);
        __builtin_amdgcn_s_barrier();                       // every wave is done reading stage buf^1
        staged_epilogue<EPI, FT == 1>(p, acc, lds + (buf ^ 1) * STAGE_BYTES + w * 8192, m0, n0, bz, wr, wc, lane, full, rowred);
        counted = full;
        if (!has_next) break;
        pid = next_pid;
        m0 = nm0;
        n0 = nn0;
        bz = nbz;
    }
}


#undef PGLDS_A
#undef PGLDS_W





#include "gemm_quad.inc"
#include "gemm_stream.inc"
#include "gemm_slim.inc"

// =====================================================================================================
// Persistent PING-PONG variant (VAR 5).  Same tile, LDS image, tile map and staged epilogue as the persistent kernel
// above; what changes is WHO feeds the matrix pipe WHEN.  The two waves that share a SIMD (w and w+4, wave row
// wr = 0 / 1) run one barrier apart: while one is in an MFMA segment (8 MFMAs = one quadrant of its 128x64 output
// over the whole K-tile, 256 pipe cycles) the other is in a LOAD segment (ds_read_b128 of its next quadrant's
// fragments + two LDS-DMA pieces), then they swap.  A wave never issues an LDS-DMA or a ds_read between its own
// MFMAs, so the 60-180 cycles a DMA piece costs its issuer are spent while the other wave owns the pipe.
//
//   K-tile = 4 phases: Q1 (A0,W0)  Q2 (A0,W1)  Q3 (A1,W1)  Q4 (A1,W0)     A_i = wave rows m in {2i,2i+1}; W_j = n = j
//   loads per phase  : Q1 A0+W0 (12 b128)   Q2 W1 (4)   Q3 A1 (8)   Q4 W0 (4)
//
// Staging runs as ONE stream of 16-KiB half-tiles (A0 W0 W1 A1 per K-tile, 2 pieces per wave each) three half-tiles
// ahead of the compute, across K-tiles and across output tiles: phase Q1 issues this K-tile's A1, Q2..Q4 the next
// K-tile's A0, W0, W1 (into the other stage).  Waits are counted, never 0 in steady state:
//   a half-tile first read in phase p must be waited for (by EVERY wave, for its own pieces) before the barrier that
//   opens wave-row 0's phase p: wave row 0 waits at the end of its MFMA segment p-1, wave row 1 at the end of its
//   LOAD segment p-1 (same barrier).  VMEM retires in order, so vmcnt(2) = "all but the newest half-tile".
//   After a full tile's epilogue the first wait allows the epilogue's stores on top (they are younger than W1).
// Tile end: wave row 0 idles one barrier so both rows run the epilogue together, then row 1 idles one to re-stagger.
// =====================================================================================================
#define VQS_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define VQS_PIN() __builtin_amdgcn_sched_barrier(0)
#define VQS_BAR()                           \
    do {                                    \
        __builtin_amdgcn_sched_barrier(0);  \
        __builtin_amdgcn_s_barrier();       \
        __builtin_amdgcn_sched_barrier(0);  \
    } while (0)

template <int EPI>
__global__ void __launch_bounds__(512) gemm_bf16_pingpong(const GemmParams p) {
    __shared__ __attribute__((aligned(16))) char lds[2 * STAGE_BYTES];
    __shared__ float rowred[EPI == EPI_RESID_RMS ? 1024 : 1];   // per-row partial sums of squares of the 4 wave columns

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = w >> 2, wc = w & 3;
    const bool g1 = (wr == 1);
    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
    const int tiles_pb = tiles_m * tiles_n;
    const int nbatch = p.batch > 0 ? p.batch : 1;
    const int nwg = tiles_pb * nbatch;
    const int nt = p.K / BK;
    const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)LDS_PTR(lds));

    auto tile_coords = [&](int pid, int& m0, int& n0, int& bz) {
        tile_of_slot(pid, nwg, tiles_m, tiles_n, p.tile_gm, p.tile_ns, m0, n0, bz);
    };

    // ---- staging stream.  Half-tile A_i = tile rows {c*128 + i*64 + 0..63}, W_j = rows {q*64 + j*32 + 0..31};
    // each is 16 pieces of 8 rows, wave w takes pieces w and w+8 (c = 0 / 1).  Same XOR chunk swizzle as above.
    const int sw = ((w & 1) << 2) + (lane >> 4);
    const int gchunk = (lane & 7) ^ sw;
    uint32_t pa[4], pb[4];                 // [half * 2 + c]: byte offset of this lane's 16 B (K-tile 0) from the batch base
    v4i_t rsA, rsW;
    auto set_ptrs = [&](int m0, int n0, int bz) {
        rsA = make_rsrc(p.A + (size_t)bz * p.sA);
        rsW = make_rsrc(p.W + (size_t)bz * p.sW);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int ra = c * 128 + h * 64 + w * 8 + (lane >> 3);
                const int rw = (c * 2 + (w >> 2)) * 64 + h * 32 + (w & 3) * 8 + (lane >> 3);
                pa[h * 2 + c] = (uint32_t)(((size_t)min(m0 + ra, p.M - 1) * p.lda + gchunk * 8) * 2);
                pb[h * 2 + c] = (uint32_t)(((size_t)min(n0 + rw, p.N - 1) * p.ldw + gchunk * 8) * 2);
            }
    };
    int s_pid = blockIdx.x;
    if (s_pid >= nwg) return;
    int s_kt = 0, s_buf = 0;
    bool s_valid = true;
    auto issue_a = [&](int h) {
        if (!s_valid) return;
        const uint32_t base = lds_base + s_buf * STAGE_BYTES + h * 8192 + w * 1024;
        const uint32_t soff = (uint32_t)s_kt * (BK * 2);
        bglds16(rsA, pa[h * 2 + 0], soff, base);
        bglds16(rsA, pa[h * 2 + 1], soff, base + 16384);
    };
    auto issue_w = [&](int h) {
        if (!s_valid) return;
        const uint32_t base = lds_base + s_buf * STAGE_BYTES + W_OFF + ((w >> 2) * 64 + h * 32 + (w & 3) * 8) * 128;
        const uint32_t soff = (uint32_t)s_kt * (BK * 2);
        bglds16(rsW, pb[h * 2 + 0], soff, base);
        bglds16(rsW, pb[h * 2 + 1], soff, base + 16384);
    };
    auto advance = [&]() {                 // stream moves on to the next K-tile (possibly the next tile's first)
        s_buf ^= 1;
        if (++s_kt == nt) {
            s_kt = 0;
            s_pid += gridDim.x;
            if (s_pid < nwg) {
                int sm0, sn0, sbz;
                tile_coords(s_pid, sm0, sn0, sbz);
                set_ptrs(sm0, sn0, sbz);
            } else {
                s_valid = false;
            }
        }
    };

    // ---- fragment addressing (as in the persistent kernel)
    const int swr = (lane >> 1) & 7;
    int koff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) koff[ks] = (((ks * 2 + (lane >> 5)) ^ swr) << 4);
    const int a_row = (wr * 128 + (lane & 31)) * 128;
    const int b_row = W_OFF + (wc * 64 + (lane & 31)) * 128;

    int pid = blockIdx.x;
    int m0, n0, bz;
    tile_coords(pid, m0, n0, bz);
    set_ptrs(m0, n0, bz);
    issue_a(0);
    issue_w(0);
    issue_w(1);
    VQS_VMCNT(2);
    VQS_BAR();
    if (g1) VQS_BAR();                     // wave row 1 runs one barrier behind from here on

    int cb = 0;                            // LDS stage of the K-tile being computed
    bool after_epi = false;                // this tile follows a FULL tile's epilogue (its stores are still counted)

    while (true) {
        f32x16 acc[4][2];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.0f;

        for (int t = 0; t < nt; ++t) {
            const char* sb = lds + cb * STAGE_BYTES;
            uint4 af[2][4], wf[4];

#define VQS_LOAD_A(H)                                                                                              \
    _Pragma("unroll") for (int mm = 0; mm < 2; ++mm) _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) af[mm][ks] = \
        *reinterpret_cast<const uint4*>(sb + a_row + ((H) * 2 + mm) * 4096 + koff[ks])
#define VQS_LOAD_W(J) \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) wf[ks] = *reinterpret_cast<const uint4*>(sb + b_row + (J) * 4096 + koff[ks])
#define VQS_MMA(H, J)                                                                                        \
    do {                                                                                                     \
        __builtin_amdgcn_s_setprio(1);                                                                       \
        _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) _Pragma("unroll") for (int mm = 0; mm < 2; ++mm)    \
            acc[(H) * 2 + mm][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(                                  \
                __builtin_bit_cast(bf16x8, wf[ks]), __builtin_bit_cast(bf16x8, af[mm][ks]), acc[(H) * 2 + mm][J], 0, 0, 0); \
        __builtin_amdgcn_s_setprio(0);                                                                       \
    } while (0)
#define VQS_END_LOAD(WAIT)                                  \
    do {                                                    \
        VQS_PIN();                                          \
        if (g1) { WAIT; }                                   \
        VQS_BAR();                                          \
    } while (0)
#define VQS_END_MMA(WAIT)        \
    do {                         \
        VQS_PIN();               \
        if (!g1) { WAIT; }       \
        VQS_BAR();               \
    } while (0)
#define VQS_WAIT_Q1                                                        \
    do {                                                                   \
        if (t == 0 && after_epi) {                                         \
            if constexpr (EpiStores<EPI>::value == 8) VQS_VMCNT(10);       \
            else if constexpr (EpiStores<EPI>::value == 16) VQS_VMCNT(18); \
            else if constexpr (EpiStores<EPI>::value == 64) VQS_VMCNT(63); \
            else VQS_VMCNT(34);                                            \
        } else {                                                           \
            VQS_VMCNT(2);                                                  \
        }                                                                  \
    } while (0)
#define VQS_WAIT_NEXT                   \
    do {                                \
        if (s_valid) VQS_VMCNT(2);      \
        else VQS_VMCNT(0);              \
    } while (0)
#define VQS_WAIT_NONE \
    do {              \
    } while (0)

            // ---- Q1: A0 x W0; stages this K-tile's A1, then the stream moves to the next K-tile
            VQS_LOAD_A(0);
            VQS_LOAD_W(0);
            issue_a(1);
            advance();
            VQS_END_LOAD(VQS_WAIT_Q1);            // for Q2: W1 of this K-tile
            VQS_MMA(0, 0);
            VQS_END_MMA(VQS_WAIT_Q1);
            // ---- Q2: A0 x W1; stages the next K-tile's A0
            VQS_LOAD_W(1);
            issue_a(0);
            VQS_END_LOAD(VQS_WAIT_NEXT);          // for Q3: A1 of this K-tile
            VQS_MMA(0, 1);
            VQS_END_MMA(VQS_WAIT_NEXT);
            // ---- Q3: A1 x W1; stages the next K-tile's W0
            VQS_LOAD_A(1);
            issue_w(0);
            VQS_END_LOAD(VQS_WAIT_NONE);          // Q4 reads W0 again: landed since Q1
            VQS_MMA(1, 1);
            VQS_END_MMA(VQS_WAIT_NONE);
            // ---- Q4: A1 x W0; stages the next K-tile's W1
            VQS_LOAD_W(0);
            issue_w(1);
            VQS_END_LOAD(VQS_WAIT_NEXT);          // for the next K-tile's Q1: its A0 and W0
            VQS_MMA(1, 0);
            VQS_END_MMA(VQS_WAIT_NEXT);
            cb ^= 1;
        }

        // ---------------- tile end: both wave rows run the epilogue together, staged through the stage just consumed
        const bool full = (m0 + BM <= p.M) && (n0 + BN <= p.N);
        if (!g1) VQS_BAR();
        staged_epilogue<EPI>(p, acc, lds + (cb ^ 1) * STAGE_BYTES + w * 8192, m0, n0, bz, wr, wc, lane, full, rowred);
        after_epi = full;
        pid += gridDim.x;
        if (pid >= nwg) break;
        tile_coords(pid, m0, n0, bz);
        if (g1) VQS_BAR();
    }
}
#undef VQS_LOAD_A
#undef VQS_LOAD_W
#undef VQS_MMA
#undef VQS_END_LOAD
#undef VQS_END_MMA
#undef VQS_WAIT_Q1
#undef VQS_WAIT_NEXT
#undef VQS_WAIT_NONE

// VQS_L2_TOUCH: 4 (default) = A-panel touch for N <= 2048; 3 = A and W panels for N <= 2048; 1 = A and W for every
// lock-step launch; 0 = off
static int l2_touch_mode() {
    return 4;
}

// Which launches the quad form carries: a function of the epilogue and the WEIGHT's shape only (never of M).
static bool quad_eligible_rt(const GemmParams& p, int epi) {
    if (!(epi == EPI_BF16 || epi == EPI_BF16_QGELU || epi == EPI_BF16_GELU || epi == EPI_GATED || epi == EPI_HEADS)) return false;
    if (p.batch > 1 || p.K < 2 * BK || p.rowss_in != nullptr || p.split_off != 0) return false;
    if (epi == EPI_GATED && (p.N % 64) != 0) return false;
    if (epi == EPI_HEADS) {
        // a wave's 128 columns must lie inside ONE of the q / k / v tensors (one head count per block); rows step by 4 within a sample
        const int ikv = p.inner_kv > 0 ? p.inner_kv : p.inner;
        if (p.S < 8 || (p.inner % 128) != 0 || (ikv % 128) != 0) return false;
        if (p.hd_src > 0 && ((p.hd_src % 8) != 0 || p.hd_src > (p.hd > 0 ? p.hd : 64))) return false;
    }
    return true;
}
template <int EPI>
static bool quad_eligible(const GemmParams& p) { return quad_eligible_rt(p, EPI); }
// Kernel family a launch resolves to (host arithmetic, shared with the test hook vqs_debug_gemm_form): 10 quad form, 0 one tile per
// workgroup (64-bit pointers), 3 an 8-wave persistent / ping-pong kernel or a lab form (32-bit byte offsets into a batch entry's
// operands), -1 not launchable.  The 32-bit forms refuse operands of 4 GiB or more per batch entry; a plain single-entry launch
// then falls back to the one-tile-per-workgroup kernel (same bits), anything else is an error -- never a wrapped offset.
int gemm_form(const GemmParams& p, int epilogue, int variant) {
    if (p.f16)      // fp16 operands: the quad form only (every 16-bit-result linear of the vision tower, the projector and -- option enc_fp16 --
                    // the T5 encoder's attention side resolves to it); 1 = fp16 result (no gated epilogue), 2 = bf16 result (plain / gated only)
    {
        if ((variant == 3 || variant == 10) && quad_eligible_rt(p, epilogue))
            return (p.f16 == 4 ? epilogue == EPI_BF16 : p.f16 == 3 ? epilogue != EPI_BF16_QGELU : p.f16 == 1 ? epilogue != EPI_GATED : (p.f16 == 2 && (epilogue == EPI_BF16 || epilogue == EPI_GATED))) ? 10 : -1;
        if (p.f16 >= 3) return -1;      // the scaled families exist in the quad form only
        // round 5: the batched / fp32-result launches of the decoder's cross-attention score path (option dec_fp16) -- plain epilogues of the
        // stream form and of the persistent 8-wave kernel (bias-free, no row scale; fp32 result, fp16 result, bf16 / split-bf16 result)
        if (variant != 3 || !(epilogue == EPI_BF16 || epilogue == EPI_F32) || p.bias != nullptr || p.rowss_in != nullptr) return -1;
        if (p.f16 == 1 && p.split_off != 0) return -1;
        if (stream_eligible(p, epilogue)) return 12;
        const bool fit16 = (uint64_t)p.M * (uint64_t)p.lda * 2ull < (1ull << 32) && (uint64_t)p.N * (uint64_t)p.ldw * 2ull < (1ull << 32);
        return fit16 ? 3 : -1;
    }
    if (epilogue == EPI_RESID_RMS) variant = variant == 5 ? 5 : 3;
    const bool plain_v0 = !(p.hd > 64 || p.hd_src > 0 || p.inner_kv > 0 || p.Hkv > 0 || p.gate_act != 0 || (epilogue == EPI_GATED && p.bias != nullptr) || p.rowss_in != nullptr ||
                            p.split_off != 0) &&
                          p.batch <= 1 && epilogue != EPI_RESID_RMS;
    if (variant == 0 || variant == 2 || variant == 1) return plain_v0 ? 0 : -1;
    if (epilogue == EPI_HEADS && p.S < 8) return plain_v0 ? 0 : -1;
    if ((variant == 3 || variant == 10) && quad_eligible_rt(p, epilogue)) return 10;       // first: a quad call site stays quad for every M
    if ((variant == 3 || variant == 11) && stream_eligible(p, epilogue)) return 12;         // stream form: bitwise the 8-wave forms (gemm_stream.inc)
    if (epilogue == EPI_HEADS && p.hd_src > 0 && p.hd_src != (p.hd > 0 ? p.hd : 64)) return -1;   // narrow heads in wide slots: quad form only
    if (epilogue == EPI_F32_RESID) return p.batch <= 1 ? 0 : -1;
    const bool fit = (uint64_t)p.M * (uint64_t)p.lda * 2ull < (1ull << 32) && (uint64_t)p.N * (uint64_t)p.ldw * 2ull < (1ull << 32);
    if (fit) return 3;
    return plain_v0 ? 0 : -1;
}

bool gemm_takes_slim(const GemmParams& p, int epilogue, int variant) {
    return variant == 3 && !p.no_stream && gemm_form(p, epilogue, variant) == 10 && slim_eligible(p, epilogue);
}

template <int EPI>
static hipError_t launch_epi(const GemmParams& p, int variant, hipStream_t stream) {   // NOLINT(variant is resolved below)
    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
    dim3 grid(tiles_m * tiles_n), block(512);
    // variant 11 = the 8-wave forms by the round-1/2 shape rule for every launch (A/B against the quad form; below it is variant 3)
    const int form = gemm_form(p, EPI, variant);
    if (form < 0) return hipErrorInvalidValue;
    if (form == 0 && variant != 2 && variant != 1) variant = 0;      // incl. the fallback of a >= 4 GiB operand from the 32-bit forms
    if constexpr (EPI == EPI_RESID_RMS) {
        // only the persistent kernels carry this epilogue (LDS-staged, needs the cross-wave row reduction)
        const int nwg = tiles_m * tiles_n;
        dim3 pgrid(nwg < PERSISTENT_WGS ? nwg : PERSISTENT_WGS);
        if (variant == 5)
            hipLaunchKernelGGL((gemm_bf16_pingpong<EPI>), pgrid, block, 0, stream, p);
        else
            hipLaunchKernelGGL((gemm_bf16_persistent<EPI>), pgrid, block, 0, stream, p);
        return hipGetLastError();
    } else
    if (variant == 0 || (EPI == EPI_HEADS && p.S < 8 && variant != 1 && variant != 2))   // the staged epilogue steps rows by 8 within a sample
        hipLaunchKernelGGL((gemm_bf16_kernel<EPI, 0>), grid, block, 0, stream, p);
    else if (variant == 2)
        hipLaunchKernelGGL((gemm_bf16_kernel<EPI, 2>), grid, block, 0, stream, p);
    else if (form == 12) {
        if constexpr (EPI == EPI_F32 || EPI == EPI_BF16) {
            const int items = ((p.N + 127) / 128) * (p.batch > 0 ? p.batch : 1);
            if (p.f16 == 1) hipLaunchKernelGGL((gemm_bf16_stream<EPI, 1>), dim3(items), dim3(256), 0, stream, p);
            else if (p.f16 == 2) hipLaunchKernelGGL((gemm_bf16_stream<EPI, 2>), dim3(items), dim3(256), 0, stream, p);
            else hipLaunchKernelGGL((gemm_bf16_stream<EPI>), dim3(items), dim3(256), 0, stream, p);
        }
    } else if (p.f16 != 0 && form == 3) {
        // fp16 operands on the persistent 8-wave kernel (plain schedule, no touch): gemm_form let only EPI_BF16 / EPI_F32 through
        if constexpr (EPI == EPI_F32 || EPI == EPI_BF16) {
            const int nwg = tiles_m * tiles_n * (p.batch > 0 ? p.batch : 1);
            dim3 pgrid(nwg < PERSISTENT_WGS ? nwg : PERSISTENT_WGS);
            if (p.f16 == 1) hipLaunchKernelGGL((gemm_bf16_persistent<EPI, 0, 1>), pgrid, block, 0, stream, p);
            else hipLaunchKernelGGL((gemm_bf16_persistent<EPI, 0, 2>), pgrid, block, 0, stream, p);
        } else {
            return hipErrorInvalidValue;
        }
    } else if (form == 10) {
        // quad form (gemm_quad.inc): every bf16-result launch of a call site, WHATEVER its M -- the form's k-order differs from the
        // 8-wave forms' in the last ulp, so a weight must not change form with the batch size (a pair's bits are batch-invariant)
        if constexpr (EPI == EPI_BF16 || EPI == EPI_BF16_QGELU || EPI == EPI_BF16_GELU || EPI == EPI_GATED || EPI == EPI_HEADS) {
            // Few-row launches (a caller that scores one or two pairs): the slim form, gemm_slim.inc -- same bits, so the switch is a function of the launch's shape
            // that never shows in a score.  variant 10 = the quad kernel whatever the shape (tests, A/B).  (As 128-row blocks of the STREAM form the same launches
            // were slower than the quad kernel: its 128 x 32 wave tiles are built to keep HBM busy, not the matrix pipe -- profiles/r6_call24_*.)
            if (gemm_takes_slim(p, EPI, variant)) {
                const dim3 sgrid(((p.M + 127) / 128) * (p.N / 128));
                if (p.f16 == 2) {
                    if constexpr (EPI == EPI_BF16 || EPI == EPI_GATED) hipLaunchKernelGGL((gemm_slim<EPI, 2>), sgrid, dim3(256), 0, stream, p);
                } else if (p.f16 == 1) {
                    if constexpr (EPI != EPI_GATED) hipLaunchKernelGGL((gemm_slim<EPI, 1>), sgrid, dim3(256), 0, stream, p);
                } else {
                    hipLaunchKernelGGL((gemm_slim<EPI, 0>), sgrid, dim3(256), 0, stream, p);
                }
                return hipGetLastError();
            }
            const int nwg = tiles_m * tiles_n;
            const dim3 qgrid(nwg < VQS_QUAD_WGS ? nwg : VQS_QUAD_WGS);
            if (p.f16 == 4) {
                if constexpr (EPI == EPI_BF16) hipLaunchKernelGGL((gemm_f16bs_quad<EPI>), qgrid, dim3(256), 0, stream, p);
                else return hipErrorInvalidValue;
            } else if (p.f16 == 3) {
                if constexpr (EPI != EPI_BF16_QGELU) hipLaunchKernelGGL((gemm_f16s_quad<EPI>), qgrid, dim3(256), 0, stream, p);
                else return hipErrorInvalidValue;
            } else if (p.f16 == 2) {
                if constexpr (EPI == EPI_BF16 || EPI == EPI_GATED) hipLaunchKernelGGL((gemm_f16b_quad<EPI>), qgrid, dim3(256), 0, stream, p);
                else return hipErrorInvalidValue;
            } else if (p.f16) {
                if constexpr (EPI != EPI_GATED) hipLaunchKernelGGL((gemm_f16_quad<EPI>), qgrid, dim3(256), 0, stream, p);
                else return hipErrorInvalidValue;
            } else {
                hipLaunchKernelGGL((gemm_bf16_quad<EPI>), qgrid, dim3(256), 0, stream, p);
            }
        }
    } else if (variant == 5 && EPI != EPI_F32_RESID) {
        if constexpr (EPI != EPI_F32_RESID) {
            const int nwg = tiles_m * tiles_n * (p.batch > 0 ? p.batch : 1);
            dim3 pgrid(nwg < PERSISTENT_WGS ? nwg : PERSISTENT_WGS);
            hipLaunchKernelGGL((gemm_bf16_pingpong<EPI>), pgrid, block, 0, stream, p);
        }
    } else {
        if constexpr (EPI == EPI_F32_RESID) {
            // the in-epilogue fp32 read-modify-write does not fit the persistent kernel's register budget (it
            // spills); the engine's hot path uses EPI_F32 + a fused add in the following norm kernel instead
            hipLaunchKernelGGL((gemm_bf16_kernel<EPI, 0>), grid, block, 0, stream, p);
        } else {
            const int nwg = tiles_m * tiles_n * (p.batch > 0 ? p.batch : 1);
            dim3 pgrid(nwg < PERSISTENT_WGS ? nwg : PERSISTENT_WGS);
            // L2 touch (see gemm_bf16_persistent): shapes whose A panel has at most two window columns of reuse (N <= 2048:
            // ViT out_proj / fc2, T5 o / wo) stream A from HBM and gain 2-8 % from the prefetch; wider shapes re-read A
            // from L2 / Infinity Cache and lose 1-3 % to the extra requests, so they keep the plain kernel.  Exception: the
            // XXL wo GEMM (N = 4096, K = 10240) -- its 8-M-tile A panels (42 MB per XCD, 336 MB per chip) overflow the
            // 256 MB Infinity Cache, so the re-reads come from HBM: +1.4 % with the touch (1 231 -> 1 248 TFLOP/s, A/B r2 call 21).
            const int tm = l2_touch_mode();
            const bool touch_ok = tm != 0 && p.batch <= 1 && nwg >= PERSISTENT_WGS && p.K >= 4 * BK;
            const bool touch = touch_ok && (p.l2_touch == 1 || (p.l2_touch == 0 && (tm == 1 || p.N <= 2048 || (p.N <= 4096 && p.K >= 8192))));   // l2_touch: caller's override (1 on, 2 off)
            // schedule by shape: the ViT qkv (K = 1024, 2048 < N <= 3072) is 0-1 % faster on the ping-pong schedule; with
            // the touch the lock-step kernel wins on out_proj (+7 %).  All of them produce bitwise-identical results.
            if (variant != 7 && !touch && p.K <= 1024 && p.N <= 3072 && p.batch <= 1 && nwg >= PERSISTENT_WGS)   // 7 = lock-step forced (lab)
                hipLaunchKernelGGL((gemm_bf16_pingpong<EPI>), pgrid, block, 0, stream, p);
            else if (touch && (tm == 1 || tm == 3))
                hipLaunchKernelGGL((gemm_bf16_persistent<EPI, 1>), pgrid, block, 0, stream, p);
            else if (touch)
                hipLaunchKernelGGL((gemm_bf16_persistent<EPI, 2>), pgrid, block, 0, stream, p);
            else
                hipLaunchKernelGGL((gemm_bf16_persistent<EPI>), pgrid, block, 0, stream, p);
        }
    }
    return hipGetLastError();
}


hipError_t launch_gemm(const GemmParams& p_in, int epilogue, int variant, hipStream_t stream) {
    GemmParams p = p_in;
    if (p.rope_cos != nullptr || p.rope_sin != nullptr) {     // rotary embedding in the epilogue: scaled fp16 family, head-major scatter of whole 128-lane heads only
        if (!(p.rope_cos && p.rope_sin) || p.f16 != 3 || epilogue != EPI_HEADS || p.hd != 128 || (p.hd_src != 0 && p.hd_src != 128)) return hipErrorInvalidValue;
    }
    resolve_tile_order(p, PERSISTENT_WGS);
    if (variant == 1 || variant == 4 || variant == 6 || variant == 7 || variant == 8 || variant == 9) return hipErrorInvalidValue;     // lab-only forms (see the file header)
    // N: a lane stores 4 consecutive columns; fp32 output may have a ragged N if ldc leaves room for the overhang
    if (p.M <= 0 || p.N <= 0 || p.K <= 0 || (p.K % BK) != 0) return hipErrorInvalidValue;
    if ((p.N % 8) != 0 && !(epilogue == EPI_F32 && p.ldc >= ((p.N + 3) & ~3) && p.bias == nullptr)) return hipErrorInvalidValue;
    if (p.batch > 1 && ((variant != 3 && variant != 4 && variant != 5 && variant != 6 && variant != 7 && variant != 8 && variant != 9 && variant != 10 && variant != 11) || epilogue == EPI_HEADS || epilogue == EPI_F32_RESID)) return hipErrorInvalidValue;
    if ((p.lda % 8) != 0 || (p.ldw % 8) != 0) return hipErrorInvalidValue;
    if (p.split_off != 0 && (epilogue != EPI_BF16 || (p.split_off % 8) != 0)) return hipErrorInvalidValue;   // split-bf16 results: plain bf16 epilogue, staged form
    if ((p.hd > 64 || p.inner_kv > 0 || p.Hkv > 0 || p.gate_act != 0 || (epilogue == EPI_GATED && p.bias != nullptr)) &&
        variant != 3 && variant != 5 && variant != 6 && variant != 7 && variant != 8 && variant != 9 && variant != 10 && variant != 11)
        return hipErrorInvalidValue;   // generalised HEADS / GATED epilogues live in the persistent kernels only
    if (p.rowss_in != nullptr && (p.rowss_parts < 0 || (variant != 3 && variant != 5 && variant != 6 && variant != 7 && variant != 8 && variant != 9 && variant != 10 && variant != 11) || epilogue == EPI_F32_RESID))
        return hipErrorInvalidValue;   // the row scale lives in the persistent kernels' staged epilogue only
    switch (epilogue) {
        case EPI_BF16: return launch_epi<EPI_BF16>(p, variant, stream);
        case EPI_BF16_QGELU: return launch_epi<EPI_BF16_QGELU>(p, variant, stream);
        case EPI_BF16_GELU: return launch_epi<EPI_BF16_GELU>(p, variant, stream);
        case EPI_F32: return launch_epi<EPI_F32>(p, variant, stream);
        case EPI_F32_RESID: return launch_epi<EPI_F32_RESID>(p, variant, stream);
        case EPI_GATED: return launch_epi<EPI_GATED>(p, variant, stream);
        case EPI_HEADS: return launch_epi<EPI_HEADS>(p, variant, stream);
        case EPI_RESID_RMS:
            if (!p.hres || !p.lnw || !p.rowss_out || p.batch > 1 || p.bias != nullptr || (p.ldh % 4) != 0 || (p.ldc % 4) != 0)
                return hipErrorInvalidValue;
            return launch_epi<EPI_RESID_RMS>(p, variant, stream);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace vqs
