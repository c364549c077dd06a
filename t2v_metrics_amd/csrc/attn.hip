// Attention kernels for gfx950 (MI355X).
//
// attn_fwd_kernel  -- K4/K5 of SURVEY.md §8(a-bis): softmax(scale*Q.K^T + bias[h, key-query] + key mask).V
//   * CLIP ViT self-attention  (HF models/clip/modeling_clip.py:259-277: scale = 1/8, fp32 softmax, no mask)
//   * T5 encoder self-attention (HF models/t5/modeling_t5.py:144-173,196-197: scale = 1.0, additive relative
//     position bias shared by all layers, additive key-padding mask)
//   Flash-style: the S x S score matrix is never materialised.  One workgroup = 4 waves x 32 query rows
//   of one (sample, head); K/V are walked in 64-key tiles staged through LDS.
//     - scores are computed TRANSPOSED (mfma(a = K tile, b = Q^T)), so a lane owns ONE query and 16 keys
//       per 32x32 accumulator: the row max/sum are in-lane reductions plus one lane<->lane+32 exchange;
//     - the accumulator register order of S^T is exactly the B-operand order of the P^T.V^T product when
//       the k-slot -> key map of that MFMA is chosen as key = 16t + 4*(lane>>5) + 8*(j>>2) + (j&3):
//       P never moves between lanes (no LDS round trip, no permutes);
//     - V arrives [key][d] (d contiguous); the MFMA wants keys contiguous per lane, so V is transposed
//       while it is staged: each thread packs (V[2k][d], V[2k+1][d]) pairs into conflict-free ds_write_b32;
//       V^T rows are padded to 136 B so the ds_read_b64 fragment reads are conflict-free too;
//     - the K tile uses the same (row>>1)&7 XOR chunk swizzle as the GEMM (conflict-free ds_read_b128);
//     - the T5 bias is a per-head [2S-1] fp32 table in LDS indexed by key-query (never an [H,S,S] tensor);
//     - next tile's global loads are issued before the current tile's MFMAs (register prefetch).
//   Softmax statistics, the running rescale and the output normalisation are fp32; P is rounded to bf16
//   for the PV MFMA.
//
// dec_attn_kernel  -- K6: the T<=16-row decoder attentions (causal self-attention with the unidirectional
//   bucket bias; cross-attention over the encoder keys with the key mask), fp32 VALU math: the work is
//   ~0.1% of the path's FLOPs and latency-bound.
#include "vqs_kernels.h"
#include <stdlib.h>

namespace vqs {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __forceinline__ float a_bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ bf16_t a_f2bf(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}
typedef __attribute__((ext_vector_type(2))) __bf16 a_bf16x2;
__device__ __forceinline__ uint32_t a_pack2(float a, float b) {   // v_cvt_pk_bf16_f32
    a_bf16x2 v;
    v[0] = (__bf16)a;
    v[1] = (__bf16)b;
    return __builtin_bit_cast(uint32_t, v);
}

// fp16 operand forms (AttnParams::f16: the fp16 vision tower): the 32x32x16 MFMA takes bf16 or fp16 fragments in the same layout at the
// same rate; P and the output are packed to the operand type
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 a_f16x2;
__device__ __forceinline__ uint32_t a_pack2h(float a, float b) {   // v_cvt_pk_f16_f32 on gfx950 (RNE, overflow -> inf)
    a_f16x2 v;
    v[0] = (_Float16)a;
    v[1] = (_Float16)b;
    return __builtin_bit_cast(uint32_t, v);
}

// v_max3_f32 without the IEEE-mode operand canonicalisation (v_max_f32 x, x) that fmaxf() costs per input:
// the scores are finite by construction (masking uses NEG_BIG, not -inf)
// max over the two half-waves (lane <-> lane ^ 32) without the LDS round trip of __shfl_xor (ds_bpermute_b32 + a
// lgkmcnt(0) wait in the middle of every tile): gfx950's v_permlane32_swap exchanges lanes 32-63 of one register with
// lanes 0-31 of another, so two copies of x hold {x[l], x[l^32]} afterwards.  Same value, no rounding involved.
__device__ __forceinline__ float a_max_xhalf(float x) {
    typedef unsigned a_v2u __attribute__((ext_vector_type(2)));
    const a_v2u r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    float a = __uint_as_float(r[0]), b = __uint_as_float(r[1]), m;
    asm("v_max_f32 %0, %1, %2" : "=v"(m) : "v"(a), "v"(b));        // finite by construction: no canonicalising v_max x, x
    return m;
}

__device__ __forceinline__ float a_max3(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

static constexpr int KT = 64;            // keys per tile
static constexpr int VT_LD = 136;        // bytes per V^T row (64 keys * 2 B + 8 B pad)
static constexpr int K_LDS = KT * 128;   // 8192
static constexpr int VT_LDS = 64 * VT_LD;  // 8704
static constexpr float NEG_BIG = -1.0e30f;
static constexpr float RESCALE_THR = 6.0f;    // log2 units

__host__ __device__ __forceinline__ int bias_copy_chunks(int S) {
    const int need = (2 * S + 64 + 3 + 3) / 4;              // entries 0 .. 2S+66 are addressable
    return need + ((4 - (need & 15)) & 15);                  // round up to = 4 (mod 16)
}

// Table entry j goes to position j - c of copy c (c = 0..3).  Eight independent global loads per thread are in flight
// before the first LDS store (a load-store-load chain costs a full L2 latency per 256 entries).
__device__ __forceinline__ void fill_bias_copies(float* bias_s, const float* bt, int n, int cs4, int tid, float fac) {
    for (int base = 0; base < cs4 + 3; base += 2048) {
        float bv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int j = base + tid + 256 * i;
            bv[i] = j < n ? bt[j] * fac : 0.0f;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int j = base + tid + 256 * i;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int jj = j - c;
                if (jj >= 0 && jj < cs4) bias_s[c * cs4 + jj] = bv[i];
            }
        }
    }
}


// =====================================================================================================
// attn_fwd_dma_kernel -- same math, same MFMA operand maps and the same softmax code as attn_fwd_kernel; what changes
// is how K and V reach the matrix pipe (tools/attn_lab.sh ablation of the kernel above, T5-XL shape: 1.80 ms total,
// 0.57 ms of it the global->VGPR->ds_write staging with its in-register V transpose, and two barriers per tile):
//   * K and V tiles are staged by LDS-DMA (buffer_load_dwordx4 ... lds), 4 one-KiB pieces per wave per tile, into a
//     two-stage ring: tile kt+1 is in flight while tile kt is computed, ONE barrier per tile, no staging VALU, no
//     ds_write, no staging VGPRs;
//   * K image: [64 keys][128 B] rows, 16-B chunks XOR-swizzled by (key>>1)&7 on the SOURCE address (the DMA writes
//     lane-linear), read back conflict-free with ds_read_b128 exactly as before;
//   * V image: 8 sub-tiles [32 keys][16 d] (32-B rows, one DMA piece each).  The PV MFMA wants, per lane, 4+4
//     consecutive keys of one d column (k-slot j <-> key 16t + 4*half + 8*(j>>2) + (j&3)): that is two
//     ds_read_b64_tr_b16 (hardware 4x16 transpose inside each 16-lane group; semantics pinned by
//     tools/probes/probe_tr.hip) -- V is never transposed by software.  The 128-B blocks of the odd-d sub-tiles are
//     rotated by one so that the two 16-lane groups served together hit different bank halves.
// =====================================================================================================
typedef int a_v4i __attribute__((ext_vector_type(4)));
typedef short a_v4s __attribute__((ext_vector_type(4)));
#define A_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
__device__ __forceinline__ void a_bglds16(a_v4i rsrc, uint32_t voff, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %1, %2, 0 offen lds\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(rsrc), "s"(lds_dst));
}
__device__ __forceinline__ a_v4i a_make_rsrc(const void* base) {
    const uint64_t b = (uint64_t)base;
    a_v4i r;
    r.x = __builtin_amdgcn_readfirstlane((uint32_t)b);
    r.y = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32) & 0xffffu);
    r.z = (int)0xffffffffu;
    r.w = 0x00020000;
    return r;
}
static constexpr int ST_BYTES = 16384;       // one stage: K 8 KiB + V 8 KiB

// VQS_ATTN_TIMING (lab builds only: make variant NAME=attn_timing VFLAGS=-DVQS_ATTN_TIMING=1): per-phase s_memtime cycle totals of
// the LDS-DMA kernel, summed over all waves into a device buffer registered with vqs_lab_set_attn_timing (8 x int64:
// [0] wait for the staged tile + barrier, [1] K reads + QK^T MFMAs (+ bias), [2] mask / max / rescale branch, [3] exp + pack + V
// reads + PV MFMAs, [4] prologue (bias table fill, Q load), [5] epilogue, [6] wave-tiles, [7] waves).  Each probe is an
// s_memtime + s_waitcnt lgkmcnt(0): it perturbs what it measures (the waits drain the LDS reads in flight), so read the split,
// not the total.
#ifndef VQS_ATTN_TIMING
#define VQS_ATTN_TIMING 0
#endif
#if VQS_ATTN_TIMING
__device__ unsigned long long* g_attn_timing = nullptr;
__device__ __forceinline__ unsigned long long a_now() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
#define A_T(var) const unsigned long long var = a_now()
#define A_ACC(slot, t1, t0) t_acc[slot] += (t1) - (t0)
#else
#define A_T(var)
#define A_ACC(slot, t1, t0)
#endif

// BIAS_ACC (compile-time form of the biased kernel): 1 = the position bias enters through the MFMA accumulators
#ifndef VQS_ATTN_BIAS_ACC
#define VQS_ATTN_BIAS_ACC 0
#endif
#define ATTN_DMA_KERNEL attn_fwd_dma_kernel
#define ATTN_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0)
#define ATTN_PACK2(x, y) a_pack2(x, y)
#define ATTN_ONES 0x3f803f80u
#include "attn_dma_kernel.inc"
#undef ATTN_DMA_KERNEL
#undef ATTN_MFMA
#undef ATTN_PACK2
#undef ATTN_ONES
// the same kernel on IEEE fp16 q / k / v with an fp16 output (AttnParams::f16: the fp16 vision tower)
#define ATTN_DMA_KERNEL attn_fwd_dma_f16_kernel
#define ATTN_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0)
#define ATTN_PACK2(x, y) a_pack2h(x, y)
#define ATTN_ONES 0x3c003c00u
#include "attn_dma_kernel.inc"
#undef ATTN_DMA_KERNEL
#undef ATTN_MFMA
#undef ATTN_PACK2
#undef ATTN_ONES

#if VQS_ATTN_TIMING
hipError_t lab_set_attn_timing(unsigned long long* d_buf) { return hipMemcpyToSymbol(HIP_SYMBOL(g_attn_timing), &d_buf, sizeof(d_buf)); }
#endif


// =====================================================================================================
// attn_fwd_hd_kernel<HD, CAUSAL> -- the LDS-DMA flash kernel above generalised for the Qwen2.5-VL row:
// head_dim HD (128; the tower's 80-wide heads are zero-padded to 128 by the packer), grouped-query attention (query
// head h reads key/value head h / (H / Hkv)), optional causal mask (HF Qwen2_5_VLAttention is_causal, key <= query),
// no position bias (RoPE is applied to Q/K beforehand).  Same MFMA operand maps and softmax; per tile 2*HD/16 QK^T
// MFMAs and (HD/32 + 1) * 4 PV MFMAs per wave; K rows are HD*2 bytes with the 16-B chunk index XOR-ed with row & 15
// (256-B rows span all 64 banks: 16 rows of a ds_read_b128 lane group land on 16 different slots); V is 2 * HD/16
// sub-tiles [32 keys][16 d], read with ds_read_b64_tr_b16.
// q [B, H, S, HD]; k, v [B, Hkv, S, HD]; out [B*S, H*HD].
// =====================================================================================================
// F16 (round 6): q / k / v / out are IEEE fp16 tensors (the Qwen2.5-VL row's range-safe fp16 forms; power-of-two scales on q, k fold into
// p.scale on the host, a scale on v passes through to the output).
template <int HD, bool CAUSAL, bool F16 = false>
__global__ void __launch_bounds__(256) attn_fwd_hd_kernel(const AttnParams p) {
#define HD_MFMA(a, b, c) (F16 ? __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0) \
                              : __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0))
#define HD_PACK2(a, b) (F16 ? a_pack2h(a, b) : a_pack2(a, b))
    constexpr int KS = HD / 16;                 // k-steps of the QK^T product
    constexpr int DF = HD / 32;                 // 32-wide output blocks
    constexpr int ROWB = HD * 2;                // bytes per K row
    constexpr int KBYTES = KT * ROWB;           // one K (or V) tile
    constexpr int STB = 2 * KBYTES;             // one stage
    constexpr int RPP = 1024 / ROWB;            // K rows per 1-KiB DMA piece
    constexpr int KPW = KT / RPP / 4;           // K pieces per wave per tile
    constexpr int NDB = HD / 16;                // 16-wide d blocks
    constexpr int VPW = 2 * NDB / 4;            // V sub-tiles per wave per tile
    static_assert(HD == 128, "row chunk swizzle below is written for 256-byte rows");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)A_LDS_PTR(smem));

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hh = lane >> 5;
    const int S = p.S;
    const int nqb = (S + 127) >> 7;
    const int slot = blockIdx.x >> 3;
    const int qb = slot % nqb;
    const int g = (slot / nqb) * 8 + (blockIdx.x & 7);
    if (g >= p.B * p.H) return;
    const int b = g / p.H, h = g - b * p.H;
    const int Hkv = p.Hkv > 0 ? p.Hkv : p.H;
    const int hk = h / (p.H / Hkv);
    const bf16_t* Q = p.q + ((size_t)b * p.H + h) * S * HD;
    const bf16_t* K = p.k + ((size_t)b * Hkv + hk) * S * HD;
    const bf16_t* V = p.v + ((size_t)b * Hkv + hk) * S * HD;
    const int klen = p.key_len ? min(p.key_len[b], S) : S;
    int ntiles = (klen + KT - 1) / KT;
    if (CAUSAL) ntiles = min(ntiles, (min(qb * 128 + 127, S - 1) >> 6) + 1);   // keys beyond the block's last query

    constexpr float LOG2E = 1.4426950408889634f;
    const float sl2 = p.scale * LOG2E;
    const int qrow = qb * 128 + wv * 32 + (lane & 31);
    const int qrow_c = min(qrow, S - 1);
    uint4 qf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
        qf[ks] = *reinterpret_cast<const uint4*>(Q + (size_t)qrow_c * HD + 16 * ks + 8 * hh);

    // ---- staging.  K piece q (RPP rows each): wave wv issues pieces wv + 4j; lane = (row in piece, 16-B position).
    // V sub-tile (kh, db): wave wv issues db = wv + 4*(j>>1), kh = j&1.
    const a_v4i rsK = a_make_rsrc(K), rsV = a_make_rsrc(V);
    constexpr int CPR = ROWB / 16;              // 16-B chunks per K row
    int k_row[KPW], v_row[VPW];
    uint32_t k_col[KPW], v_col[VPW], v_dst[VPW];
#pragma unroll
    for (int j = 0; j < KPW; ++j) {
        const int r = RPP * (wv + 4 * j) + lane / CPR;
        k_row[j] = r;
        k_col[j] = (uint32_t)(((lane % CPR) ^ (r & 15)) << 4);
    }
#pragma unroll
    for (int j = 0; j < VPW; ++j) {
        const int db = wv + 4 * (j >> 1), kh = j & 1;
        v_row[j] = 32 * kh + 4 * (((lane >> 3) - (db & 1)) & 7) + ((lane >> 1) & 3);
        v_col[j] = (uint32_t)((16 * db + 8 * (lane & 1)) * 2);
        v_dst[j] = (uint32_t)(KBYTES + (kh * NDB + db) * 1024);
    }
    auto stage = [&](int kt, int st) {
        const int kb = kt * KT;
        const uint32_t sb = lds_base + st * STB;
#pragma unroll
        for (int j = 0; j < KPW; ++j)
            a_bglds16(rsK, (uint32_t)min(kb + k_row[j], S - 1) * (uint32_t)ROWB + k_col[j], sb + (wv + 4 * j) * 1024);
#pragma unroll
        for (int j = 0; j < VPW; ++j)
            a_bglds16(rsV, (uint32_t)min(kb + v_row[j], S - 1) * (uint32_t)ROWB + v_col[j], sb + v_dst[j]);
    };

    f32x16 o[DF];
#pragma unroll
    for (int df = 0; df < DF; ++df)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[df][r] = 0.0f;
    f32x16 osum;
#pragma unroll
    for (int r = 0; r < 16; ++r) osum[r] = 0.0f;
    float m_run = NEG_BIG;
    uint4 ones;
    ones.x = ones.y = ones.z = ones.w = F16 ? 0x3c003c00u : 0x3f803f80u;

    const int swr = lane & 15;                                  // row & 15 of this lane's K rows (kf*32 + lane&31)
    const int k_rd = (lane & 31) * ROWB;
    const int dbl = (lane >> 4) & 1, tq = lane & 15;
    int v_rd[4];
#pragma unroll
    for (int xi = 0; xi < 4; ++xi)
        v_rd[xi] = KBYTES + dbl * 1024 + (((2 * xi + hh + dbl) & 7) << 7) + (tq >> 2) * 32 + (tq & 3) * 8;

#pragma unroll
    for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(qf[ks].x), "+v"(qf[ks].y), "+v"(qf[ks].z), "+v"(qf[ks].w));
    if (ntiles > 0) stage(0, 0);
    for (int kt = 0; kt < ntiles; ++kt) {
        const int st = kt & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < ntiles) stage(kt + 1, st ^ 1);
        const char* k_lds = smem + st * STB;

        // ---- S^T = K . Q^T, four k-steps of fragments in flight at a time
        f32x16 s[2];
#pragma unroll
        for (int kf = 0; kf < 2; ++kf)
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kf][r] = 0.0f;
#pragma unroll
        for (int k4 = 0; k4 < KS; k4 += 4) {
            uint4 kfr[4][2];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int kf = 0; kf < 2; ++kf)
                    kfr[ks][kf] = *reinterpret_cast<const uint4*>(k_lds + kf * 32 * ROWB + k_rd + (((2 * (k4 + ks) + hh) ^ swr) << 4));
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int kf = 0; kf < 2; ++kf)
                    s[kf] = HD_MFMA(kfr[ks][kf], qf[k4 + ks], s[kf]);
        }

        // ---- masks (ragged last tile, causal diagonal), running max with deferred rescale, p = 2^(s*c - m)
        const int kb = kt * KT;
        if (kb + KT > klen || (CAUSAL && kb + KT - 1 > qb * 128 + wv * 32)) {      // wave-uniform
#pragma unroll
            for (int kf = 0; kf < 2; ++kf)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kb + kf * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                    if (key >= klen || (CAUSAL && key > qrow)) s[kf][r] = NEG_BIG;
                }
        }
#define VQS_SV(i) s[(i) >> 4][(i) & 15]
        float mx = a_max3(VQS_SV(0), VQS_SV(1), VQS_SV(2));
#pragma unroll
        for (int i = 3; i < 31; i += 2) mx = a_max3(mx, VQS_SV(i), VQS_SV(i + 1));
        mx = fmaxf(mx, VQS_SV(31));
#undef VQS_SV
        mx = a_max_xhalf(mx);
        mx *= sl2;
        if (__any(mx > m_run + RESCALE_THR)) {
            const float m_new = fmaxf(m_run, mx);
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            m_run = m_new;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
#pragma unroll
                for (int df = 0; df < DF; ++df) o[df][r] *= alpha;
                osum[r] *= alpha;
            }
        }
        const float neg_m = -m_run;
#pragma unroll
        for (int kf = 0; kf < 2; ++kf)
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kf][r] = __builtin_amdgcn_exp2f(fmaf(s[kf][r], sl2, neg_m));

        // ---- O^T += V^T . P^T
#pragma unroll
        for (int kf = 0; kf < 2; ++kf)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                uint4 pb;
                pb.x = HD_PACK2(s[kf][8 * t + 0], s[kf][8 * t + 1]);
                pb.y = HD_PACK2(s[kf][8 * t + 2], s[kf][8 * t + 3]);
                pb.z = HD_PACK2(s[kf][8 * t + 4], s[kf][8 * t + 5]);
                pb.w = HD_PACK2(s[kf][8 * t + 6], s[kf][8 * t + 7]);
                osum = HD_MFMA(ones, pb, osum);
#pragma unroll
                for (int df = 0; df < DF; ++df) {
                    const char* vp = k_lds + (kf * NDB + 2 * df) * 1024;
                    const a_v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) a_v4s*)A_LDS_PTR(vp + v_rd[2 * t + 0]));
                    const a_v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) a_v4s*)A_LDS_PTR(vp + v_rd[2 * t + 1]));
                    const uint2 lo2 = __builtin_bit_cast(uint2, lo), hi2 = __builtin_bit_cast(uint2, hi);
                    uint4 va;
                    va.x = lo2.x; va.y = lo2.y; va.z = hi2.x; va.w = hi2.y;
                    o[df] = HD_MFMA(va, pb, o[df]);
                }
            }
    }

    const float l_tot = osum[0];
    const float inv = l_tot > 0.0f ? 1.0f / l_tot : 0.0f;
    if (qrow < S) {
        const int ohd = p.out_hd > 0 ? p.out_hd : HD;        // lanes of the head that leave (a multiple of 4)
        bf16_t* orow = p.out + ((size_t)b * S + qrow) * ((size_t)p.H * ohd) + h * ohd;
#pragma unroll
        for (int df = 0; df < DF; ++df)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                uint2 v;
                v.x = HD_PACK2(o[df][4 * gq + 0] * inv, o[df][4 * gq + 1] * inv);
                v.y = HD_PACK2(o[df][4 * gq + 2] * inv, o[df][4 * gq + 3] * inv);
                if (df * 32 + 8 * gq + 4 * hh < ohd) *reinterpret_cast<uint2*>(orow + df * 32 + 8 * gq + 4 * hh) = v;
            }
    }
#undef HD_MFMA
#undef HD_PACK2
}

static constexpr int attn_variant() { return 1; }        // the shipped library holds the LDS-DMA kernel only
static constexpr size_t attn_lds_pad() { return 0; }

template <typename KernelT>
static hipError_t launch_attn_t(KernelT kern, const AttnParams& p, size_t lds, hipStream_t stream) {
    lds += attn_lds_pad();
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    if (lds > 65536) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    const int nqb = (p.S + 127) / 128, G = p.B * p.H;
    dim3 grid(((G + 7) / 8) * 8 * nqb), block(256);
    hipLaunchKernelGGL(kern, grid, block, lds, stream, p);
    return hipGetLastError();
}

hipError_t launch_attention(const AttnParams& p, hipStream_t stream) {
    if (p.B <= 0 || p.H <= 0 || p.S <= 0) return hipErrorInvalidValue;
    if (p.hd == 128) {
        const int Hkv = p.Hkv > 0 ? p.Hkv : p.H;
        if (p.bias_table != nullptr || Hkv <= 0 || (p.H % Hkv) != 0) return hipErrorInvalidValue;
        if (p.out_hd < 0 || p.out_hd > 128 || (p.out_hd & 3)) return hipErrorInvalidValue;
        const size_t lds = 2 * 2 * (size_t)KT * 128 * 2;            // two stages of K + V tiles
        if (p.f16)
            return p.causal ? launch_attn_t(attn_fwd_hd_kernel<128, true, true>, p, lds, stream)
                            : launch_attn_t(attn_fwd_hd_kernel<128, false, true>, p, lds, stream);
        return p.causal ? launch_attn_t(attn_fwd_hd_kernel<128, true>, p, lds, stream)
                        : launch_attn_t(attn_fwd_hd_kernel<128, false>, p, lds, stream);
    }
    if ((p.hd != 0 && p.hd != 64) || p.out_hd != 0) return hipErrorInvalidValue;
    if (p.causal || (p.Hkv > 0 && p.Hkv != p.H)) return hipErrorInvalidValue;
    const size_t bias_bytes = p.bias_table ? (size_t)bias_copy_chunks(p.S) * 64 : 0;
    const size_t lds = 2 * ST_BYTES + bias_bytes;
    if (p.f16)      // fp16 q / k / v / out: the vision tower (no bias) and the T5 encoder of option enc_fp16 (position bias + key mask)
        return p.bias_table ? launch_attn_t(attn_fwd_dma_f16_kernel<true>, p, lds, stream)
                            : launch_attn_t(attn_fwd_dma_f16_kernel<false>, p, lds, stream);
    return p.bias_table ? launch_attn_t(attn_fwd_dma_kernel<true>, p, lds, stream)
                        : launch_attn_t(attn_fwd_dma_kernel<false>, p, lds, stream);
}

// Dynamic LDS request of the self-attention kernel launch_attention() picks for (S, bias, hd): what the occupancy of the
// kernel hangs on (gfx950 allocates LDS in 1 280-B granules out of 160 KiB per CU; T5-XL, S = 608 with bias: 53 504 B ->
// three workgroups per CU with 1.1 KiB to spare each -- tests/test_build_invariants.py guards it).
size_t attention_lds_bytes(int S, bool has_bias, int hd) {
    if (hd == 128) return 2 * 2 * (size_t)KT * 128 * 2;
    const size_t bias_bytes = has_bias ? (size_t)bias_copy_chunks(S) * 64 : 0;
    return (attn_variant() == 0 ? (size_t)(K_LDS + VT_LDS) : (size_t)(2 * ST_BYTES)) + bias_bytes;
}

// =====================================================================================================
// Decoder attention: T query rows per (sample, head); fp32 VALU.
// =====================================================================================================
static constexpr int DEC_TMAX = 16;

__device__ __forceinline__ float block_reduce(float v, bool is_max, float* red, int tid) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float o = __shfl_xor(v, off);
        v = is_max ? fmaxf(v, o) : v + o;
    }
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    float r = red[0];
#pragma unroll
    for (int i = 1; i < 4; ++i) r = is_max ? fmaxf(r, red[i]) : r + red[i];
    return r;
}

__global__ void __launch_bounds__(256) dec_attn_kernel(const DecAttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int T = p.T, S = p.S;
    float* q_s = reinterpret_cast<float*>(smem);       // [T][64]
    float* sc = q_s + T * 64;                          // [T][S]
    float* part = sc + (size_t)T * S;                  // [4][T][64]
    float* red = part + 4 * T * 64;                    // [4]
    float* rsum = red + 4;                             // [T]

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int h = blockIdx.x, b = blockIdx.y;
    const int klen = p.cross ? (p.key_len ? min(p.key_len[b], S) : S) : S;
    const size_t kvb = p.kv_stride_b ? (size_t)p.kv_stride_b : (size_t)T * p.ldk;     // self form: sample stride of the K/V rows
    const int bias_ld = p.bias_ld ? p.bias_ld : T;

    for (int i = tid; i < T * 64; i += 256) {
        const int t = i >> 6, d = i & 63;
        q_s[i] = a_bf2f(p.q[((size_t)b * T + t) * p.ldq + h * 64 + d]);
    }
    __syncthreads();

    // ---- scores
    for (int j = tid; j < S; j += 256) {
        const bf16_t* krow = p.cross ? p.k + (((size_t)b * p.H + h) * S + j) * 64
                                     : p.k + (size_t)b * kvb + (size_t)j * p.ldk + h * 64;
        float kv[64];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const uint4 u = *reinterpret_cast<const uint4*>(krow + c * 8);
            const uint32_t w4[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                kv[c * 8 + 2 * e] = __uint_as_float(w4[e] << 16);
                kv[c * 8 + 2 * e + 1] = __uint_as_float(w4[e] & 0xffff0000u);
            }
        }
        for (int t = 0; t < T; ++t) {
            float a = 0.0f;
#pragma unroll
            for (int d = 0; d < 64; ++d) a = fmaf(q_s[t * 64 + d], kv[d], a);
            bool masked;
            if (p.cross) {
                masked = j >= klen;
            } else {
                masked = j > t + p.qpos0;                          // causal
                if (!masked && p.bias_table) a += p.bias_table[h * bias_ld + (t + p.qpos0 - j)];
            }
            sc[(size_t)t * S + j] = masked ? NEG_BIG : a;
        }
    }
    __syncthreads();

    // ---- softmax statistics per query row
    for (int t = 0; t < T; ++t) {
        float mx = NEG_BIG;
        for (int j = tid; j < S; j += 256) mx = fmaxf(mx, sc[(size_t)t * S + j]);
        mx = block_reduce(mx, true, red, tid);
        float sm = 0.0f;
        for (int j = tid; j < S; j += 256) {
            const float e = __expf(sc[(size_t)t * S + j] - mx);
            sc[(size_t)t * S + j] = e;
            sm += e;
        }
        sm = block_reduce(sm, false, red, tid);
        if (tid == 0) rsum[t] = sm;
    }
    __syncthreads();

    // ---- O[t][d] = sum_j p[t][j] V[j][d]; wave wv takes keys j = wv, wv+4, ...; lane = d
    float acc[DEC_TMAX];
#pragma unroll
    for (int t = 0; t < DEC_TMAX; ++t) acc[t] = 0.0f;
#pragma unroll 8
    for (int j = wv; j < S; j += 4) {
        const bf16_t* vrow = p.cross ? p.v + (((size_t)b * p.H + h) * S + j) * 64
                                     : p.v + (size_t)b * kvb + (size_t)j * p.ldk + h * 64;
        const float vv = a_bf2f(vrow[lane]);
#pragma unroll
        for (int t = 0; t < DEC_TMAX; ++t)
            if (t < T) acc[t] = fmaf(sc[(size_t)t * S + j], vv, acc[t]);
    }
#pragma unroll
    for (int t = 0; t < DEC_TMAX; ++t)
        if (t < T) part[(wv * T + t) * 64 + lane] = acc[t];
    __syncthreads();
    for (int i = tid; i < T * 64; i += 256) {
        const int t = i >> 6, d = i & 63;
        const float v = part[(0 * T + t) * 64 + d] + part[(1 * T + t) * 64 + d] + part[(2 * T + t) * 64 + d] +
                        part[(3 * T + t) * 64 + d];
        p.out[((size_t)b * T + t) * ((size_t)p.H * 64) + h * 64 + d] = a_f2bf(v / rsum[t]);
    }
}

// Precise self form (teacher-forced rows, no cache): q | k | v are FP32 columns of one [B*T, ldq] tensor (the summed partials of the
// stacked hi / lo qkv GEMM -- never rounded), scores / softmax / P.V in fp32 as above, the output leaves as a split-bf16 tensor
// (hi plane at out, lo plane at out + out_plane).  One workgroup per (head, sample), one wave: T <= 16 rows of 64 lanes.
__global__ void __launch_bounds__(64) dec_self_attn_precise_kernel(const DecAttnParams p) {
    __shared__ float q_s[DEC_TMAX][64], k_s[DEC_TMAX][64], v_s[DEC_TMAX][64], pr[DEC_TMAX][DEC_TMAX], rsum[DEC_TMAX];
    const int T = p.T, lane = threadIdx.x, h = blockIdx.x, b = blockIdx.y;
    const float* q = reinterpret_cast<const float*>(p.q);
    const float* k = reinterpret_cast<const float*>(p.k);
    const float* v = reinterpret_cast<const float*>(p.v);
    for (int t = 0; t < T; ++t) {
        const size_t row = ((size_t)b * T + t) * p.ldq + h * 64 + lane;
        q_s[t][lane] = q[row];
        k_s[t][lane] = k[row];
        v_s[t][lane] = v[row];
    }
    __syncthreads();
    // scores: lane -> (query t = lane / 16, key j = lane % 16) in passes of 4 queries; the dot product in d-ascending fmaf order
    for (int t0 = 0; t0 < T; t0 += 4) {
        const int t = t0 + (lane >> 4), j = lane & 15;
        if (t < T && j < T) {
            float a = 0.0f;
#pragma unroll
            for (int d = 0; d < 64; ++d) a = fmaf(q_s[t][d], k_s[j][d], a);
            const bool masked = j > t;
            if (!masked && p.bias_table) a += p.bias_table[h * T + (t - j)];
            pr[t][j] = masked ? NEG_BIG : a;
        }
    }
    __syncthreads();
    if (lane < T) {                                   // one lane per query row: T <= 16 keys
        const int t = lane;
        float mx = NEG_BIG;
        for (int j = 0; j < T; ++j) mx = fmaxf(mx, pr[t][j]);
        float sm = 0.0f;
        for (int j = 0; j < T; ++j) {
            const float e = __expf(pr[t][j] - mx);
            pr[t][j] = e;
            sm += e;
        }
        rsum[t] = sm;
    }
    __syncthreads();
    for (int t = 0; t < T; ++t) {
        float o = 0.0f;
        for (int j = 0; j <= t; ++j) o = fmaf(pr[t][j], v_s[j][lane], o);     // masked keys hold exp(NEG_BIG - mx) = 0
        o = o / rsum[t];                                                       // one division at the end, as dec_attn_kernel
        const bf16_t hi = a_f2bf(o);
        const size_t dst = ((size_t)b * T + t) * ((size_t)p.H * 64) + h * 64 + lane;
        p.out[dst] = hi;
        p.out[dst + p.out_plane] = a_f2bf(o - a_bf2f(hi));
    }
}

hipError_t launch_decoder_attention(const DecAttnParams& p, hipStream_t stream) {
    if (p.T <= 0 || p.T > DEC_TMAX || p.S <= 0) return hipErrorInvalidValue;
    if (p.precise) {
        if (p.cross || p.S != p.T || p.kv_stride_b != 0 || p.qpos0 != 0 || p.ldq != p.ldk || (p.bias_ld != 0 && p.bias_ld != p.T) || p.out_plane <= 0)
            return hipErrorInvalidValue;              // teacher-forced self form only
        hipLaunchKernelGGL(dec_self_attn_precise_kernel, dim3(p.H, p.B), dim3(64), 0, stream, p);
        return hipGetLastError();
    }
    size_t lds = ((size_t)p.T * 64 + (size_t)p.T * p.S + 4 * (size_t)p.T * 64 + 4 + p.T) * sizeof(float);
    lds = (lds + 15) & ~(size_t)15;
    if (lds > 65536) return hipErrorInvalidValue;
    dim3 grid(p.H, p.B), block(256);
    hipLaunchKernelGGL(dec_attn_kernel, grid, block, lds, stream, p);
    return hipGetLastError();
}

}  // namespace vqs
