// Greedy-decoding step of the Qwen2.5-VL language model against a KV cache (include/vqs_qwen.h vqs_qwen_decode): the two kernels
// a one-token step needs besides the GEMMs and norms of the prefill.  Replaces, for max_new_tokens > 1 and free-form generation
// (/root/reference/t2v_metrics/models/vqascore_models/qwen2vl_model.py:222-230, :495-563), HF generate's cached forward
// (Qwen2_5_VLAttention.forward with past_key_values, modeling_qwen2_5_vl.py:653-717): K after RoPE and V are appended at the
// sample's current length and the new query attends over [0, length].
// Both kernels are HBM-bound row streams (a step reads every cached K / V row of the layer once): 16-byte accesses per lane, fp32
// arithmetic, no matrix pipe -- one query row per (sample, head) is 1/128 of an MFMA tile.
#include "vqs_kernels.h"

#include <algorithm>

namespace vqs {

namespace {
__device__ __forceinline__ float d_bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
typedef __attribute__((ext_vector_type(2))) __bf16 d_bf16x2;
__device__ __forceinline__ bf16_t d_f2bf(float f) {           // v_cvt_pk_bf16_f32, RNE
    d_bf16x2 v;
    v[0] = (__bf16)f;
    v[1] = (__bf16)0.0f;
    return (bf16_t)(__builtin_bit_cast(uint32_t, v) & 0xffff);
}
}  // namespace

// qkv [B, (Hq + 2 Hkv) * hd] = bf16(x W^T + b) of the new token, heads in hd-lane slots (q heads, then k heads, then v heads).
// Block (head slot, sample), 64 threads: rotates q and k by the new position's table (x[i] <- x[i] c[i] - x[i+half] s[i],
// x[i+half] <- x[i+half] c[i] + x[i] s[i], i < half; lanes >= 2 half are the heads' zero padding), writes q dense [B, Hq * hd] and
// k / v into the cache rows [b, head, len[b], :] of this layer.
__global__ void __launch_bounds__(64) qwen_decode_rope_append_kernel(const bf16_t* __restrict__ qkv, const float* __restrict__ cs,
                                                                     const float* __restrict__ sn, const int* __restrict__ len,
                                                                     bf16_t* __restrict__ q_out, bf16_t* __restrict__ kc,
                                                                     bf16_t* __restrict__ vc, int Hq, int Hkv, int hd, int half, int Lmax) {
    const int slot = blockIdx.x, b = blockIdx.y, t = threadIdx.x;
    const int QN = (Hq + 2 * Hkv) * hd;
    const bf16_t* src = qkv + (size_t)b * QN + (size_t)slot * hd;
    const int pos = len[b];
    // a stale or over-advanced d_len must not corrupt a neighbouring cache slab: positions outside [0, Lmax) write nothing
    // (block-uniform early-out; the C-API caller owns d_len, only the Python wrapper checks it on the host)
    if ((unsigned)pos >= (unsigned)Lmax) return;
    bf16_t* dst;
    bool rot = true;
    if (slot < Hq) {
        dst = q_out + ((size_t)b * Hq + slot) * hd;
    } else if (slot < Hq + Hkv) {
        dst = kc + (((size_t)b * Hkv + (slot - Hq)) * Lmax + pos) * hd;
    } else {
        dst = vc + (((size_t)b * Hkv + (slot - Hq - Hkv)) * Lmax + pos) * hd;
        rot = false;
    }
    if (rot) {
        if (t < half) {
            const float a = d_bf2f(src[t]), c2 = d_bf2f(src[t + half]);
            const float c = cs[(size_t)b * half + t], s = sn[(size_t)b * half + t];
            dst[t] = d_f2bf(a * c - c2 * s);
            dst[t + half] = d_f2bf(c2 * c + a * s);
        }
        for (int d = 2 * half + t; d < hd; d += 64) dst[d] = src[d];
    } else {
        for (int d = t; d < hd; d += 64) dst[d] = src[d];
    }
}

hipError_t launch_qwen_decode_rope_append(const bf16_t* qkv, const float* cs, const float* sn, const int* len, bf16_t* q_out, bf16_t* kc,
                                          bf16_t* vc, int B, int Hq, int Hkv, int hd, int half, int Lmax, hipStream_t s) {
    if (B <= 0 || Hq <= 0 || Hkv <= 0 || half <= 0 || half > 64 || 2 * half > hd || B > 65535) return hipErrorInvalidValue;
    hipLaunchKernelGGL(qwen_decode_rope_append_kernel, dim3((unsigned)(Hq + 2 * Hkv), (unsigned)B), dim3(64), 0, s, qkv, cs, sn, len, q_out,
                       kc, vc, Hq, Hkv, hd, half, Lmax);
    return hipGetLastError();
}

// One query row per (sample, query head) over the cached keys [0, len[b]] of its key/value head (h / (Hq / Hkv)); hd = 128.
// Block (head, sample), 256 threads.  Scores: thread j takes keys j, j + 256, ...: fp32 dot of the 256-byte K row with q (LDS,
// fp32) -> t_j = s_j * scale in LDS.  m = max t; p_j = exp(t_j - m), fp32, NOT rounded (a one-row softmax has no matrix-pipe operand
// to round for); l = sum p.  Output: wave w takes keys w, w + 4, ...; lane i the lanes 2i, 2i + 1 (one coalesced 256-byte V row per
// wave and key), the four partial rows are added in wave order, divided by l, rounded to bf16.  Dynamic LDS: (len + 1) floats.
// PRECISE (the precise tail of vqs_qwen_score, vqs_qwen.cpp): q is an fp32 row (already rotated), `len` holds the COUNT of valid keys
// (the prompt's own length: the row attends to itself through the K / V rows the prefill wrote), and the output leaves as a split-bf16
// tensor: hi plane at out, lo = bf16(o - hi) at out + out_plane.
// KVH (round 6, PRECISE only): the K / V rows are IEEE fp16 (the prefill's range-safe fp16 forms, held behind power-of-two scales: the key
// scale is folded into `scale` by the host, the output is multiplied by `vscale` = 1 / the value scale).
typedef __attribute__((ext_vector_type(2))) _Float16 d_f16x2;
__device__ __forceinline__ void d_cvt2(uint32_t w, bool f16, float& a, float& b) {
    if (f16) {
        const d_f16x2 h = __builtin_bit_cast(d_f16x2, w);
        a = (float)h[0]; b = (float)h[1];
    } else {
        a = __uint_as_float(w << 16); b = __uint_as_float(w & 0xffff0000u);
    }
}
template <bool PRECISE, bool KVH = false>
__global__ void __launch_bounds__(256) qwen_decode_attn_kernel(const void* __restrict__ q_, const bf16_t* __restrict__ kc,
                                                               const bf16_t* __restrict__ vc, const int* __restrict__ len,
                                                               bf16_t* __restrict__ out, int Hq, int Hkv, int Lmax, float scale,
                                                               long long out_plane, float vscale) {
    constexpr int HD = 128;
    extern __shared__ __attribute__((aligned(16))) float d_smem[];
    __shared__ float qs[HD];
    __shared__ float red[4];
    __shared__ float part[4][HD];
    const int h = blockIdx.x, b = blockIdx.y, t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int hk = h / (Hq / Hkv);
    // keys [0, len[b]]; a length outside the cache is clamped to it (the LDS score row holds Lmax floats)
    const int n = PRECISE ? min(max(len[b], 1), Lmax) : min(max(len[b], 0), Lmax - 1) + 1;
    const bf16_t* K = kc + ((size_t)b * Hkv + hk) * Lmax * HD;
    const bf16_t* V = vc + ((size_t)b * Hkv + hk) * Lmax * HD;
    if (t < HD) {
        if (PRECISE) qs[t] = reinterpret_cast<const float*>(q_)[((size_t)b * Hq + h) * HD + t];
        else qs[t] = d_bf2f(reinterpret_cast<const bf16_t*>(q_)[((size_t)b * Hq + h) * HD + t]);
    }
    __syncthreads();
    float mx = -3.0e38f;
    for (int j = t; j < n; j += 256) {
        const uint4* kr = reinterpret_cast<const uint4*>(K + (size_t)j * HD);
        float acc = 0.0f;
#pragma unroll
        for (int c = 0; c < HD / 8; ++c) {
            const uint4 u = kr[c];
            const uint32_t w4[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float k0, k1;
                d_cvt2(w4[e], KVH, k0, k1);
                acc = fmaf(k0, qs[8 * c + 2 * e], acc);
                acc = fmaf(k1, qs[8 * c + 2 * e + 1], acc);
            }
        }
        const float tj = acc * scale;
        d_smem[j] = tj;
        mx = fmaxf(mx, tj);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    if (lane == 0) red[wv] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float ls = 0.0f;
    for (int j = t; j < n; j += 256) {
        const float p = __expf(d_smem[j] - mx);
        d_smem[j] = p;
        ls += p;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ls += __shfl_xor(ls, off);
    if (lane == 0) red[wv] = ls;
    __syncthreads();
    const float l = (red[0] + red[1]) + (red[2] + red[3]);
    float o0 = 0.0f, o1 = 0.0f;
    for (int j = wv; j < n; j += 4) {
        const uint32_t u = reinterpret_cast<const uint32_t*>(V + (size_t)j * HD)[lane];
        const float p = d_smem[j];
        float v0, v1;
        d_cvt2(u, KVH, v0, v1);
        o0 = fmaf(p, v0, o0);
        o1 = fmaf(p, v1, o1);
    }
    part[wv][2 * lane] = o0;
    part[wv][2 * lane + 1] = o1;
    __syncthreads();
    if (t < HD) {
        float o = ((part[0][t] + part[1][t]) + (part[2][t] + part[3][t])) / l;
        if (KVH) o *= vscale;
        const bf16_t hi = d_f2bf(o);
        out[((size_t)b * Hq + h) * HD + t] = hi;
        if (PRECISE) out[out_plane + ((size_t)b * Hq + h) * HD + t] = d_f2bf(o - d_bf2f(hi));
    }
}

// Grouped-query form of the same step: ONE block per (key/value head, sample) serves all G = Hq / Hkv query heads that share the head,
// so a cached K / V row is read once (the per-query-head kernel above re-reads it G = 7 times and walks it one thread per row: 64 lanes
// 256 B apart per load instruction -- 385 GB/s at B = 64, L = 808: 7.7 of the step's 13 ms of kernel time, profiles/r4_call12_*).
// Lane map for both passes: a wave instruction covers 4 consecutive keys x 16 chunks of 16 B (8 dims) = four whole 256-byte rows;
// wave w takes the key groups w, w + 4, ...; eight loads are in flight per wave before the first use.  Scores: per head an 8-term dot
// per lane, summed over the row's 16 lanes by xor-shuffles (1, 2, 4, 8), scaled, kept in LDS [G][Lmax]; softmax statistics per head in
// fp32, probabilities NOT rounded (as above); output: per lane 8 dims x G heads of fp32 accumulators, the 4 key sub-rows folded by
// xor-shuffles (16, 32), the four waves' partial rows added in wave order, one division, bf16.  Dynamic LDS: G * Lmax floats.
static constexpr int QD_GMAX = 8;
static constexpr int QD_NW = 8;            // waves per block: eight split the key range, two per SIMD cover each other's load latency
template <bool PRECISE, bool KVH = false>
__global__ void __launch_bounds__(64 * QD_NW) qwen_decode_attn_gqa_kernel(const void* __restrict__ q_, const bf16_t* __restrict__ kc,
                                                                   const bf16_t* __restrict__ vc, const int* __restrict__ len,
                                                                   bf16_t* __restrict__ out, int Hq, int Hkv, int Lmax, float scale,
                                                                   long long out_plane, float vscale) {
    constexpr int HD = 128, U = 8;
    extern __shared__ __attribute__((aligned(16))) float d_smem[];      // sc[G][Lp]
    __shared__ float red[QD_NW][QD_GMAX];
    __shared__ float stat[2][QD_GMAX];                                  // max, sum per head
    __shared__ float part[QD_NW][QD_GMAX][HD];
    const int hk = blockIdx.x, b = blockIdx.y, t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int G = Hq / Hkv;
    const int n = PRECISE ? min(max(len[b], 1), Lmax) : min(max(len[b], 0), Lmax - 1) + 1;
    const int Lp = (Lmax + 3) & ~3;
    float* sc = d_smem;
    const bf16_t* K = kc + ((size_t)b * Hkv + hk) * Lmax * HD;
    const bf16_t* V = vc + ((size_t)b * Hkv + hk) * Lmax * HD;
    const int sub = lane >> 4, c = lane & 15;
    auto cvt8 = [](const uint4& u, float (&f)[8]) {                     // a cached K / V chunk
        const uint32_t w4[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) d_cvt2(w4[e], KVH, f[2 * e], f[2 * e + 1]);
    };
    auto cvt8q = [](const uint4& u, float (&f)[8]) {                    // the decode step's bf16 q row
        const uint32_t w4[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) d_cvt2(w4[e], false, f[2 * e], f[2 * e + 1]);
    };
    // ---- scores
    {
        float qf[QD_GMAX][8];
#pragma unroll
        for (int g = 0; g < QD_GMAX; ++g)
            if (g < G) {
                const size_t qo = ((size_t)b * Hq + (size_t)hk * G + g) * HD + 8 * c;
                if (PRECISE) {
                    const float4 a = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(q_) + qo)[0];
                    const float4 bq = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(q_) + qo)[1];
                    qf[g][0] = a.x; qf[g][1] = a.y; qf[g][2] = a.z; qf[g][3] = a.w;
                    qf[g][4] = bq.x; qf[g][5] = bq.y; qf[g][6] = bq.z; qf[g][7] = bq.w;
                } else {
                    cvt8q(*reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(q_) + qo), qf[g]);
                }
            }
        float mx[QD_GMAX];
#pragma unroll
        for (int g = 0; g < QD_GMAX; ++g) mx[g] = -3.0e38f;
        for (int j0 = wv * 4; j0 < n; j0 += 4 * QD_NW * U) {
            uint4 kv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int j = j0 + 4 * QD_NW * u + sub;
                kv[u] = j < n ? *reinterpret_cast<const uint4*>(K + (size_t)j * HD + 8 * c) : make_uint4(0u, 0u, 0u, 0u);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int j = j0 + 4 * QD_NW * u + sub;
                float kf[8];
                cvt8(kv[u], kf);
#pragma unroll
                for (int g = 0; g < QD_GMAX; ++g)
                    if (g < G) {
                        float d = 0.0f;
#pragma unroll
                        for (int e = 0; e < 8; ++e) d = fmaf(qf[g][e], kf[e], d);
                        d += __shfl_xor(d, 1);
                        d += __shfl_xor(d, 2);
                        d += __shfl_xor(d, 4);
                        d += __shfl_xor(d, 8);
                        const float tj = d * scale;
                        if (j < n) {
                            if (c == 0) sc[g * Lp + j] = tj;
                            mx[g] = fmaxf(mx[g], tj);
                        }
                    }
            }
        }
#pragma unroll
        for (int g = 0; g < QD_GMAX; ++g)
            if (g < G) {
                float m = mx[g];
                m = fmaxf(m, __shfl_xor(m, 16));
                m = fmaxf(m, __shfl_xor(m, 32));
                if (lane == 0) red[wv][g] = m;
            }
    }
    __syncthreads();
    if (t < G) {
        float m = red[0][t];
#pragma unroll
        for (int i = 1; i < QD_NW; ++i) m = fmaxf(m, red[i][t]);
        stat[0][t] = m;
    }
    __syncthreads();
    // ---- probabilities (fp32, not rounded) and their sums
    {
        float ls[QD_GMAX];
#pragma unroll
        for (int g = 0; g < QD_GMAX; ++g) ls[g] = 0.0f;
        for (int j = t; j < n; j += 64 * QD_NW)
#pragma unroll
            for (int g = 0; g < QD_GMAX; ++g)
                if (g < G) {
                    const float p = __expf(sc[g * Lp + j] - stat[0][g]);
                    sc[g * Lp + j] = p;
                    ls[g] += p;
                }
#pragma unroll
        for (int g = 0; g < QD_GMAX; ++g)
            if (g < G) {
                float v = ls[g];
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
                if (lane == 0) red[wv][g] = v;
            }
    }
    __syncthreads();
    if (t < G) {
        float v = red[0][t];
#pragma unroll
        for (int i = 1; i < QD_NW; ++i) v += red[i][t];
        stat[1][t] = v;
    }
    // ---- output
    {
        float acc[QD_GMAX][8];
#pragma unroll
        for (int g = 0; g < QD_GMAX; ++g)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[g][e] = 0.0f;
        for (int j0 = wv * 4; j0 < n; j0 += 4 * QD_NW * U) {
            uint4 vv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int j = j0 + 4 * QD_NW * u + sub;
                vv[u] = j < n ? *reinterpret_cast<const uint4*>(V + (size_t)j * HD + 8 * c) : make_uint4(0u, 0u, 0u, 0u);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int j = j0 + 4 * QD_NW * u + sub;
                if (j < n) {
                    float vf[8];
                    cvt8(vv[u], vf);
#pragma unroll
                    for (int g = 0; g < QD_GMAX; ++g)
                        if (g < G) {
                            const float p = sc[g * Lp + j];
#pragma unroll
                            for (int e = 0; e < 8; ++e) acc[g][e] = fmaf(p, vf[e], acc[g][e]);
                        }
                }
            }
        }
#pragma unroll
        for (int g = 0; g < QD_GMAX; ++g)
            if (g < G) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float v = acc[g][e];
                    v += __shfl_xor(v, 16);
                    v += __shfl_xor(v, 32);
                    if (sub == 0) part[wv][g][8 * c + e] = v;
                }
            }
    }
    __syncthreads();
    for (int i = t; i < G * HD; i += 64 * QD_NW) {
        const int g = i / HD, d = i - g * HD;
        float o = part[0][g][d];
#pragma unroll
        for (int i2 = 1; i2 < QD_NW; ++i2) o += part[i2][g][d];
        o = o / stat[1][g];
        if (KVH) o *= vscale;
        const bf16_t hi = d_f2bf(o);
        out[((size_t)b * Hq + (size_t)hk * G + g) * HD + d] = hi;
        if (PRECISE) out[out_plane + ((size_t)b * Hq + (size_t)hk * G + g) * HD + d] = d_f2bf(o - d_bf2f(hi));
    }
}

template <bool PRECISE, bool KVH = false>
static hipError_t launch_decode_attn_t(const void* q, const bf16_t* kc, const bf16_t* vc, const int* len, bf16_t* out, int B, int Hq, int Hkv, int Lmax,
                                       float scale, long long out_plane, hipStream_t s, float vscale = 1.0f) {
    if (B <= 0 || Hq <= 0 || Hkv <= 0 || (Hq % Hkv) != 0 || Lmax <= 0 || Lmax > 36864 /* VQS_QWEN_MAX_CACHE_POSITIONS */ || B > 65535) return hipErrorInvalidValue;
    {   // grouped-query form wherever its G score rows fit LDS (Lmax <= ~4 300 positions at G = 7); the per-query-head kernel beyond
        const int G = Hq / Hkv;
        const size_t glds = (size_t)G * ((Lmax + 3) & ~3) * sizeof(float);
        if (G <= QD_GMAX && glds <= 120 * 1024) {
            if (glds > 48 * 1024) {
                const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(qwen_decode_attn_gqa_kernel<PRECISE, KVH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)glds);
                if (e != hipSuccess) return e;
            }
            hipLaunchKernelGGL((qwen_decode_attn_gqa_kernel<PRECISE, KVH>), dim3((unsigned)Hkv, (unsigned)B), dim3(64 * QD_NW), glds, s, q, kc, vc, len, out, Hq, Hkv, Lmax, scale, out_plane, vscale);
            return hipGetLastError();
        }
    }
    const size_t lds = (size_t)Lmax * sizeof(float);
    if (lds > 48 * 1024) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(qwen_decode_attn_kernel<PRECISE, KVH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((qwen_decode_attn_kernel<PRECISE, KVH>), dim3((unsigned)Hq, (unsigned)B), dim3(256), lds, s, q, kc, vc, len, out, Hq, Hkv, Lmax, scale, out_plane, vscale);
    return hipGetLastError();
}

hipError_t launch_qwen_decode_attn(const bf16_t* q, const bf16_t* kc, const bf16_t* vc, const int* len, bf16_t* out, int B, int Hq,
                                   int Hkv, int Lmax, float scale, hipStream_t s) {
    return launch_decode_attn_t<false>(q, kc, vc, len, out, B, Hq, Hkv, Lmax, scale, 0, s);
}

// The precise tail's attention (vqs_qwen.cpp): q fp32 [B, Hq * 128] (rotated), keys / values = the prefill's head-major K / V of this layer
// ([B, Hkv, Lmax, 128] bf16, Lmax = the batch's padded length), count[b] = the sample's valid length; out = split-bf16 [2][B][Hq * 128].
hipError_t launch_qwen_tail_attn(const float* q, const bf16_t* k, const bf16_t* v, const int* count, bf16_t* out, long long out_plane, int B, int Hq,
                                 int Hkv, int Lmax, float scale, hipStream_t s, bool kv_f16, float vscale) {
    // kv_f16: K / V are the fp16 prefill's tensors; `scale` already carries 1 / the key scale, the output is multiplied by vscale
    if (kv_f16) return launch_decode_attn_t<true, true>(q, k, v, count, out, B, Hq, Hkv, Lmax, scale, out_plane, s, vscale);
    return launch_decode_attn_t<true>(q, k, v, count, out, B, Hq, Hkv, Lmax, scale, out_plane, s);
}

// K / V of an fp16 prefill layer into the bf16 cache the decode step reads: dst[r][0 : cols) = bf16(fp16 src[r][0 : cols) * unscale),
// rows `src_ld` / `dst_ld` elements apart (cols % 8 == 0)
__global__ void __launch_bounds__(256) f16_to_bf16_rows_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, int cols8, long long src_ld,
                                                               long long dst_ld, float unscale) {
    const long long r = blockIdx.y;
    for (int c = blockIdx.x * 256 + threadIdx.x; c < cols8; c += gridDim.x * 256) {
        const uint4 u = *reinterpret_cast<const uint4*>(src + r * src_ld + 8 * (long long)c);
        const uint32_t w4[4] = {u.x, u.y, u.z, u.w};
        uint32_t o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float a, b;
            d_cvt2(w4[e], true, a, b);
            o[e] = (uint32_t)d_f2bf(a * unscale) | ((uint32_t)d_f2bf(b * unscale) << 16);
        }
        *reinterpret_cast<uint4*>(dst + r * dst_ld + 8 * (long long)c) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}
hipError_t launch_f16_to_bf16_rows(const bf16_t* src, bf16_t* dst, int rows, long long cols, long long src_ld, long long dst_ld, float unscale, hipStream_t s) {
    if (rows <= 0 || rows > 65535 || cols <= 0 || (cols % 8) != 0) return hipErrorInvalidValue;
    const long long c8 = cols / 8;
    hipLaunchKernelGGL(f16_to_bf16_rows_kernel, dim3((unsigned)std::min<long long>((c8 + 255) / 256, 1024), (unsigned)rows), dim3(256), 0, s, src, dst, (int)c8,
                       src_ld, dst_ld, unscale);
    return hipGetLastError();
}

// The precise tail's rotary embedding: q columns of the fp32 q|k|v row of each sample's LAST position, rotated by that position's table
// (rope_qk_kernel's arithmetic -- x[i] c[i] - x[i + half] s[i], x[i + half] c[i] + x[i] s[i] -- without its bf16 rounding), -> fp32 [B, Hq * hd].
// Block (query head, sample), 64 threads; table row = row[b] (the sample's last valid position in the [B * L, half] tables).
__global__ void __launch_bounds__(64) qwen_tail_rope_q_kernel(const float* __restrict__ qkv, int ld, const float* __restrict__ cs,
                                                              const float* __restrict__ sn, const int* __restrict__ row, float* __restrict__ q_out,
                                                              int Hq, int hd, int half) {
    const int h = blockIdx.x, b = blockIdx.y, t = threadIdx.x;
    const float* src = qkv + (size_t)b * ld + (size_t)h * hd;
    float* dst = q_out + ((size_t)b * Hq + h) * hd;
    const size_t tr = (size_t)row[b] * half;
    if (t < half) {
        const float a = src[t], c2 = src[t + half];
        const float c = cs[tr + t], s = sn[tr + t];
        dst[t] = a * c - c2 * s;
        dst[t + half] = c2 * c + a * s;
    }
    for (int d = 2 * half + t; d < hd; d += 64) dst[d] = src[d];
}

// dst[r, :] = src[map[r], :] (fp32 rows): the precise tail's starting state = the embedding rows of the last positions
__global__ void __launch_bounds__(256) gather_rows_f32_kernel(const float* __restrict__ src, const int* __restrict__ map, float* __restrict__ dst,
                                                              int cols4, int src_ld) {
    const float4* s4 = reinterpret_cast<const float4*>(src + (size_t)map[blockIdx.x] * src_ld);
    float4* d4 = reinterpret_cast<float4*>(dst + (size_t)blockIdx.x * cols4 * 4);
    for (int i = threadIdx.x; i < cols4; i += 256) d4[i] = s4[i];
}

hipError_t launch_gather_rows_f32(const float* src, const int* map, float* dst, int rows, int cols, int src_ld, hipStream_t s) {
    if (rows <= 0 || cols <= 0 || (cols & 3) || (src_ld & 3)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(gather_rows_f32_kernel, dim3((unsigned)rows), dim3(256), 0, s, src, map, dst, cols / 4, src_ld);
    return hipGetLastError();
}

hipError_t launch_qwen_tail_rope_q(const float* qkv, int ld, const float* cs, const float* sn, const int* row, float* q_out, int B, int Hq, int hd,
                                   int half, hipStream_t s) {
    if (B <= 0 || Hq <= 0 || half <= 0 || half > 64 || 2 * half > hd || B > 65535) return hipErrorInvalidValue;
    hipLaunchKernelGGL(qwen_tail_rope_q_kernel, dim3((unsigned)Hq, (unsigned)B), dim3(64), 0, s, qkv, ld, cs, sn, row, q_out, Hq, hd, half);
    return hipGetLastError();
}


// ---------------------------------------------------------------------------------------------------------------------------------
// Bind-time range proof of the fp16 forms (vqs_qwen.cpp compute_ranges; round 6).  Every 16-bit activation T of the fp16 forms is held as
// fp16(T * 2^-s); s comes from a PROVEN bound of |T| that depends on the weights only:
//   * RMSNorm output  x^ * g:  |x^_k| <= ||x^||_2 = sqrt(D / (1 + eps D / sum x^2)) <= sqrt(D), so |out_k| <= sqrt(D) |g_k|;
//   * a linear fed by a norm output (q|k|v, gate|up, merger mlp.0):  |sum_k W_jk g_k x^_k + b_j| <= ||W_j o g||_2 ||x^||_2 + |b_j|   (Cauchy-Schwarz);
//   * a linear fed by a tensor with element bounds u_k (o / proj over the attention output -- a convex combination of value rows --,
//     down_proj over the gated product, merger mlp.2):  |sum_k W_jk a_k + b_j| <= sum_k |W_jk| u_k + |b_j|;
//   * |SiLU(g) u| <= |g| |u| (|SiLU(x)| <= |x|, |GELU(x)| <= |x|);  rotary embedding: |x'| <= sqrt(2) max|x|.
// One wave per weight row, 16-byte loads; the row's bound goes to out[j] (optional) and its maximum over rows into `slot` (atomicMax on
// the bit pattern: the values are non-negative floats; a NaN weight poisons the slot with a NaN-patterned maximum, which the host refuses).
// ---------------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float d_wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
__device__ __forceinline__ void d_slot_max(float* slot, float v) {
    if (!(v == v)) v = __uint_as_float(0x7fc00000u);                     // NaN: the largest bit pattern wins and stays
    atomicMax(reinterpret_cast<unsigned int*>(slot), __float_as_uint(v));
}
// mode 0: out[j] = R * sqrt(sum_k (W[j,k] g[k % g_len])^2) + |bias[j]|      (g bf16)
// mode 1: out[j] = sum_k |W[j,k]| u[k] + |bias[j]|                           (u fp32 [K]; u == nullptr: the constant *uconst)
__global__ void __launch_bounds__(256) rowbound_kernel(const bf16_t* __restrict__ W, long long ldw, int N, int K, int mode, const bf16_t* __restrict__ g,
                                                       int g_len, float R, const float* __restrict__ u, const float* __restrict__ uconst,
                                                       const bf16_t* __restrict__ bias, float* __restrict__ out, float* __restrict__ slot) {
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (j >= N) return;
    const bf16_t* wr = W + (size_t)j * ldw;
    const float uc = (mode == 1 && u == nullptr) ? *uconst : 0.0f;
    float acc = 0.0f;
    for (int k = lane; k < K; k += 64) {
        const float w = d_bf2f(wr[k]);
        if (mode == 0) {
            const float t = w * d_bf2f(g[k % g_len]);
            acc = fmaf(t, t, acc);
        } else {
            acc = fmaf(fabsf(w), u ? u[k] : uc, acc);
        }
    }
    acc = d_wave_sum(acc);
    float b = mode == 0 ? R * sqrtf(acc) : acc;
    if (bias) b += fabsf(d_bf2f(bias[j]));
    if (lane == 0) {
        if (out) out[j] = b;
        d_slot_max(slot, b);
    }
}
hipError_t launch_rowbound(const bf16_t* W, long long ldw, int N, int K, int mode, const bf16_t* g, int g_len, float R, const float* u,
                           const float* uconst, const bf16_t* bias, float* out, float* slot, hipStream_t s) {
    if (N <= 0 || K <= 0 || (mode == 0 && (!g || g_len <= 0)) || (mode == 1 && !u && !uconst) || !slot) return hipErrorInvalidValue;
    hipLaunchKernelGGL(rowbound_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, s, W, ldw, N, K, mode, g, g_len, R, u, uconst, bias, out, slot);
    return hipGetLastError();
}
// slot = max(slot, R * max_k |x[k]|) over a bf16 vector / matrix of n elements (norm weights: the norm output's bound; weights: their fp16 range check)
__global__ void __launch_bounds__(256) absmax_bf16_kernel(const bf16_t* __restrict__ x, size_t n, float R, float* __restrict__ slot) {
    float m = 0.0f;
    bool nan = false;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float v = fabsf(d_bf2f(x[i]));
        nan |= !(v == v);
        m = fmaxf(m, v);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if (__any(nan)) m = __uint_as_float(0x7fc00000u);
    if ((threadIdx.x & 63) == 0) d_slot_max(slot, R * m);
}
hipError_t launch_absmax_bf16(const bf16_t* x, size_t n, float R, float* slot, hipStream_t s) {
    if (!x || n == 0 || !slot) return hipErrorInvalidValue;
    const unsigned blocks = (unsigned)std::min<size_t>((n + 255) / 256, 2048);
    hipLaunchKernelGGL(absmax_bf16_kernel, dim3(blocks), dim3(256), 0, s, x, n, R, slot);
    return hipGetLastError();
}
// gated product: rb = row bounds of the PACKED gate|up weight (blocks of 32 gate rows | 32 up rows); u_act[j] = rb[gate(j)] * rb[up(j)] for
// j < mlp_p, 0 for the pad columns up to ld; slot = max_j
__global__ void __launch_bounds__(256) gate_pair_bound_kernel(const float* __restrict__ rb, int mlp_p, int ld, float* __restrict__ u_act, float* __restrict__ slot) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    float v = 0.0f;
    if (j < mlp_p) v = rb[(j >> 5) * 64 + (j & 31)] * rb[(j >> 5) * 64 + 32 + (j & 31)];
    if (j < ld) u_act[j] = v;
    float m = v;
    const bool nan = !(v == v);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if (__any(nan)) m = __uint_as_float(0x7fc00000u);
    if ((threadIdx.x & 63) == 0) d_slot_max(slot, m);
}
hipError_t launch_gate_pair_bound(const float* rb, int mlp_p, int ld, float* u_act, float* slot, hipStream_t s) {
    if (!rb || mlp_p <= 0 || ld < mlp_p || (mlp_p & 31) || !u_act || !slot) return hipErrorInvalidValue;
    hipLaunchKernelGGL(gate_pair_bound_kernel, dim3((unsigned)((ld + 255) / 256)), dim3(256), 0, s, rb, mlp_p, ld, u_act, slot);
    return hipGetLastError();
}

}  // namespace vqs
