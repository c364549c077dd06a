// HBM-bound kernels of the CLIP-FlanT5 path (K2, K3, K7-K10 of SURVEY.md §8(a-bis)) for gfx950.
// All are row-parallel streaming kernels: one 64-lane wave per row where a row reduction is needed,
// 16-byte accesses per lane, fp32 math.
#include "vqs_kernels.h"

namespace vqs {

__device__ __forceinline__ float e_bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ bf16_t e_f2bf(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}
__device__ __forceinline__ uint32_t e_pack2(float a, float b) {
    return (uint32_t)e_f2bf(a) | ((uint32_t)e_f2bf(b) << 16);
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    return v;
}

// ------------------------------------------------------------------------------------------------
// Norm kernels.  One 64-lane wave per row; fp32 residual stream in, bf16 GEMM operand out.
//   T5 RMSNorm (HF models/t5/modeling_t5.py:59-72): out = w * x * rsqrt(mean(x^2) + eps)
//   CLIP LayerNorm (torch.nn.LayerNorm at HF models/clip/modeling_clip.py:357-360,642): fp32 statistics
// With `delta` the residual update of the previous sub-layer (hidden = hidden + sublayer_out,
// modeling_t5.py:140,400,431; modeling_clip.py:366,371) is fused in front: x += delta is written back to the fp32
// stream, then normalised.  `delta` is the sub-layer's output GEMM result in bf16 -- exactly what the reference's bf16
// nn.Linear hands to its residual add -- while the stream itself stays fp32 here.
// Algorithmic HBM bytes per element with delta: 4 (x in) + 2 (delta) + 4 (x out) + 2 (out) = 12.
// ADD modes: 0 none; 1 x += delta, stored; 2 x + delta normalised but NOT stored (the caller keeps `delta` alive and
// hands it to the next norm again: 8 B/elem); 3 x = (x + delta) + delta2, stored (14 B/elem).  A layer's two norms as
// mode 2 then mode 3 move 22 B/elem instead of 24 and produce bit-identical streams ((x + d1) + d2 in fp32 either way).
// D = 1024 / 2048 / 4096 (the model widths): the row lives in registers (NV float4 per lane), every load of the row is in
// flight before the first use and nothing is read twice.  Other D (test configs): looped fallback, second pass from L2.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float4 bf4_to_f4(uint2 u) {
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                       __uint_as_float(u.y & 0xffff0000u));
}
typedef __attribute__((ext_vector_type(2))) __bf16 e_bf16x2;
__device__ __forceinline__ uint32_t e_pack2_hw(float a, float b) {   // v_cvt_pk_bf16_f32 (RNE, same result as e_pack2)
    e_bf16x2 v;
    v[0] = (__bf16)a;
    v[1] = (__bf16)b;
    return __builtin_bit_cast(uint32_t, v);
}

// IEEE fp16 forms of the two conversions (the fp16 vision tower: deltas in, operand out)
typedef __attribute__((ext_vector_type(4))) _Float16 e_f16x4;
typedef __attribute__((ext_vector_type(2))) _Float16 e_f16x2;
__device__ __forceinline__ float4 h4_to_f4(uint2 u) {
    const e_f16x4 h = __builtin_bit_cast(e_f16x4, u);
    return make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
}
__device__ __forceinline__ uint32_t e_pack2_h(float a, float b) {    // v_cvt_pk_f16_f32 on gfx950 (RNE; overflow -> inf)
    e_f16x2 v;
    v[0] = (_Float16)a;
    v[1] = (_Float16)b;
    return __builtin_bit_cast(uint32_t, v);
}
template <bool F16>
__device__ __forceinline__ float4 d4_to_f4(uint2 u) {
    if constexpr (F16) return h4_to_f4(u);
    else return bf4_to_f4(u);
}
template <bool F16>
__device__ __forceinline__ uint32_t e_pack2_t(float a, float b) {
    if constexpr (F16) return e_pack2_h(a, b);
    else return e_pack2_hw(a, b);
}

// The fp32 stream is touched once per norm and not again before gigabytes of other traffic have passed: its loads and
// stores are non-temporal (in situ +0.4 % for the loads alone, 0 for the stores alone, +0.76 % for both against cached
// accesses, profiles/r1_call88_ab_norm_nt.txt), which leaves L2 / Infinity Cache to the bf16 operands the GEMMs read
// next.  Making the deltas' last reads non-temporal as well measured nothing, so they stay plain loads.
typedef float e_f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld_stream(const float4* p) {
    const e_f4v v = __builtin_nontemporal_load(reinterpret_cast<const e_f4v*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
typedef unsigned int e_u2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint2 ld_delta_last(const uint2* p) {      // a delta's last reader
    return *p;
}
__device__ __forceinline__ void st_stream(float4* p, float4 v) {
    e_f4v u;
    u.x = v.x; u.y = v.y; u.z = v.z; u.w = v.w;
    __builtin_nontemporal_store(u, reinterpret_cast<e_f4v*>(p));
}

// KIND 0: RMSNorm (bsh unused); KIND 1: LayerNorm.  F16: the 16-bit output is IEEE fp16; DF16: the deltas are (w, bsh stay bf16).  The
// vision tower has both (its sub-layer outputs are fp16 tensors); the T5 encoder of option enc_fp16 has an fp16 operand out and bf16
// deltas in (Flan-T5's sub-layer outputs do not fit fp16).
// SC (round 6, the Qwen2.5-VL row's range-safe fp16 forms): the deltas are held behind power-of-two scales -- x += delta * sc.d1 (+ delta2 *
// sc.d2) -- and the operand leaves as fp16(value * sc.out); all three 1: the unscaled kernel bit for bit.
struct NormScales { float d1, d2, out; };
template <int NV, int KIND, int ADD, bool OUT_F32, bool F16 = false, bool DF16 = F16, bool SC = false>
__global__ void __launch_bounds__(256) norm_rows_reg_kernel(float* __restrict__ x, const bf16_t* __restrict__ delta,
                                                            const bf16_t* __restrict__ delta2,
                                                            const bf16_t* __restrict__ w, const bf16_t* __restrict__ bsh,
                                                            void* __restrict__ out, int M, float eps, int out_ld, NormScales sc) {
    constexpr int D = NV * 256;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= M) return;
    float4* xr = reinterpret_cast<float4*>(x + (size_t)row * D);
    float4 v[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) v[j] = ld_stream(xr + lane + 64 * j);
    if (ADD != 0) {
        const uint2* dr = reinterpret_cast<const uint2*>(delta + (size_t)row * D);
        uint2 d[NV];
#pragma unroll
        for (int j = 0; j < NV; ++j) d[j] = ADD == 2 ? dr[lane + 64 * j] : ld_delta_last(dr + lane + 64 * j);
        if (ADD == 3) {
            const uint2* er = reinterpret_cast<const uint2*>(delta2 + (size_t)row * D);
            uint2 e[NV];
#pragma unroll
            for (int j = 0; j < NV; ++j) e[j] = ld_delta_last(er + lane + 64 * j);
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                float4 f = d4_to_f4<DF16>(d[j]), g = d4_to_f4<DF16>(e[j]);
                if constexpr (SC) {
                    f.x *= sc.d1; f.y *= sc.d1; f.z *= sc.d1; f.w *= sc.d1;
                    g.x *= sc.d2; g.y *= sc.d2; g.z *= sc.d2; g.w *= sc.d2;
                }
                v[j].x = (v[j].x + f.x) + g.x; v[j].y = (v[j].y + f.y) + g.y;
                v[j].z = (v[j].z + f.z) + g.z; v[j].w = (v[j].w + f.w) + g.w;
                st_stream(xr + lane + 64 * j, v[j]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                float4 f = d4_to_f4<DF16>(d[j]);
                if constexpr (SC) { f.x *= sc.d1; f.y *= sc.d1; f.z *= sc.d1; f.w *= sc.d1; }
                v[j].x += f.x; v[j].y += f.y; v[j].z += f.z; v[j].w += f.w;
                if (ADD == 1) st_stream(xr + lane + 64 * j, v[j]);
            }
        }
    }
    float mu = 0.0f;
    if (KIND == 1) {
        float s1 = 0.0f;
#pragma unroll
        for (int j = 0; j < NV; ++j) s1 += (v[j].x + v[j].y) + (v[j].z + v[j].w);
        mu = wave_sum(s1) * (1.0f / (float)D);
    }
    float s2 = 0.0f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const float a = v[j].x - mu, b = v[j].y - mu, c = v[j].z - mu, d = v[j].w - mu;
        s2 += a * a + b * b + c * c + d * d;
    }
    const float rs = rsqrtf(wave_sum(s2) * (1.0f / (float)D) + eps);
    const uint2* wr = reinterpret_cast<const uint2*>(w);
    const uint2* br = reinterpret_cast<const uint2*>(bsh);
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int i = lane + 64 * j;
        const float4 wv = bf4_to_f4(wr[i]);
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (KIND == 1) bv = bf4_to_f4(br[i]);
        float o0 = (v[j].x - mu) * rs * wv.x + bv.x, o1 = (v[j].y - mu) * rs * wv.y + bv.y;
        float o2 = (v[j].z - mu) * rs * wv.z + bv.z, o3 = (v[j].w - mu) * rs * wv.w + bv.w;
        if constexpr (SC) { o0 *= sc.out; o1 *= sc.out; o2 *= sc.out; o3 *= sc.out; }
        if (OUT_F32) {
            reinterpret_cast<float4*>(reinterpret_cast<float*>(out) + (size_t)row * out_ld)[i] = make_float4(o0, o1, o2, o3);
        } else {
            uint2 o;
            o.x = e_pack2_t<F16>(o0, o1);
            o.y = e_pack2_t<F16>(o2, o3);
            reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(out) + (size_t)row * out_ld)[i] = o;
        }
    }
}

// any D % 4 == 0
template <int KIND, int ADD, bool OUT_F32, bool F16 = false, bool DF16 = F16, bool SC = false>
__global__ void __launch_bounds__(256) norm_rows_loop_kernel(float* __restrict__ x, const bf16_t* __restrict__ delta,
                                                             const bf16_t* __restrict__ delta2,
                                                             const bf16_t* __restrict__ w, const bf16_t* __restrict__ bsh,
                                                             void* __restrict__ out, int M, int D, float eps, int out_ld, NormScales sc) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= M) return;
    float4* xr = reinterpret_cast<float4*>(x + (size_t)row * D);
    const uint2* dr = ADD != 0 ? reinterpret_cast<const uint2*>(delta + (size_t)row * D) : nullptr;
    const uint2* er = ADD == 3 ? reinterpret_cast<const uint2*>(delta2 + (size_t)row * D) : nullptr;
    const int nv = D >> 2;
    // the row value as the later passes see it: stored modes re-read what the first pass wrote (same lane), mode 2
    // (not stored) re-adds the delta
    auto value = [&](int i) {
        float4 v = xr[i];
        if (ADD == 2) {
            float4 d = d4_to_f4<DF16>(dr[i]);
            if constexpr (SC) { d.x *= sc.d1; d.y *= sc.d1; d.z *= sc.d1; d.w *= sc.d1; }
            v.x += d.x; v.y += d.y; v.z += d.z; v.w += d.w;
        }
        return v;
    };
    float s1 = 0.0f;
    if (ADD == 1 || ADD == 3) {
        for (int i = lane; i < nv; i += 64) {
            float4 v = xr[i];
            float4 d = d4_to_f4<DF16>(dr[i]);
            if constexpr (SC) { d.x *= sc.d1; d.y *= sc.d1; d.z *= sc.d1; d.w *= sc.d1; }
            v.x += d.x; v.y += d.y; v.z += d.z; v.w += d.w;
            if (ADD == 3) {
                float4 e = d4_to_f4<DF16>(er[i]);
                if constexpr (SC) { e.x *= sc.d2; e.y *= sc.d2; e.z *= sc.d2; e.w *= sc.d2; }
                v.x += e.x; v.y += e.y; v.z += e.z; v.w += e.w;
            }
            xr[i] = v;
            s1 += (v.x + v.y) + (v.z + v.w);
        }
    } else if (KIND == 1) {
        for (int i = lane; i < nv; i += 64) {
            const float4 v = value(i);
            s1 += (v.x + v.y) + (v.z + v.w);
        }
    }
    const float mu = KIND == 1 ? wave_sum(s1) / (float)D : 0.0f;
    float s2 = 0.0f;
    for (int i = lane; i < nv; i += 64) {
        const float4 v = value(i);
        const float a = v.x - mu, b = v.y - mu, c = v.z - mu, d = v.w - mu;
        s2 += a * a + b * b + c * c + d * d;
    }
    const float rs = rsqrtf(wave_sum(s2) / (float)D + eps);
    const uint2* wr = reinterpret_cast<const uint2*>(w);
    const uint2* br = reinterpret_cast<const uint2*>(bsh);
    for (int i = lane; i < nv; i += 64) {
        const float4 v = value(i);
        const float4 wv = bf4_to_f4(wr[i]);
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (KIND == 1) bv = bf4_to_f4(br[i]);
        float o0 = (v.x - mu) * rs * wv.x + bv.x, o1 = (v.y - mu) * rs * wv.y + bv.y;
        float o2 = (v.z - mu) * rs * wv.z + bv.z, o3 = (v.w - mu) * rs * wv.w + bv.w;
        if constexpr (SC) { o0 *= sc.out; o1 *= sc.out; o2 *= sc.out; o3 *= sc.out; }
        if (OUT_F32) {
            reinterpret_cast<float4*>(reinterpret_cast<float*>(out) + (size_t)row * out_ld)[i] = make_float4(o0, o1, o2, o3);
        } else {
            uint2 o;
            o.x = e_pack2_t<F16>(o0, o1);
            o.y = e_pack2_t<F16>(o2, o3);
            reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(out) + (size_t)row * out_ld)[i] = o;
        }
    }
}

// out_ld: row pitch of `out` in elements (0 = D; a multiple of 4).  The stream x and the deltas are always dense.
template <int KIND, int ADD, bool OUT_F32, bool F16 = false, bool DF16 = F16, bool SC = false>
static hipError_t launch_norm_t(float* x, const bf16_t* delta, const bf16_t* delta2, const bf16_t* w, const bf16_t* b, void* out,
                                int M, int D, float eps, hipStream_t s, int out_ld = 0, NormScales sc = NormScales{1.0f, 1.0f, 1.0f}) {
    const dim3 grid((M + 3) / 4), block(256);
    if (out_ld <= 0) out_ld = D;
    if (out_ld < D || (out_ld & 3)) return hipErrorInvalidValue;
    if (D == 1024)
        hipLaunchKernelGGL((norm_rows_reg_kernel<4, KIND, ADD, OUT_F32, F16, DF16, SC>), grid, block, 0, s, x, delta, delta2, w, b, out, M, eps, out_ld, sc);
    else if (D == 2048)
        hipLaunchKernelGGL((norm_rows_reg_kernel<8, KIND, ADD, OUT_F32, F16, DF16, SC>), grid, block, 0, s, x, delta, delta2, w, b, out, M, eps, out_ld, sc);
    else if (D == 4096)
        hipLaunchKernelGGL((norm_rows_reg_kernel<16, KIND, ADD, OUT_F32, F16, DF16, SC>), grid, block, 0, s, x, delta, delta2, w, b, out, M, eps, out_ld, sc);
    else if (D == 1280)     // Qwen2.5-VL vision tower
        hipLaunchKernelGGL((norm_rows_reg_kernel<5, KIND, ADD, OUT_F32, F16, DF16, SC>), grid, block, 0, s, x, delta, delta2, w, b, out, M, eps, out_ld, sc);
    else if (D == 3584)     // Qwen2.5-VL-7B language model
        hipLaunchKernelGGL((norm_rows_reg_kernel<14, KIND, ADD, OUT_F32, F16, DF16, SC>), grid, block, 0, s, x, delta, delta2, w, b, out, M, eps, out_ld, sc);
    else
        hipLaunchKernelGGL((norm_rows_loop_kernel<KIND, ADD, OUT_F32, F16, DF16, SC>), grid, block, 0, s, x, delta, delta2, w, b, out, M, D, eps, out_ld, sc);
    return hipGetLastError();
}

// add mode from the arguments: delta only -> 1 (stored) or 2 (store_x == false); delta + delta2 -> 3 (always stored)
static int norm_add_mode(const bf16_t* delta, const bf16_t* delta2, bool store_x) {
    if (!delta) return (delta2 || !store_x) ? -1 : 0;
    if (delta2) return store_x ? 3 : -1;
    return store_x ? 1 : 2;
}

hipError_t launch_rmsnorm(float* x, const bf16_t* delta, const bf16_t* w, bf16_t* out, int M, int D, float eps,
                          hipStream_t s, const bf16_t* delta2, bool store_x, int out_ld, bool out_f16) {
    if (D % 4) return hipErrorInvalidValue;
    if (out_f16) {          // bf16 deltas in, IEEE fp16 operand out (the T5 encoder of option enc_fp16)
        switch (norm_add_mode(delta, delta2, store_x)) {
            case 0: return launch_norm_t<0, 0, false, true, false>(x, delta, delta2, w, nullptr, out, M, D, eps, s, out_ld);
            case 1: return launch_norm_t<0, 1, false, true, false>(x, delta, delta2, w, nullptr, out, M, D, eps, s, out_ld);
            case 2: return launch_norm_t<0, 2, false, true, false>(x, delta, delta2, w, nullptr, out, M, D, eps, s, out_ld);
            case 3: return launch_norm_t<0, 3, false, true, false>(x, delta, delta2, w, nullptr, out, M, D, eps, s, out_ld);
            default: return hipErrorInvalidValue;
        }
    }
    switch (norm_add_mode(delta, delta2, store_x)) {
        case 0: return launch_norm_t<0, 0, false>(x, delta, delta2, w, nullptr, out, M, D, eps, s, out_ld);
        case 1: return launch_norm_t<0, 1, false>(x, delta, delta2, w, nullptr, out, M, D, eps, s, out_ld);
        case 2: return launch_norm_t<0, 2, false>(x, delta, delta2, w, nullptr, out, M, D, eps, s, out_ld);
        case 3: return launch_norm_t<0, 3, false>(x, delta, delta2, w, nullptr, out, M, D, eps, s, out_ld);
        default: return hipErrorInvalidValue;
    }
}

// RMSNorm of the scaled fp16 forms (the Qwen2.5-VL row, vqs_qwen.cpp): fp16 deltas behind 1 / d1 and 1 / d2, fp16 operand out behind `out`
hipError_t launch_rmsnorm_f16s(float* x, const bf16_t* delta, const bf16_t* w, bf16_t* out, int M, int D, float eps, hipStream_t s,
                               const bf16_t* delta2, bool store_x, int out_ld, float d1, float d2, float osc, bool out_bf16) {
    if (D % 4) return hipErrorInvalidValue;
    const NormScales sc{d1, d2, osc};
    if (out_bf16) {         // the language model's final norm: fp16 deltas in, bf16 operand out (the lm_head's bf16 weights)
        switch (norm_add_mode(delta, delta2, store_x)) {
            case 0: return launch_norm_t<0, 0, false, false, true, true>(x, delta, delta2, w, nullptr, out, M, D, eps, s, out_ld, sc);
            case 1: return launch_norm_t<0, 1, false, false, true, true>(x, delta, delta2, w, nullptr, out, M, D, eps, s, out_ld, sc);
            case 3: return launch_norm_t<0, 3, false, false, true, true>(x, delta, delta2, w, nullptr, out, M, D, eps, s, out_ld, sc);
            default: return hipErrorInvalidValue;
        }
    }
    switch (norm_add_mode(delta, delta2, store_x)) {
        case 0: return launch_norm_t<0, 0, false, true, true, true>(x, delta, delta2, w, nullptr, out, M, D, eps, s, out_ld, sc);
        case 1: return launch_norm_t<0, 1, false, true, true, true>(x, delta, delta2, w, nullptr, out, M, D, eps, s, out_ld, sc);
        case 2: return launch_norm_t<0, 2, false, true, true, true>(x, delta, delta2, w, nullptr, out, M, D, eps, s, out_ld, sc);
        case 3: return launch_norm_t<0, 3, false, true, true, true>(x, delta, delta2, w, nullptr, out, M, D, eps, s, out_ld, sc);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_layernorm(float* x, const bf16_t* delta, const bf16_t* w, const bf16_t* b, void* out, int out_f32, int M,
                            int D, float eps, hipStream_t s, const bf16_t* delta2, bool store_x, bool f16) {
    if (D % 4) return hipErrorInvalidValue;
    const int mode = norm_add_mode(delta, delta2, store_x);
    if (f16 && !out_f32) {  // fp16 deltas in, fp16 operand out (the fp16 vision tower)
        switch (mode) {
            case 0: return launch_norm_t<1, 0, false, true>(x, delta, delta2, w, b, out, M, D, eps, s);
            case 1: return launch_norm_t<1, 1, false, true>(x, delta, delta2, w, b, out, M, D, eps, s);
            case 2: return launch_norm_t<1, 2, false, true>(x, delta, delta2, w, b, out, M, D, eps, s);
            case 3: return launch_norm_t<1, 3, false, true>(x, delta, delta2, w, b, out, M, D, eps, s);
            default: return hipErrorInvalidValue;
        }
    }
    if (f16 && delta) return hipErrorInvalidValue;       // an fp32-output norm with an fp16 delta: not a form the tower has
    if (out_f32) {          // fp32 output is only needed for the plain and the stored single-delta form
        if (mode == 0) return launch_norm_t<1, 0, true>(x, delta, delta2, w, b, out, M, D, eps, s);
        if (mode == 1) return launch_norm_t<1, 1, true>(x, delta, delta2, w, b, out, M, D, eps, s);
        return hipErrorInvalidValue;
    }
    switch (mode) {
        case 0: return launch_norm_t<1, 0, false>(x, delta, delta2, w, b, out, M, D, eps, s);
        case 1: return launch_norm_t<1, 1, false>(x, delta, delta2, w, b, out, M, D, eps, s);
        case 2: return launch_norm_t<1, 2, false>(x, delta, delta2, w, b, out, M, D, eps, s);
        case 3: return launch_norm_t<1, 3, false>(x, delta, delta2, w, b, out, M, D, eps, s);
        default: return hipErrorInvalidValue;
    }
}

// ------------------------------------------------------------------------------------------------
// Patch gather for the 14x14/14 convolution (HF models/clip/modeling_clip.py:209-210) as a GEMM A
// operand: out[(n*G + gy)*G + gx][c*p*p + ky*p + kx] = pixels[n][c][gy*p+ky][gx*p+kx]; columns
// >= 3*p*p are zero.  One workgroup per (image, patch row gy): reads whole image rows (coalesced),
// writes whole output rows.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) im2col_kernel(const bf16_t* __restrict__ px, bf16_t* __restrict__ out, int img,
                                                     int patch, int kpad) {
    const int G = img / patch;
    const int n = blockIdx.y, gy = blockIdx.x;
    const int kreal = 3 * patch * patch;
    bf16_t* obase = out + ((size_t)(n * G + gy) * G) * kpad;
    const int total = G * kpad;
    for (int i = threadIdx.x; i < total; i += 256) {
        const int gx = i / kpad, kk = i - gx * kpad;
        bf16_t v = 0;
        if (kk < kreal) {
            const int c = kk / (patch * patch);
            const int rem = kk - c * patch * patch;
            const int ky = rem / patch, kx = rem - ky * patch;
            v = px[(((size_t)n * 3 + c) * img + (gy * patch + ky)) * img + gx * patch + kx];
        }
        obase[i] = v;
    }
}

hipError_t launch_im2col(const bf16_t* pixels, bf16_t* out, int N, int img, int patch, int kpad, hipStream_t s) {
    const int G = img / patch;
    hipLaunchKernelGGL(im2col_kernel, dim3(G, N), dim3(256), 0, s, pixels, out, img, patch, kpad);
    return hipGetLastError();
}

// hidden[n,0,:] = cls + pos[0]; hidden[n,1+p,:] = patch_out[n*P+p] + pos[1+p]   (modeling_clip.py:213-217)
__global__ void __launch_bounds__(256) vit_assemble_kernel(const float* __restrict__ patch_out,
                                                           const bf16_t* __restrict__ cls,
                                                           const bf16_t* __restrict__ pos, float* __restrict__ hidden,
                                                           int P, int D) {
    const int n = blockIdx.y, t = blockIdx.x;   // token 0..P
    float* hrow = hidden + ((size_t)n * (P + 1) + t) * D;
    const bf16_t* prow = pos + (size_t)t * D;
    for (int i = threadIdx.x; i < D; i += 256) {
        const float base = (t == 0) ? e_bf2f(cls[i]) : patch_out[((size_t)n * P + (t - 1)) * D + i];
        hrow[i] = base + e_bf2f(prow[i]);
    }
}

hipError_t launch_vit_assemble(const float* patch_out, const bf16_t* cls, const bf16_t* pos, float* hidden, int N,
                               int P, int D, hipStream_t s) {
    hipLaunchKernelGGL(vit_assemble_kernel, dim3(P + 1, N), dim3(256), 0, s, patch_out, cls, pos, hidden, P, D);
    return hipGetLastError();
}

// feature select: hidden_states[-2][:, 1:] -> bf16 rows for the projector GEMM
// F16: the pending delta is an fp16 tensor (the fp16 tower blocks); OF16: the selected features leave as fp16 (option proj_fp16)
template <bool F16, bool OF16 = F16>
__global__ void __launch_bounds__(256) drop_cls_cast_kernel(float* __restrict__ hidden,
                                                            const bf16_t* __restrict__ delta, bf16_t* __restrict__ out,
                                                            int P, int D, float oscale) {
    const int n = blockIdx.y, pidx = blockIdx.x;
    const size_t roff = ((size_t)n * (P + 1) + 1 + pidx) * D;
    float4* src = reinterpret_cast<float4*>(hidden + roff);
    const uint2* dsrc = delta ? reinterpret_cast<const uint2*>(delta + roff) : nullptr;
    uint2* dst = reinterpret_cast<uint2*>(out + ((size_t)n * P + pidx) * D);
    for (int i = threadIdx.x; i < (D >> 2); i += 256) {
        float4 v = src[i];
        if (dsrc) {
            const float4 d = d4_to_f4<F16>(dsrc[i]);
            v.x += d.x; v.y += d.y; v.z += d.z; v.w += d.w;
            src[i] = v;      // materialise hidden_states[-2] for the patch rows (the CLS row is never needed)
        }
        uint2 o;
        if constexpr (OF16) {     // oscale: a power of two the range proof put the stream behind (1 = none; exact)
            o.x = e_pack2_h(v.x * oscale, v.y * oscale);
            o.y = e_pack2_h(v.z * oscale, v.w * oscale);
        } else {
            o.x = e_pack2(v.x, v.y);
            o.y = e_pack2(v.z, v.w);
        }
        dst[i] = o;
    }
}

hipError_t launch_drop_cls_cast(float* hidden, const bf16_t* delta, bf16_t* out, int N, int P, int D, hipStream_t s, bool f16, int out_f16, float oscale) {
    if (D % 4) return hipErrorInvalidValue;
    const bool of16 = out_f16 < 0 ? f16 : out_f16 != 0;
    if (!of16 && oscale != 1.0f) return hipErrorInvalidValue;
    if (f16 && of16) hipLaunchKernelGGL((drop_cls_cast_kernel<true, true>), dim3(P, N), dim3(256), 0, s, hidden, delta, out, P, D, oscale);
    else if (f16) hipLaunchKernelGGL((drop_cls_cast_kernel<true, false>), dim3(P, N), dim3(256), 0, s, hidden, delta, out, P, D, oscale);
    else if (of16) hipLaunchKernelGGL((drop_cls_cast_kernel<false, true>), dim3(P, N), dim3(256), 0, s, hidden, delta, out, P, D, oscale);
    else hipLaunchKernelGGL((drop_cls_cast_kernel<false, false>), dim3(P, N), dim3(256), 0, s, hidden, delta, out, P, D, oscale);
    return hipGetLastError();
}

// 16-bit type changes, 8 elements per lane: bf16 -> fp16 (the fp16 tower's copies of a bf16 checkpoint's weights; exact for
// 2^-14 <= |w| < 65 520, below that the fp16 subnormal grid / zero, RNE) and fp16 -> bf16 (its projector output back into the bf16
// image-feature tensor of the C ABI; RNE from 11 to 8 significant bits).
template <bool TO_F16>
__global__ void __launch_bounds__(256) cast16_kernel(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n8) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    const uint4 v = in[i];
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if constexpr (TO_F16) {
            o[k] = e_pack2_h(__uint_as_float(w[k] << 16), __uint_as_float(w[k] & 0xffff0000u));
        } else {
            const e_f16x2 h = __builtin_bit_cast(e_f16x2, w[k]);
            o[k] = e_pack2_hw((float)h[0], (float)h[1]);
        }
    }
    out[i] = make_uint4(o[0], o[1], o[2], o[3]);
}

hipError_t launch_cast16(const bf16_t* in, bf16_t* out, size_t n, bool to_f16, hipStream_t s) {
    if (n % 8) return hipErrorInvalidValue;
    const size_t n8 = n / 8;
    if (n8 == 0) return hipSuccess;
    const dim3 grid((unsigned)((n8 + 255) / 256));
    if (to_f16) hipLaunchKernelGGL(cast16_kernel<true>, grid, dim3(256), 0, s, (const uint4*)in, (uint4*)out, n8);
    else hipLaunchKernelGGL(cast16_kernel<false>, grid, dim3(256), 0, s, (const uint4*)in, (uint4*)out, n8);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Prompt scan: ids [B,L] int32, trailing pad (0), exactly one sentinel (-200) among the non-pad ids
// (/root/reference/t2v_metrics/models/vqascore_models/mm_utils.py:164-179).  err_flag bit0 = bad prompt.
// ------------------------------------------------------------------------------------------------
__global__ void prompt_scan_kernel(const int* __restrict__ ids, int B, int L, int P, int* __restrict__ sent_pos,
                                   int* __restrict__ enc_len, int* __restrict__ err_flag) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int* r = ids + (size_t)b * L;
    int n = 0, sp = -1, nsent = 0;
    bool seen_pad = false, bad = false;
    for (int i = 0; i < L; ++i) {
        const int v = r[i];
        if (v == 0) { seen_pad = true; continue; }
        if (seen_pad) bad = true;                 // non-pad after pad: not right-padded
        if (v == -200) { sp = n; ++nsent; }
        ++n;
    }
    if (nsent != 1) bad = true;
    if (bad) atomicOr(err_flag, 1);
    sent_pos[b] = sp < 0 ? 0 : sp;
    enc_len[b] = bad ? 1 : (n - 1 + P);
}

hipError_t launch_prompt_scan(const int* ids, int B, int L, int P, int* sent_pos, int* enc_len, int* err_flag,
                              hipStream_t s) {
    hipLaunchKernelGGL(prompt_scan_kernel, dim3((B + 63) / 64), dim3(64), 0, s, ids, B, L, P, sent_pos, enc_len,
                       err_flag);
    return hipGetLastError();
}

// inputs_embeds[b, s] = shared[id] | proj[img_index[b], s - sent_pos] | 0   
__global__ void __launch_bounds__(256) embed_splice_kernel(const int* __restrict__ ids, const int* __restrict__ sent_pos,
                                                           const int* __restrict__ enc_len,
                                                           const int* __restrict__ img_index,
                                                           const bf16_t* __restrict__ shared,
                                                           const bf16_t* __restrict__ proj, float* __restrict__ out,
                                                           int L, int P, int D, int vocab) {
    const int b = blockIdx.y, s = blockIdx.x;
    const int S_e = L - 1 + P;
    float4* orow = reinterpret_cast<float4*>(out + ((size_t)b * S_e + s) * D);
    const int sp = sent_pos[b], len = enc_len[b];
    const bf16_t* src = nullptr;
    if (s < len) {
        if (s < sp) {
            int id = ids[(size_t)b * L + s];
            id = min(max(id, 0), vocab - 1);
            src = shared + (size_t)id * D;
        } else if (s < sp + P) {
            src = proj + ((size_t)img_index[b] * P + (s - sp)) * D;
        } else {
            int id = ids[(size_t)b * L + (s - P + 1)];
            id = min(max(id, 0), vocab - 1);
            src = shared + (size_t)id * D;
        }
    }
    for (int i = threadIdx.x; i < (D >> 2); i += 256) {
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if (src) {
            const uint2 v = reinterpret_cast<const uint2*>(src)[i];
            o.x = e_bf2f((bf16_t)(v.x & 0xffff));
            o.y = e_bf2f((bf16_t)(v.x >> 16));
            o.z = e_bf2f((bf16_t)(v.y & 0xffff));
            o.w = e_bf2f((bf16_t)(v.y >> 16));
        }
        orow[i] = o;
    }
}

hipError_t launch_embed_splice(const int* ids, const int* sent_pos, const int* enc_len, const int* img_index,
                               const bf16_t* shared, const bf16_t* proj, float* out, int B, int L, int P, int D,
                               int vocab, hipStream_t s) {
    if (D % 4) return hipErrorInvalidValue;
    hipLaunchKernelGGL(embed_splice_kernel, dim3(L - 1 + P, B), dim3(256), 0, s, ids, sent_pos, enc_len, img_index,
                       shared, proj, out, L, P, D, vocab);
    return hipGetLastError();
}

// decoder_input_ids = shift_right(labels) (HF models/t5/modeling_t5.py:618-637); h = shared[id]
__global__ void __launch_bounds__(256) decoder_embed_kernel(const int* __restrict__ labels, int ld_labels,
                                                            const bf16_t* __restrict__ shared, float* __restrict__ out,
                                                            int T, int D, int vocab, int pos0) {
    const int b = blockIdx.y, t = blockIdx.x;
    int id = 0;                                          // decoder_start_token_id = pad = 0
    if (pos0 + t > 0) {
        id = labels[(size_t)b * ld_labels + pos0 + t - 1];
        if (id == -100) id = 0;
        id = min(max(id, 0), vocab - 1);
    }
    const bf16_t* src = shared + (size_t)id * D;
    float* orow = out + ((size_t)b * T + t) * D;
    for (int i = threadIdx.x; i < D; i += 256) orow[i] = e_bf2f(src[i]);
}

hipError_t launch_decoder_embed(const int* labels, int ld_labels, const bf16_t* shared, float* out, int B, int T, int D,
                                int vocab, hipStream_t s, int pos0) {
    hipLaunchKernelGGL(decoder_embed_kernel, dim3(T, B), dim3(256), 0, s, labels, ld_labels, shared, out, T, D, vocab, pos0);
    return hipGetLastError();
}

// Fused residual + RMSNorm: per-tile partial row sums of squares [parts][M] -> 1/rms per row (index-order sum)
__global__ void __launch_bounds__(256) rowss_to_rs_kernel(const float* __restrict__ rowss, int parts, int M, float invd,
                                                          float eps, float* __restrict__ rs) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= M) return;
    float ss = 0.0f;
    for (int q = 0; q < parts; ++q) ss += rowss[(size_t)q * M + i];
    rs[i] = rsqrtf(ss * invd + eps);
}

hipError_t launch_rowss_to_rs(const float* rowss, int parts, int M, float invd, float eps, float* rs, hipStream_t s) {
    hipLaunchKernelGGL(rowss_to_rs_kernel, dim3((M + 255) / 256), dim3(256), 0, s, rowss, parts, M, invd, eps, rs);
    return hipGetLastError();
}

// Rotary position embedding, rotate-half convention, in place on a head-major tensor x [B, H, S, hd] bf16
// (HF models/qwen2_5_vl/modeling_qwen2_5_vl.py:153-172 vision, :557-599 multimodal sections; fp32 math, cast back):
//   x[i] <- x[i]*cos[i] - x[i+half]*sin[i],  x[i+half] <- x[i+half]*cos[i] + x[i]*sin[i],  i < half
// cos/sin: fp32 [B*S, half], already section-selected per token (the tables depend on the batch, not on the layer).
// Dims >= 2*half (the zero padding of 80-wide heads) are untouched.  One thread per (token, head, pair of i).
__global__ void __launch_bounds__(256) rope_kernel(bf16_t* __restrict__ x, const float* __restrict__ cs,
                                                   const float* __restrict__ sn, int B, int H, int S, int hd, int half) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int hp = half >> 1;                                  // two i per thread (4-byte accesses)
    const size_t total = (size_t)B * H * S * hp;
    if (idx >= total) return;
    const int ip = (int)(idx % hp);
    const size_t row = idx / hp;                               // (b*H + h)*S + s
    const int s_ = (int)(row % S);
    const int b = (int)(row / ((size_t)H * S));
    const size_t tok = (size_t)b * S + s_;
    bf16_t* xr = x + row * hd;
    const uint32_t lo = *reinterpret_cast<const uint32_t*>(xr + 2 * ip);
    const uint32_t hi = *reinterpret_cast<const uint32_t*>(xr + half + 2 * ip);
    const float2 c = *reinterpret_cast<const float2*>(cs + tok * half + 2 * ip);
    const float2 sv = *reinterpret_cast<const float2*>(sn + tok * half + 2 * ip);
    const float a0 = e_bf2f((bf16_t)(lo & 0xffff)), a1 = e_bf2f((bf16_t)(lo >> 16));
    const float b0 = e_bf2f((bf16_t)(hi & 0xffff)), b1 = e_bf2f((bf16_t)(hi >> 16));
    *reinterpret_cast<uint32_t*>(xr + 2 * ip) = e_pack2_hw(a0 * c.x - b0 * sv.x, a1 * c.y - b1 * sv.y);
    *reinterpret_cast<uint32_t*>(xr + half + 2 * ip) = e_pack2_hw(b0 * c.x + a0 * sv.x, b1 * c.y + a1 * sv.y);
}

// Same operation, indices from the launch geometry instead of run-time 64-bit divisions (the flat kernel spends 516
// instructions per thread, 257 of them VALU, on 16 bytes of tensor traffic and is issue-bound): block (32, 8) =
// (pair index ip, position within an 8-row slab), grid (ceil(S / 8), H, B).
__global__ void __launch_bounds__(256) rope_grid_kernel(bf16_t* __restrict__ x, const float* __restrict__ cs,
                                                        const float* __restrict__ sn, int H, int S, int hd, int half) {
    const int ip = threadIdx.x, s_ = blockIdx.x * 8 + threadIdx.y;
    if (ip >= (half >> 1) || s_ >= S) return;
    const int h = blockIdx.y, b = blockIdx.z;
    const size_t tok = (size_t)b * S + s_;
    bf16_t* xr = x + (((size_t)b * H + h) * S + s_) * hd;
    const uint32_t lo = *reinterpret_cast<const uint32_t*>(xr + 2 * ip);
    const uint32_t hi = *reinterpret_cast<const uint32_t*>(xr + half + 2 * ip);
    const float2 c = *reinterpret_cast<const float2*>(cs + tok * half + 2 * ip);
    const float2 sv = *reinterpret_cast<const float2*>(sn + tok * half + 2 * ip);
    const float a0 = e_bf2f((bf16_t)(lo & 0xffff)), a1 = e_bf2f((bf16_t)(lo >> 16));
    const float b0 = e_bf2f((bf16_t)(hi & 0xffff)), b1 = e_bf2f((bf16_t)(hi >> 16));
    *reinterpret_cast<uint32_t*>(xr + 2 * ip) = e_pack2_hw(a0 * c.x - b0 * sv.x, a1 * c.y - b1 * sv.y);
    *reinterpret_cast<uint32_t*>(xr + half + 2 * ip) = e_pack2_hw(b0 * c.x + a0 * sv.x, b1 * c.y + a1 * sv.y);
}

// Q and K of one layer in ONE launch, 16-byte accesses: thread = (8-lane chunk c of the first half, position), it rotates lanes
// [8c, 8c+8) against [half + 8c, half + 8c + 8) -- same fp32 expressions as rope_kernel, element for element.  blockIdx.y walks the
// Hq query heads, then the Hk key heads.  The 4-byte kernels above move 16 B per thread and are issue-bound at half the HBM rate.
// F16: q / k are IEEE fp16 tensors (the Qwen2.5-VL row's fp16 forms; a power-of-two scale on them passes through: the rotation is linear)
template <bool F16>
__global__ void __launch_bounds__(256) rope_qk_kernel(bf16_t* __restrict__ q, bf16_t* __restrict__ k, const float* __restrict__ cs,
                                                      const float* __restrict__ sn, int Hq, int Hk, int S, int hd, int half) {
    const int c = threadIdx.x, s_ = blockIdx.x * 32 + threadIdx.y;
    if (c >= (half >> 3) || s_ >= S) return;
    const int hy = blockIdx.y, b = blockIdx.z;
    const bool isq = hy < Hq;
    const int h = isq ? hy : hy - Hq, H = isq ? Hq : Hk;
    bf16_t* xr = (isq ? q : k) + (((size_t)b * H + h) * S + s_) * hd;
    const size_t tok = (size_t)b * S + s_;
    const uint4 lo = *reinterpret_cast<const uint4*>(xr + 8 * c);
    const uint4 hi = *reinterpret_cast<const uint4*>(xr + half + 8 * c);
    const float4 c0 = *reinterpret_cast<const float4*>(cs + tok * half + 8 * c), c1 = *reinterpret_cast<const float4*>(cs + tok * half + 8 * c + 4);
    const float4 s0 = *reinterpret_cast<const float4*>(sn + tok * half + 8 * c), s1 = *reinterpret_cast<const float4*>(sn + tok * half + 8 * c + 4);
    const uint32_t lw[4] = {lo.x, lo.y, lo.z, lo.w}, hw[4] = {hi.x, hi.y, hi.z, hi.w};
    const float cc[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w}, ss[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
    uint32_t ol[4], oh[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float a0, a1, b0, b1;
        if constexpr (F16) {
            const e_f16x2 la = __builtin_bit_cast(e_f16x2, lw[j]), hb = __builtin_bit_cast(e_f16x2, hw[j]);
            a0 = (float)la[0]; a1 = (float)la[1]; b0 = (float)hb[0]; b1 = (float)hb[1];
        } else {
            a0 = e_bf2f((bf16_t)(lw[j] & 0xffff)); a1 = e_bf2f((bf16_t)(lw[j] >> 16));
            b0 = e_bf2f((bf16_t)(hw[j] & 0xffff)); b1 = e_bf2f((bf16_t)(hw[j] >> 16));
        }
        ol[j] = e_pack2_t<F16>(a0 * cc[2 * j] - b0 * ss[2 * j], a1 * cc[2 * j + 1] - b1 * ss[2 * j + 1]);
        oh[j] = e_pack2_t<F16>(b0 * cc[2 * j] + a0 * ss[2 * j], b1 * cc[2 * j + 1] + a1 * ss[2 * j + 1]);
    }
    *reinterpret_cast<uint4*>(xr + 8 * c) = make_uint4(ol[0], ol[1], ol[2], ol[3]);
    *reinterpret_cast<uint4*>(xr + half + 8 * c) = make_uint4(oh[0], oh[1], oh[2], oh[3]);
}

hipError_t launch_rope(bf16_t* x, const float* cs, const float* sn, int B, int H, int S, int hd, int half, hipStream_t s);
hipError_t launch_rope_qk(bf16_t* q, bf16_t* k, const float* cs, const float* sn, int B, int Hq, int Hk, int S, int hd, int half,
                          hipStream_t s, bool f16) {
    if ((half & 1) || 2 * half > hd || (hd & 1)) return hipErrorInvalidValue;
    if ((half & 7) == 0 && (hd & 7) == 0 && half <= 64 && B <= 65535 && Hq + Hk <= 65535) {
        const dim3 grid((unsigned)((S + 31) / 32), (unsigned)(Hq + Hk), (unsigned)B), block(8, 32);
        if (f16) hipLaunchKernelGGL(rope_qk_kernel<true>, grid, block, 0, s, q, k, cs, sn, Hq, Hk, S, hd, half);
        else hipLaunchKernelGGL(rope_qk_kernel<false>, grid, block, 0, s, q, k, cs, sn, Hq, Hk, S, hd, half);
        return hipGetLastError();
    }
    if (f16) return hipErrorInvalidValue;      // the fp16 forms exist for 16-byte head halves only (vqs_qwen.cpp checks at create)
    const hipError_t e = launch_rope(q, cs, sn, B, Hq, S, hd, half, s);
    return e != hipSuccess ? e : launch_rope(k, cs, sn, B, Hk, S, hd, half, s);
}

hipError_t launch_rope(bf16_t* x, const float* cs, const float* sn, int B, int H, int S, int hd, int half, hipStream_t s) {
    if ((half & 1) || 2 * half > hd || (hd & 1)) return hipErrorInvalidValue;
    if (half <= 64 && B <= 65535 && H <= 65535) {
        hipLaunchKernelGGL(rope_grid_kernel, dim3((unsigned)((S + 7) / 8), (unsigned)H, (unsigned)B), dim3(32, 8), 0, s, x, cs, sn, H, S,
                           hd, half);
        return hipGetLastError();
    }
    const size_t total = (size_t)B * H * S * (half >> 1);
    hipLaunchKernelGGL(rope_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, cs, sn, B, H, S, hd, half);
    return hipGetLastError();
}

// Split-K epilogue: out[i] = bf16(sum_s part[s][i]), slices summed in index order (deterministic)
__global__ void __launch_bounds__(256) reduce_slices_kernel(const float* __restrict__ part, int nslices, size_t n4,
                                                            uint2* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const float4* p = reinterpret_cast<const float4*>(part) + i;
    float4 a = p[0];
    for (int s = 1; s < nslices; ++s) {
        const float4 b = p[(size_t)s * n4];
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    uint2 o;
    o.x = e_pack2_hw(a.x, a.y);
    o.y = e_pack2_hw(a.z, a.w);
    out[i] = o;
}

hipError_t launch_reduce_slices(const float* part, int nslices, size_t n, bf16_t* out, hipStream_t s) {
    if (n % 4) return hipErrorInvalidValue;
    const size_t n4 = n / 4;
    hipLaunchKernelGGL(reduce_slices_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, part, nslices, n4,
                       reinterpret_cast<uint2*>(out));
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Precise decoder glue (round 4).  The error attribution (profiles/r4_error_attribution.md) puts most of the path's
// end-to-end |delta log P| into the bf16 roundings of the T-row DECODER (its rows feed the head directly; the encoder's
// errors average out over ~600 keys), so those tensors are split-bf16 (two planes hi + lo = 16 significant bits, consumed by
// the bf16 MFMA as two stacked row blocks) or stay fp32.  M = B*T rows: none of this is bandwidth that matters.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void e_split2(float a, float b, uint32_t& hi, uint32_t& lo) {
    hi = e_pack2_hw(a, b);
    lo = e_pack2_hw(a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xffff0000u));
}
__device__ __forceinline__ void e_split4_store(bf16_t* hi_p, bf16_t* lo_p, float4 v) {     // 4 consecutive elements, 8-byte aligned
    uint2 h, l;
    e_split2(v.x, v.y, h.x, l.x);
    e_split2(v.z, v.w, h.y, l.y);
    *reinterpret_cast<uint2*>(hi_p) = h;
    *reinterpret_cast<uint2*>(lo_p) = l;
}

// One wave per row, any D % 4 == 0.  x += delta (fp32, stored) when delta != nullptr; statistics and scaling in the order of
// norm_rows_reg_kernel (per-lane partial sums of squares, wave reduction, (x * rs) * w).
__global__ void __launch_bounds__(256) rmsnorm_split_kernel(float* __restrict__ x, const float* __restrict__ delta,
                                                            const bf16_t* __restrict__ w, bf16_t* __restrict__ out, long long out_plane,
                                                            int M, int D, float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= M) return;
    float4* xr = reinterpret_cast<float4*>(x + (size_t)row * D);
    const float4* dr = delta ? reinterpret_cast<const float4*>(delta + (size_t)row * D) : nullptr;
    const int n4 = D >> 2;
    float s2 = 0.0f;
    for (int i = lane; i < n4; i += 64) {
        float4 v = xr[i];
        if (dr) {
            const float4 d = dr[i];
            v.x += d.x; v.y += d.y; v.z += d.z; v.w += d.w;
            xr[i] = v;
        }
        s2 += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    const float rs = rsqrtf(wave_sum(s2) * (1.0f / (float)D) + eps);
    const uint2* wr = reinterpret_cast<const uint2*>(w);
    bf16_t* oh = out + (size_t)row * D;
    for (int i = lane; i < n4; i += 64) {
        const float4 v = xr[i];                      // this lane's own store above (same thread: ordered)
        const float4 wv = bf4_to_f4(wr[i]);
        e_split4_store(oh + 4 * i, oh + out_plane + 4 * i, make_float4(v.x * rs * wv.x, v.y * rs * wv.y, v.z * rs * wv.z, v.w * rs * wv.w));
    }
}

hipError_t launch_rmsnorm_split(float* x, const float* delta, const bf16_t* w, bf16_t* out, long long out_plane, int M, int D,
                                float eps, hipStream_t s) {
    if (M <= 0 || D <= 0 || (D % 4) != 0 || (out_plane % 4) != 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(rmsnorm_split_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, s, x, delta, w, out, out_plane, M, D, eps);
    return hipGetLastError();
}

__device__ __forceinline__ float e_gelu_new(float x) {      // gemm.hip act_gelu_new: x * sigmoid(2u), hardware reciprocal
    const float u2 = 1.5957691216057308f * (x + 0.044715f * x * x * x);   // 2u
    return x * __builtin_amdgcn_rcpf(1.0f + __expf(-u2));
}

// SILU_BIAS (the Qwen2.5-VL precise tail, vqs_qwen.cpp): a bf16 bias [cols] is added to the gathered sums (nullptr = none) and the gated
// form applies SiLU(gate) * up instead of gelu_new(gate) * linear -- the arithmetic of gemm.hip's EPI_GATED with gate_act = 1.
template <int MODE, bool SILU_BIAS = false>
__global__ void __launch_bounds__(256) sum_planes_kernel(const float* __restrict__ part, int nslices, long long slice_stride, int rows,
                                                         int cols4, int ldp, void* __restrict__ out, int ld_out, long long out_plane,
                                                         const bf16_t* __restrict__ bias = nullptr) {
    // thread -> (row, group of 4 output columns); cols4 = output columns / 4
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)rows * cols4) return;
    const int r = (int)(i / cols4), c4 = (int)(i - (long long)r * cols4);
    auto gather = [&](int col) {                      // (sum over slices of the hi rows) + (sum over slices of the lo rows)
        const float* ph = part + (size_t)r * ldp + col;
        const float* pl = part + (size_t)(rows + r) * ldp + col;
        float4 a = *reinterpret_cast<const float4*>(ph);
        float4 b = *reinterpret_cast<const float4*>(pl);
        for (int k = 1; k < nslices; ++k) {
            const float4 u = *reinterpret_cast<const float4*>(ph + (size_t)k * slice_stride);
            const float4 v = *reinterpret_cast<const float4*>(pl + (size_t)k * slice_stride);
            a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
            b.x += v.x; b.y += v.y; b.z += v.z; b.w += v.w;
        }
        float4 y = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
        if (SILU_BIAS && bias != nullptr) {
            const float4 bb = bf4_to_f4(*reinterpret_cast<const uint2*>(bias + col));
            y.x += bb.x; y.y += bb.y; y.z += bb.z; y.w += bb.w;
        }
        return y;
    };
    if (MODE == SUM_GATED_SPLIT) {
        const int oc = c4 * 4;                        // output column; packed order: block of 64 = 32 gate | 32 linear columns
        const int gc = (oc >> 5) * 64 + (oc & 31);
        const float4 g = gather(gc), l = gather(gc + 32);
        bf16_t* oh = reinterpret_cast<bf16_t*>(out) + (size_t)r * ld_out + oc;
        auto act = [](float x) {
            if (SILU_BIAS) return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x));
            return e_gelu_new(x);
        };
        e_split4_store(oh, oh + out_plane, make_float4(act(g.x) * l.x, act(g.y) * l.y, act(g.z) * l.z, act(g.w) * l.w));
    } else {
        const float4 y = gather(c4 * 4);
        if (MODE == SUM_F32) {
            *reinterpret_cast<float4*>(reinterpret_cast<float*>(out) + (size_t)r * ld_out + c4 * 4) = y;
        } else if (MODE == SUM_F16) {
            uint2 o;
            o.x = e_pack2_h(y.x, y.y);
            o.y = e_pack2_h(y.z, y.w);
            *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(out) + (size_t)r * ld_out + c4 * 4) = o;
        } else {
            bf16_t* oh = reinterpret_cast<bf16_t*>(out) + (size_t)r * ld_out + c4 * 4;
            e_split4_store(oh, oh + out_plane, y);
        }
    }
}

hipError_t launch_sum_planes(const float* part, int nslices, long long slice_stride, int rows, int cols, int ldp, int mode, void* out,
                             int ld_out, long long out_plane, hipStream_t s, const bf16_t* bias, int silu) {
    if (rows <= 0 || cols <= 0 || nslices <= 0 || (ldp % 4) != 0 || (ld_out % 4) != 0 || (out_plane % 4) != 0 || (slice_stride % 4) != 0)
        return hipErrorInvalidValue;
    const int out_cols = mode == SUM_GATED_SPLIT ? cols / 2 : cols;
    if ((out_cols % 4) != 0 || (mode == SUM_GATED_SPLIT && (cols % 64) != 0)) return hipErrorInvalidValue;
    const int cols4 = out_cols / 4;
    const long long n = (long long)rows * cols4;
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    if (bias != nullptr || silu) {        // the Qwen2.5-VL tail's forms: bias of the qkv projection, SiLU gating
        if (mode == SUM_F32) hipLaunchKernelGGL((sum_planes_kernel<SUM_F32, true>), grid, block, 0, s, part, nslices, slice_stride, rows, cols4, ldp, out, ld_out, out_plane, bias);
        else if (mode == SUM_SPLIT) hipLaunchKernelGGL((sum_planes_kernel<SUM_SPLIT, true>), grid, block, 0, s, part, nslices, slice_stride, rows, cols4, ldp, out, ld_out, out_plane, bias);
        else if (mode == SUM_GATED_SPLIT) hipLaunchKernelGGL((sum_planes_kernel<SUM_GATED_SPLIT, true>), grid, block, 0, s, part, nslices, slice_stride, rows, cols4, ldp, out, ld_out, out_plane, bias);
        else return hipErrorInvalidValue;
        return hipGetLastError();
    }
    if (mode == SUM_F32) hipLaunchKernelGGL((sum_planes_kernel<SUM_F32>), grid, block, 0, s, part, nslices, slice_stride, rows, cols4, ldp, out, ld_out, out_plane);
    else if (mode == SUM_F16) hipLaunchKernelGGL((sum_planes_kernel<SUM_F16>), grid, block, 0, s, part, nslices, slice_stride, rows, cols4, ldp, out, ld_out, out_plane);
    else if (mode == SUM_SPLIT) hipLaunchKernelGGL((sum_planes_kernel<SUM_SPLIT>), grid, block, 0, s, part, nslices, slice_stride, rows, cols4, ldp, out, ld_out, out_plane);
    else if (mode == SUM_GATED_SPLIT) hipLaunchKernelGGL((sum_planes_kernel<SUM_GATED_SPLIT>), grid, block, 0, s, part, nslices, slice_stride, rows, cols4, ldp, out, ld_out, out_plane);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

// Split-K epilogue with the GEMM epilogues' arithmetic, for launches that run as fp32 partials (the Qwen2.5-VL decode step's skinny
// nn.Linears on the stream form): out bf16 [rows, ld_out] = f(sum_k part[k][r][c] (+ bias[c])), slices summed in index order.
// gated = 0: f = identity.  gated = 1: the N = 2 F' columns are in the packed gate|up order (blocks of 64 = 32 gate | 32 up columns)
// and out[r][32 j + w] = SiLU(gate) * up (gemm.hip EPI_GATED, gate_act = 1: x * rcp(1 + exp(-x)), hardware reciprocal).
template <int GATED>
__global__ void __launch_bounds__(256) reduce_slices_act_kernel(const float* __restrict__ part, int nslices, long long slice_stride, int rows,
                                                                int cols4, int ldp, const bf16_t* __restrict__ bias, bf16_t* __restrict__ out,
                                                                int ld_out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)rows * cols4) return;
    const int r = (int)(i / cols4), c4 = (int)(i - (long long)r * cols4);
    auto gather = [&](int col) {
        const float* ph = part + (size_t)r * ldp + col;
        float4 a = *reinterpret_cast<const float4*>(ph);
        for (int k = 1; k < nslices; ++k) {
            const float4 u = *reinterpret_cast<const float4*>(ph + (size_t)k * slice_stride);
            a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
        }
        if (bias != nullptr) {
            const float4 b = bf4_to_f4(*reinterpret_cast<const uint2*>(bias + col));
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        return a;
    };
    const int oc = c4 * 4;
    float4 y;
    if (GATED) {
        const int gc = (oc >> 5) * 64 + (oc & 31);
        const float4 g = gather(gc), u = gather(gc + 32);
        auto silu = [](float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); };
        y = make_float4(silu(g.x) * u.x, silu(g.y) * u.y, silu(g.z) * u.z, silu(g.w) * u.w);
    } else {
        y = gather(oc);
    }
    uint2 o;
    o.x = e_pack2_hw(y.x, y.y);
    o.y = e_pack2_hw(y.z, y.w);
    *reinterpret_cast<uint2*>(out + (size_t)r * ld_out + oc) = o;
}

hipError_t launch_reduce_slices_act(const float* part, int nslices, long long slice_stride, int rows, int cols, int ldp, const bf16_t* bias,
                                    int gated, bf16_t* out, int ld_out, hipStream_t s) {
    if (rows <= 0 || cols <= 0 || nslices <= 0 || (ldp % 4) != 0 || (ld_out % 4) != 0 || (slice_stride % 4) != 0) return hipErrorInvalidValue;
    const int out_cols = gated ? cols / 2 : cols;
    if ((out_cols % 4) != 0 || (gated && (cols % 64) != 0)) return hipErrorInvalidValue;
    const int cols4 = out_cols / 4;
    const long long n = (long long)rows * cols4;
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    if (gated) hipLaunchKernelGGL((reduce_slices_act_kernel<1>), grid, block, 0, s, part, nslices, slice_stride, rows, cols4, ldp, bias, out, ld_out);
    else hipLaunchKernelGGL((reduce_slices_act_kernel<0>), grid, block, 0, s, part, nslices, slice_stride, rows, cols4, ldp, bias, out, ld_out);
    return hipGetLastError();
}

// Greedy step of vqs_generate: tokens[b, T-1] = argmax_v logits[(b*T + T-1), v] (lowest index on ties, as torch.argmax)
__global__ void __launch_bounds__(256) argmax_append_kernel(const float* __restrict__ logits, int ldl, int V,
                                                            int* __restrict__ tokens, int ld_tokens, int T, int dst_col, int* __restrict__ flags) {
    __shared__ float s_val[256];
    __shared__ int s_idx[256];
    const int b = blockIdx.x;
    const float* row = logits + ((size_t)b * T + (T - 1)) * ldl;
    float best = -3.0e38f;
    int bi = 0;
    bool bad = false;                   // a NaN never wins a comparison: a non-finite row would otherwise emit token 0 silently
    for (int v = threadIdx.x; v < V; v += 256) {
        const float x = row[v];
        bad |= !(fabsf(x) <= 3.0e38f);
        if (x > best) { best = x; bi = v; }
    }
    if (flags != nullptr && __any(bad) && (threadIdx.x & 63) == 0) atomicOr(flags, 2);      // status bit 1, as vqs_score's head sets it
    s_val[threadIdx.x] = best;
    s_idx[threadIdx.x] = bi;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            const float o = s_val[threadIdx.x + off];
            const int oi = s_idx[threadIdx.x + off];
            if (o > s_val[threadIdx.x] || (o == s_val[threadIdx.x] && oi < s_idx[threadIdx.x])) {
                s_val[threadIdx.x] = o;
                s_idx[threadIdx.x] = oi;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) tokens[(size_t)b * ld_tokens + dst_col] = s_idx[0];
}

hipError_t launch_argmax_append(const float* logits, int ldl, int V, int* tokens, int ld_tokens, int B, int T,
                                hipStream_t s, int dst_col, int* flags) {
    hipLaunchKernelGGL(argmax_append_kernel, dim3(B), dim3(256), 0, s, logits, ldl, V, tokens, ld_tokens, T, dst_col < 0 ? T - 1 : dst_col, flags);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Relative position bias tables (HF models/t5/modeling_t5.py:264-279) from a host-computed bucket LUT
// (integer bucket function evaluated on the host with the oracle-pinned fp32 formula).
//   enc_table[h][rel + S - 1] = W[bucket_bidir(rel)][h],  rel = key - query in [-(S-1), S-1]
//   dec_table[h][dist]        = W[bucket_causal(-dist)][h], dist = query - key in [0, T-1]
// lut index = min(|rel|, lut_len-1); bidirectional adds buckets/2 for rel > 0.
// ------------------------------------------------------------------------------------------------
__global__ void relpos_table_kernel(const bf16_t* __restrict__ rel_w, const int* __restrict__ lut_bidir,
                                    const int* __restrict__ lut_causal, int lut_len, int buckets,
                                    float* __restrict__ enc_table, int H, int S, float* __restrict__ dec_table, int T) {
    const int h = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int n_enc = 2 * S - 1;
    if (enc_table && i < n_enc) {
        const int rel = i - (S - 1);
        const int a = min(abs(rel), lut_len - 1);
        const int bucket = lut_bidir[a] + (rel > 0 ? buckets / 2 : 0);
        enc_table[(size_t)h * n_enc + i] = e_bf2f(rel_w[(size_t)bucket * H + h]);
    }
    if (dec_table && i < T) {
        const int a = min(i, lut_len - 1);
        dec_table[(size_t)h * T + i] = e_bf2f(rel_w[(size_t)lut_causal[a] * H + h]);
    }
}

// two launches (encoder table / decoder table) share this entry point: pass nullptr for the one not wanted
hipError_t launch_relpos_table(const bf16_t* rel_weight, const int* bucket_lut_bidir, const int* bucket_lut_causal,
                               int lut_len, int buckets, float* enc_table, int H, int S, float* dec_table, int T,
                               hipStream_t s) {
    const int n = enc_table ? 2 * S - 1 : T;
    hipLaunchKernelGGL(relpos_table_kernel, dim3((n + 255) / 256, H), dim3(256), 0, s, rel_weight, bucket_lut_bidir,
                       bucket_lut_causal, lut_len, buckets, enc_table, H, S, dec_table, T);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Scoring tail: fp32 log-softmax over the vocabulary + label gather; score = exp(mean over valid labels)
// (CrossEntropyLoss(reduction='mean', ignore_index=-100) then exp(-loss): SURVEY.md §8a row a21).
// One workgroup per (b,t) row; a second tiny kernel folds T positions.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) logprob_kernel(const float* __restrict__ logits, int ldl, int V,
                                                      const int* __restrict__ labels, float* __restrict__ lp) {
    __shared__ float red[4];
    const int row = blockIdx.x;
    const float* lr = logits + (size_t)row * ldl;
    const int tid = threadIdx.x;
    float mx = -3.0e38f;
    for (int i = tid; i < V; i += 256) mx = fmaxf(mx, lr[i]);
    mx = wave_max(mx);
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float sm = 0.0f;
    for (int i = tid; i < V; i += 256) sm += expf(lr[i] - mx);
    sm = wave_sum(sm);
    if ((tid & 63) == 0) red[tid >> 6] = sm;
    __syncthreads();
    if (tid == 0) {
        const float tot = red[0] + red[1] + red[2] + red[3];
        const int lab = labels[row];
        lp[row] = (lab < 0 || lab >= V) ? 0.0f : (lr[lab] - mx - logf(tot));
    }
}

// flags (optional, the pass's status word): bit 1 is set when a pair's label log-probs are not finite.  With the fp16 options (vit_fp16 /
// enc_fp16) an activation beyond the fp16 range is stored as +-inf by v_cvt_pk_f16_f32; it reaches this kernel as NaN logits whatever
// stage it came from (an inf operand makes the next norm's rsqrt(mean x^2) zero: inf * 0), so ONE check here guards every fp16 store
// of the pass without touching a hot kernel.
__global__ void score_fold_kernel(const float* __restrict__ lp, const int* __restrict__ labels,
                                  float* __restrict__ scores, int B, int T, int* __restrict__ flags) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float s = 0.0f;
    int n = 0;
    for (int t = 0; t < T; ++t)
        if (labels[(size_t)b * T + t] != -100) { s += lp[(size_t)b * T + t]; ++n; }
    scores[b] = expf(s / (float)max(n, 1));
    if (flags != nullptr && !(fabsf(s) <= 3.0e38f)) atomicOr(flags, 2);
}

hipError_t launch_score_head(const float* logits, int ldl, int V, const int* labels, float* label_logprobs,
                             float* scores, int B, int T, hipStream_t s, int* flags) {
    hipLaunchKernelGGL(logprob_kernel, dim3(B * T), dim3(256), 0, s, logits, ldl, V, labels, label_logprobs);
    hipLaunchKernelGGL(score_fold_kernel, dim3((B + 63) / 64), dim3(64), 0, s, label_logprobs, labels, scores, B, T, flags);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Decoder cross-attention through the encoder output (see vqs_api.cpp "reassociated cross-attention"):
//   enc_out [B,S,D] -> enc_outT [B,D,S_pad] (zero padded keys), once per call, shared by all decoder layers;
//   masked softmax of the [B, rows, S_pad] fp32 scores into bf16 probabilities (0 for keys >= enc_len[b]).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) transpose_pad_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out, int S,
                                                            int D, int S_pad) {
    __shared__ bf16_t tile[64][66];
    const int b = blockIdx.z, s0 = blockIdx.x * 64, d0 = blockIdx.y * 64;
    const bf16_t* ib = in + (size_t)b * S * D;
    bf16_t* ob = out + (size_t)b * D * S_pad;
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int r = i >> 6, c = i & 63;             // r: key within tile, c: feature
        tile[r][c] = (s0 + r < S) ? ib[(size_t)(s0 + r) * D + d0 + c] : (bf16_t)0;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int r = i >> 6, c = i & 63;             // r: feature, c: key
        ob[(size_t)(d0 + r) * S_pad + s0 + c] = tile[c][r];
    }
}

hipError_t launch_transpose_pad(const bf16_t* in, bf16_t* out, int B, int S, int D, int S_pad, hipStream_t s) {
    if ((D % 64) || (S_pad % 64) || S_pad < S) return hipErrorInvalidValue;
    hipLaunchKernelGGL(transpose_pad_kernel, dim3(S_pad / 64, D / 64, B), dim3(256), 0, s, in, out, S, D, S_pad);
    return hipGetLastError();
}

// scores fp32 [B*rows, S_pad] -> probs bf16 [B*rows, S_pad]; one wave per row; keys >= key_len[b] get 0
template <bool F16>      // F16: the probabilities leave as IEEE fp16 (the decoder's cross-attention of option dec_fp16), else bf16
__global__ void __launch_bounds__(256) masked_softmax_kernel(const float* __restrict__ scores, bf16_t* __restrict__ probs,
                                                             const int* __restrict__ key_len, int rows, int S_pad,
                                                             int total_rows) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= total_rows) return;
    const int klen = min(key_len[row / rows], S_pad);
    const float* sr = scores + (size_t)row * S_pad;
    bf16_t* pr = probs + (size_t)row * S_pad;
    float mx = -3.0e38f;
    for (int j = lane; j < klen; j += 64) mx = fmaxf(mx, sr[j]);
    mx = wave_max(mx);
    float sm = 0.0f;
    for (int j = lane; j < klen; j += 64) sm += __expf(sr[j] - mx);
    sm = wave_sum(sm);
    const float inv = sm > 0.0f ? 1.0f / sm : 0.0f;
    for (int j = lane; j < S_pad; j += 64) {
        const float pv = j < klen ? __expf(sr[j] - mx) * inv : 0.0f;
        if (F16) pr[j] = (bf16_t)(e_pack2_h(pv, 0.0f) & 0xffff);
        else pr[j] = e_f2bf(pv);
    }
}

hipError_t launch_masked_softmax(const float* scores, bf16_t* probs, const int* key_len, int B, int rows, int S_pad,
                                 hipStream_t s, bool out_f16) {
    const int total = B * rows;
    if (out_f16) hipLaunchKernelGGL(masked_softmax_kernel<true>, dim3((total + 3) / 4), dim3(256), 0, s, scores, probs, key_len, rows, S_pad, total);
    else hipLaunchKernelGGL(masked_softmax_kernel<false>, dim3((total + 3) / 4), dim3(256), 0, s, scores, probs, key_len, rows, S_pad, total);
    return hipGetLastError();
}

// dst [cols, rows] = src [rows, cols]^T  (bind-time weight transpose; rows, cols multiples of 64)
__global__ void __launch_bounds__(256) transpose_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out, int rows,
                                                        int cols) {
    __shared__ bf16_t tile[64][66];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    for (int i = threadIdx.x; i < 64 * 64; i += 256) tile[i >> 6][i & 63] = in[(size_t)(r0 + (i >> 6)) * cols + c0 + (i & 63)];
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 64; i += 256) out[(size_t)(c0 + (i >> 6)) * rows + r0 + (i & 63)] = tile[i & 63][i >> 6];
}

hipError_t launch_transpose(const bf16_t* in, bf16_t* out, int rows, int cols, hipStream_t s) {
    if ((rows % 64) || (cols % 64)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(transpose_kernel, dim3(cols / 64, rows / 64), dim3(256), 0, s, in, out, rows, cols);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Bind-time weight packing
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) copy_rows_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst,
                                                        int cols, int src_ld, int dst_ld, int dst_row_offset) {
    const int r = blockIdx.x;
    const bf16_t* s = src + (size_t)r * src_ld;
    bf16_t* d = dst + (size_t)(dst_row_offset + r) * dst_ld;
    for (int i = threadIdx.x; i < dst_ld; i += 256) d[i] = i < cols ? s[i] : (bf16_t)0;
}

hipError_t launch_copy_rows(const bf16_t* src, bf16_t* dst, int rows, int cols, int src_ld, int dst_ld,
                            int dst_row_offset, hipStream_t s) {
    hipLaunchKernelGGL(copy_rows_kernel, dim3(rows), dim3(256), 0, s, src, dst, cols, src_ld, dst_ld, dst_row_offset);
    return hipGetLastError();
}

// dst rows [64j, 64j+32) = wi_0[32j .. 32j+32), rows [64j+32, 64j+64) = wi_1[32j .. 32j+32)
__global__ void __launch_bounds__(256) interleave_gate_kernel(const bf16_t* __restrict__ wi0,
                                                              const bf16_t* __restrict__ wi1, bf16_t* __restrict__ dst,
                                                              int D) {
    const int r = blockIdx.x;                 // 0 .. 2F-1
    const int blk = r >> 6, within = r & 63;
    const bf16_t* s = (within < 32 ? wi0 : wi1) + (size_t)(blk * 32 + (within & 31)) * D;
    bf16_t* d = dst + (size_t)r * D;
    for (int i = threadIdx.x; i < D; i += 256) d[i] = s[i];
}

hipError_t launch_interleave_gate(const bf16_t* wi0, const bf16_t* wi1, bf16_t* dst, int F, int D, hipStream_t s) {
    if (F % 32) return hipErrorInvalidValue;
    hipLaunchKernelGGL(interleave_gate_kernel, dim3(2 * F), dim3(256), 0, s, wi0, wi1, dst, D);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Row / column gathers used by the Qwen2.5-VL row (weight packing at bind time, window permutation, splice).
// ------------------------------------------------------------------------------------------------
// dst[r][0..dst_ld) = row map[r] of src0 (map >= 0), of src1 (map <= -2: row -map-2), or zeros (map == -1); columns
// beyond `cols` are zero-filled.  map == nullptr: identity.
__global__ void __launch_bounds__(256) gather_rows_bf16_kernel(const bf16_t* __restrict__ src0, const bf16_t* __restrict__ src1,
                                                               const int* __restrict__ map, bf16_t* __restrict__ dst, int cols,
                                                               int src_ld, int dst_ld) {
    const int r = blockIdx.x;
    const int m = map ? map[r] : r;
    const bf16_t* s = m >= 0 ? src0 + (size_t)m * src_ld : (m <= -2 ? src1 + (size_t)(-m - 2) * src_ld : nullptr);
    bf16_t* d = dst + (size_t)r * dst_ld;
    for (int i = threadIdx.x; i < dst_ld; i += 256) d[i] = (s != nullptr && i < cols) ? s[i] : (bf16_t)0;
}
hipError_t launch_gather_rows_bf16(const bf16_t* src0, const bf16_t* src1, const int* map, bf16_t* dst, int rows, int cols,
                                   int src_ld, int dst_ld, hipStream_t s) {
    if (rows <= 0) return hipSuccess;
    hipLaunchKernelGGL(gather_rows_bf16_kernel, dim3(rows), dim3(256), 0, s, src0, src1, map, dst, cols, src_ld, dst_ld);
    return hipGetLastError();
}

// dst[r][c] = cmap[c] >= 0 ? src[r][cmap[c]] : 0
__global__ void __launch_bounds__(256) gather_cols_bf16_kernel(const bf16_t* __restrict__ src, const int* __restrict__ cmap,
                                                               bf16_t* __restrict__ dst, int src_ld, int dst_cols) {
    const int r = blockIdx.x;
    for (int c = threadIdx.x; c < dst_cols; c += 256) {
        const int m = cmap[c];
        dst[(size_t)r * dst_cols + c] = m >= 0 ? src[(size_t)r * src_ld + m] : (bf16_t)0;
    }
}
hipError_t launch_gather_cols_bf16(const bf16_t* src, const int* cmap, bf16_t* dst, int rows, int src_ld, int dst_cols,
                                   hipStream_t s) {
    hipLaunchKernelGGL(gather_cols_bf16_kernel, dim3(rows), dim3(256), 0, s, src, cmap, dst, src_ld, dst_cols);
    return hipGetLastError();
}

// fp32 rows: dst[r] = src[map[r]]
__global__ void __launch_bounds__(256) gather_rows_f32_kernel(const float* __restrict__ src, const int* __restrict__ map,
                                                              float* __restrict__ dst, int D) {
    const int r = blockIdx.x;
    const int m = map[r];                                     // -1: zero row (padding slot of a partial window)
    const float4* s = reinterpret_cast<const float4*>(src + (size_t)max(m, 0) * D);
    float4* d = reinterpret_cast<float4*>(dst + (size_t)r * D);
    for (int i = threadIdx.x; i < (D >> 2); i += 256) d[i] = m >= 0 ? s[i] : make_float4(0.f, 0.f, 0.f, 0.f);
}
hipError_t launch_gather_rows_f32(const float* src, const int* map, float* dst, int rows, int D, hipStream_t s) {
    if (D % 4) return hipErrorInvalidValue;
    hipLaunchKernelGGL(gather_rows_f32_kernel, dim3(rows), dim3(256), 0, s, src, map, dst, D);
    return hipGetLastError();
}

// Qwen2.5-VL input embeddings (HF modeling_qwen2_5_vl.py:1205-1232): token row = embed[id], or the merged vision token
// vis_slot[row] where that is >= 0 (masked_scatter of the video/image features in order).  fp32 residual stream out.
// merged_f16: the vision tokens are an fp16 tensor held behind a power-of-two scale (the tower's fp16 forms): value = fp16 * unscale
__global__ void __launch_bounds__(256) qwen_embed_kernel(const int* __restrict__ ids, const int* __restrict__ vis_slot,
                                                         const bf16_t* __restrict__ embed, const bf16_t* __restrict__ merged,
                                                         float* __restrict__ out, int D, int vocab, int merged_f16, float unscale) {
    const int r = blockIdx.x;
    const int vs = vis_slot[r];
    const bf16_t* s = vs >= 0 ? merged + (size_t)vs * D : embed + (size_t)min(max(ids[r], 0), vocab - 1) * D;
    float* o = out + (size_t)r * D;
    if (vs >= 0 && merged_f16) {
        for (int i = threadIdx.x; i < D; i += 256) o[i] = (float)__builtin_bit_cast(_Float16, s[i]) * unscale;
        return;
    }
    for (int i = threadIdx.x; i < D; i += 256) o[i] = e_bf2f(s[i]);
}
hipError_t launch_qwen_embed(const int* ids, const int* vis_slot, const bf16_t* embed, const bf16_t* merged, float* out, int rows,
                             int D, int vocab, hipStream_t s, bool merged_f16, float merged_unscale) {
    hipLaunchKernelGGL(qwen_embed_kernel, dim3(rows), dim3(256), 0, s, ids, vis_slot, embed, merged, out, D, vocab, merged_f16 ? 1 : 0, merged_unscale);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Input pipeline tail on the device: uint8 [N, H, W, 3] (decoded, padded, resized on the host by
// PIL exactly as the reference does) -> CLIP-normalised bf16 [N, 3, H, W].  Same fp32 arithmetic as the HF processor
// (HF image_processing_backends rescale + normalize: x * (1/255), then (x - mean) / std), then the reference's
// .to(bfloat16) (mm_utils.py:228).  Moves the per-pixel float work and 3/4 of the H2D bytes off the host.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) u8_to_norm_bf16_kernel(const unsigned char* __restrict__ in, bf16_t* __restrict__ out,
                                                              int HW, float m0, float m1, float m2, float s0, float s1,
                                                              float s2) {
    const int n = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x;           // pixel index inside the image
    if (p >= HW) return;
    const unsigned char* px = in + ((size_t)n * HW + p) * 3;
    const float r = (float)px[0] * (1.0f / 255.0f), g = (float)px[1] * (1.0f / 255.0f), b = (float)px[2] * (1.0f / 255.0f);
    bf16_t* o = out + (size_t)n * 3 * HW + p;
    o[0] = e_f2bf((r - m0) / s0);
    o[HW] = e_f2bf((g - m1) / s1);
    o[2 * (size_t)HW] = e_f2bf((b - m2) / s2);
}
hipError_t launch_u8_to_norm_bf16(const unsigned char* in, bf16_t* out, int N, int H, int W, const float* mean3,
                                  const float* std3, hipStream_t s) {
    if (N <= 0 || H <= 0 || W <= 0) return hipErrorInvalidValue;
    const int HW = H * W;
    hipLaunchKernelGGL(u8_to_norm_bf16_kernel, dim3((HW + 255) / 256, N), dim3(256), 0, s, in, out, HW, mean3[0], mean3[1],
                       mean3[2], std3[0], std3[1], std3[2]);
    return hipGetLastError();
}

}  // namespace vqs
