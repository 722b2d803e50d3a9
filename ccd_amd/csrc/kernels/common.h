// common.h - small device helpers shared by every kernel header (no #includes: see prelude_hip.h).
#pragma once

namespace ccd {
typedef unsigned short bf16_t;                                    // raw bfloat16 storage
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;       // one 16-byte global/LDS transaction per lane
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((ext_vector_type(2))) unsigned short u16x2;   // packed 16-bit integer VALU (v_pk_*_u16)
typedef __attribute__((ext_vector_type(4))) float f32x4v;

__device__ __forceinline__ unsigned f2bits(float f) { unsigned u; __builtin_memcpy(&u, &f, 4); return u; }
__device__ __forceinline__ float bits2f(unsigned u) { float f; __builtin_memcpy(&f, &u, 4); return f; }
__device__ __forceinline__ float bf2f(bf16_t h) { return bits2f((unsigned)h << 16); }
// float -> bf16, round-to-nearest-even (torch's rule); cvt_bf16 / cvt_pk_bf16 come from the prelude
// (one v_cvt_pk_bf16_f32 on gfx950; bit arithmetic in the CPU SIMT executor)
__device__ __forceinline__ bf16_t f2bf(float f) { return cvt_bf16(f); }
__device__ __forceinline__ unsigned pack_bf2(float lo, float hi) { return cvt_pk_bf16(lo, hi); }
__device__ __forceinline__ float bf_lo(unsigned w) { return bits2f(w << 16); }
__device__ __forceinline__ float bf_hi(unsigned w) { return bits2f(w & 0xffff0000u); }

// IEEE binary16 -> fp32 by bit arithmetic (subnormals included): the overlay planes of the data pipeline
__device__ __forceinline__ float half2f(unsigned short h) {
    const unsigned sign = (unsigned)(h & 0x8000u) << 16, e = (h >> 10) & 31u, m = h & 0x3ffu;
    if (e == 0) return bits2f(sign) == 0.f ? (float)m * 5.9604644775390625e-8f : -((float)m * 5.9604644775390625e-8f);     // m * 2^-24
    if (e == 31) return bits2f(sign | 0x7f800000u | (m << 13));
    return bits2f(sign | ((e + 112u) << 23) | (m << 13));
}
__device__ __forceinline__ void unpack8(const u32x4& w, float* v) {
    v[0] = bf_lo(w.x); v[1] = bf_hi(w.x); v[2] = bf_lo(w.y); v[3] = bf_hi(w.y);
    v[4] = bf_lo(w.z); v[5] = bf_hi(w.z); v[6] = bf_lo(w.w); v[7] = bf_hi(w.w);
}
__device__ __forceinline__ u32x4 pack8(const float* v) {
    u32x4 w;
    w.x = pack_bf2(v[0], v[1]); w.y = pack_bf2(v[2], v[3]); w.z = pack_bf2(v[4], v[5]); w.w = pack_bf2(v[6], v[7]);
    return w;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += shfl_xor(v, m);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, shfl_xor(v, m));
    return v;
}

// erf-GELU as nn.GELU() computes it, and its derivative.  erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, far
// below the bf16 resolution of every consumer): one v_rcp + one v_exp + 7 FMA-class ops, and the Gaussian term it needs,
// exp(-x^2/2), is the same one the derivative's density uses.  (libm's erff + expf cost ~50 instructions per element
// and made the gelu'(u) epilogue VALU-bound.)
struct GeluTerms { float cdf, gauss; };                  // Phi(x), exp(-x^2/2)
__device__ __forceinline__ GeluTerms gelu_terms(float x) {
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = fast_rcp(fmaf(0.3275911f, z, 1.0f));
    const float e = fast_exp2(x * x * -0.72134752044448170f);          // exp(-x^2/2) = 2^(-x^2 / (2 ln 2))
    float poly = fmaf(t, 1.061405429f, -1.453152027f);
    poly = fmaf(t, poly, 1.421413741f);
    poly = fmaf(t, poly, -0.284496736f);
    poly = fmaf(t, poly, 0.254829592f);
    const float half_erfc = 0.5f * poly * t * e;                       // 0.5 * (1 - erf(|x| / sqrt 2))
    GeluTerms r;
    r.cdf = x >= 0.f ? 1.0f - half_erfc : half_erfc;
    r.gauss = e;
    return r;
}
// forward: libm's erff alone measured faster than the rational form (which pays a v_rcp and a v_exp); the derivative
// needs erf AND the Gaussian, where the shared form wins
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float dgelu_f(float x) {
    const GeluTerms g = gelu_terms(x);
    return fmaf(x * 0.3989422804014327f, g.gauss, g.cdf);
}

// (Round 3 measured a table-free GELU for the fused MLP - Phi(-a) = 2^P6(a), seven packed FMAs + one v_exp_f32 per element, no
// LDS access - against the LDS table of Phi: 10.66 vs 9.99 ms per step for the 24 launches.  With one wave per SIMD the VALU
// work sits in the same issue stream as the MFMAs; the table's gathers cost less than 60 more VALU instructions per chunk.)

// XCD-aware bijective remap of a linear workgroup id: consecutive work items land on the same XCD
// (hardware dispatches block b to XCD b % 8; guide T1, bijective variant for nwg % 8 != 0)
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nwg) {
    const unsigned xcd = bid & 7u, q = nwg >> 3, r = nwg & 7u;
    const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}
}  // namespace ccd
