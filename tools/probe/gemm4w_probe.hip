// gemm4w_probe.hip - go / no-go for a 4-wave variant of the 256 x 256 NT tile (round 6, config #4's large-K products).
// gemm256.h runs 8 waves (128 x 64 per wave, 240 - 254 registers) and keeps ONE k-tile of LDS-DMA in flight: a third of a workgroup's
// cycles pass at the k-tile barrier waiting for it (profiles/r06_gemm256_phases.txt), and there is neither LDS (128 of 160 KiB) nor a
// register left for a second k-tile.  Here: 4 waves, 128 x 128 per wave (256 accumulator registers, one wave per SIMD), operands staged
// through REGISTERS two k-tiles ahead (global_load_dwordx4 -> 64 registers per k-tile and lane -> ds_write_b128 under the MFMAs of the
// next iteration), 8 fragment reads per 16 MFMAs instead of 6 per 8, one LDS-only barrier per 64-deep k-tile.
// C[M, N] (bf16) = A[M, K] . B[N, K]^T, A / B bf16 row-major; M, N multiples of 256, K of 64.  Epilogue: plain 8-byte stores from the
// accumulators (transposed product: a lane owns 4 consecutive columns of a row) - good enough for a main-loop verdict at large K.
// Build: hipcc --offload-arch=gfx950 -O3 gemm4w_probe.hip -o gemm4w_probe ; ./gemm4w_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned u4;
typedef __attribute__((ext_vector_type(2))) unsigned u2;
typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
typedef __attribute__((ext_vector_type(16))) float f16v;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ unsigned pk_bf16(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}

constexpr int BM = 256, BN = 256, BK = 64, THREADS = 256;
constexpr int OPB = BM * BK * 2;                 // 32 KiB per operand and buffer
constexpr int SMEM = 4 * OPB;                    // 2 buffers x (A + B)

struct Args { const unsigned short* A; const unsigned short* B; unsigned short* C; int M, N, K; };

__global__ __launch_bounds__(THREADS, 1) void gemm4w_kernel(Args p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63, hf = lane >> 5, lq = lane & 31;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6), wm = w & 1, wn = w >> 1;
    const int tiles_n = p.N / BN, tiles = (p.M / BM) * tiles_n, nk = p.K / BK;
    // fragment row offsets: row r of an image at r * 128 B, 16-byte slot s at (s ^ (r & 7)) * 16; k-step kk, half hf -> slot 2 kk + hf
    unsigned fa[4], fb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ra = 128 * wm + 32 * i + lq, rb = 128 * wn + 32 * i + lq;
        fa[i] = (unsigned)(ra * 128 + ((hf ^ (ra & 1)) << 4) + (((ra >> 1) & 3) << 5));
        fb[i] = (unsigned)(OPB + rb * 128 + ((hf ^ (rb & 1)) << 4) + (((rb >> 1) & 3) << 5));
    }
    // with slot = 2 kk + hf and swizzle s ^ (r & 7): ((2 kk + hf) ^ (r & 7)) * 16 = ((kk ^ ((r >> 1) & 3)) << 5) + ((hf ^ (r & 1)) << 4):
    // the address of k-step kk is base ^ (kk << 5) - one register per fragment row
    // staging: slot index s = t + 256 i (i < 8): row = s >> 3, p = s & 7 -> global row + p * 16 B, LDS row * 128 + (p ^ (row & 7)) * 16
    // (row of slot index t + 256 i = (t >> 3) + 32 i: the lane's eight rows differ by uniform amounts - ONE offset register per operand,
    // the rest rides in the instruction's scalar / immediate offsets)
    const int srow = t >> 3, spp = t & 7;
    const unsigned st_lds = (unsigned)(srow * 128 + ((spp ^ (srow & 7)) << 4));
    const unsigned st_g = (unsigned)((srow * p.K + spp * 8) * 2);
    typedef __amdgpu_buffer_rsrc_t rsrc_t;
    auto mk = [&](const void* ptr, unsigned bytes) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(ptr), (short)0, (int)bytes, 0x00020000); };
    const rsrc_t ra_ = mk(p.A, (unsigned)((long)p.M * p.K * 2)), rb_ = mk(p.B, (unsigned)((long)p.N * p.K * 2));
    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
        const unsigned ga0 = (unsigned)((long)m0 * p.K * 2), gb0 = (unsigned)((long)n0 * p.K * 2), rstep = (unsigned)(32 * p.K * 2);
        f16v acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        u4 sa[2][8], sb[2][8];                    // two k-tiles of operands in registers
        auto gload = [&](int kt, auto SET) {
            constexpr int set = decltype(SET)::value;
#pragma unroll
            for (int i = 0; i < 8; ++i)
                sa[set][i] = __builtin_bit_cast(u4, __builtin_amdgcn_raw_buffer_load_b128(ra_, (int)st_g, (int)(ga0 + i * rstep + kt * (BK * 2)), 0));
#pragma unroll
            for (int i = 0; i < 8; ++i)
                sb[set][i] = __builtin_bit_cast(u4, __builtin_amdgcn_raw_buffer_load_b128(rb_, (int)st_g, (int)(gb0 + i * rstep + kt * (BK * 2)), 0));
        };
        auto stage = [&](int buf, auto SET, int part) {          // part 0 / 1: the A half / the B half of a k-tile
            constexpr int set = decltype(SET)::value;
            char* base = smem + buf * (2 * OPB) + st_lds;
            if (part == 0) {
#pragma unroll
                for (int i = 0; i < 8; ++i) *reinterpret_cast<u4*>(base + i * 4096) = sa[set][i];
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) *reinterpret_cast<u4*>(base + OPB + i * 4096) = sb[set][i];
            }
        };
        using S0 = std::integral_constant<int, 0>;
        using S1 = std::integral_constant<int, 1>;
        // prologue: k-tile 0 -> LDS buffer 0, k-tile 1 in registers (set 1)
        __syncthreads();                                         // the previous tile's readers are done with both buffers
        gload(0, S0{});
        if (nk > 1) gload(1, S1{});
        stage(0, S0{}, 0);
        stage(0, S0{}, 1);
        __builtin_amdgcn_s_waitcnt(0xc07f);                      // lgkmcnt(0)
        __builtin_amdgcn_s_barrier();
        auto ktile = [&](int kt, auto CUR) {                     // CUR = parity of kt: set CUR is free (k-tile kt was staged), set 1 - CUR holds kt + 1
            constexpr int cur = decltype(CUR)::value;
            using NXT = std::integral_constant<int, 1 - cur>;
            if (kt + 2 < nk) gload(kt + 2, CUR);
            const unsigned bufoff = (unsigned)(kt & 1) * (2 * OPB);
            bf8 a[2][4], b[2][4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a[0][i] = *reinterpret_cast<const bf8*>(smem + bufoff + fa[i]);
                b[0][i] = *reinterpret_cast<const bf8*>(smem + bufoff + fb[i]);
            }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int c = kk & 1, n = c ^ 1;
                if (kk < 3) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        a[n][i] = *reinterpret_cast<const bf8*>(smem + bufoff + (fa[i] ^ (unsigned)((kk + 1) << 5)));
                        b[n][i] = *reinterpret_cast<const bf8*>(smem + bufoff + (fb[i] ^ (unsigned)((kk + 1) << 5)));
                    }
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[c][j], a[c][i], acc[i][j], 0, 0, 0);
                // k-tile kt + 1 (requested an iteration ago) goes into the other buffer under the MFMAs of k-steps 1 and 2
                if (kt + 1 < nk) {
                    if (kk == 1) stage((kt + 1) & 1, NXT{}, 0);
                    if (kk == 2) stage((kt + 1) & 1, NXT{}, 1);
                }
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);                  // lgkmcnt(0): my reads of this buffer and my writes of the other are done
            __builtin_amdgcn_s_barrier();
        };
        int kt = 0;
        for (; kt + 1 < nk; kt += 2) {
            ktile(kt, S0{});
            ktile(kt + 1, S1{});
        }
        if (kt < nk) ktile(kt, S0{});
        // epilogue: acc[i][j][4 g + e] = C[m0 + 128 wm + 32 i + lq][n0 + 128 wn + 32 j + 8 g + 4 hf + e]
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long row = m0 + 128 * wm + 32 * i + lq;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    u2 o;
                    o.x = pk_bf16(acc[i][j][4 * g], acc[i][j][4 * g + 1]);
                    o.y = pk_bf16(acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
                    *reinterpret_cast<u2*>(p.C + row * p.N + n0 + 128 * wn + 32 * j + 8 * g + 4 * hf) = o;
                }
        }
    }
}

static unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16); }
static float bf2f(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }

int main() {
    const int shapes[][3] = {{65536, 768, 3072}, {65536, 3072, 768}, {65536, 512, 2048}, {65536, 2048, 512}, {131072, 1536, 384}, {65536, 768, 768}};
    CK(hipFuncSetAttribute((const void*)gemm4w_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
    for (auto& s : shapes) {
        const int M = s[0], N = s[1], K = s[2];
        std::vector<unsigned short> hA((size_t)M * K), hB((size_t)N * K);
        unsigned rng = 12345u;
        auto rnd = [&]() { rng = rng * 1664525u + 1013904223u; return ((rng >> 9) & 0xffff) / 65536.0f - 0.5f; };
        for (auto& v : hA) v = f2bf(rnd());
        for (auto& v : hB) v = f2bf(rnd() * 0.1f);
        unsigned short *dA, *dB, *dC;
        CK(hipMalloc(&dA, hA.size() * 2)); CK(hipMalloc(&dB, hB.size() * 2)); CK(hipMalloc(&dC, (size_t)M * N * 2));
        CK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(dB, hB.data(), hB.size() * 2, hipMemcpyHostToDevice));
        Args a{dA, dB, dC, M, N, K};
        const int tiles = (M / BM) * (N / BN), grid = tiles < 256 ? tiles : 256;
        for (int i = 0; i < 20; ++i) gemm4w_kernel<<<grid, THREADS, SMEM>>>(a);
        CK(hipDeviceSynchronize());
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const int iters = 30;
        CK(hipEventRecord(e0));
        for (int i = 0; i < iters; ++i) gemm4w_kernel<<<grid, THREADS, SMEM>>>(a);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= iters;
        // spot check: 64 entries against a double-precision dot product of the bf16 operands
        std::vector<unsigned short> hC((size_t)M * N);
        CK(hipMemcpy(hC.data(), dC, hC.size() * 2, hipMemcpyDeviceToHost));
        double worst = 0.0;
        for (int q = 0; q < 64; ++q) {
            const int r = (int)(((long)q * 7919 * 131 + 17) % M), c = (int)(((long)q * 104729 + 5) % N);
            double ref = 0.0;
            for (int k = 0; k < K; ++k) ref += (double)bf2f(hA[(size_t)r * K + k]) * bf2f(hB[(size_t)c * K + k]);
            const double err = fabs(ref - bf2f(hC[(size_t)r * N + c])) / (fabs(ref) + 1e-2);
            worst = err > worst ? err : worst;
        }
        printf("{\"shape\": [%d, %d, %d], \"ms\": %.4f, \"tflops\": %.1f, \"worst_rel_err_of_64\": %.4f}\n", M, N, K, ms, 2.0 * M * N * K / ms / 1e9, worst);
        fflush(stdout);
        CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC));
    }
    return 0;
}
