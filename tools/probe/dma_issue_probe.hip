// What does an LDS-DMA instruction cost the wave that issues it?  (gfx950, round 4)
// The row-owner kernels run ONE wave per SIMD; with no memory traffic their MFMA loop takes 45 cycles per MFMA, with the ring's DMA
// requests (8 per 24-MFMA window and wave) 62.  This probe runs the same loop shape - 24 x (hand-issued ds_read_b128, counted wait,
// v_mfma_f32_32x32x16_bf16), one s_barrier per window - and issues the window's DMA requests (global_load_lds_dwordx4, 1 KiB each,
// source L2-resident)  mode 0: not at all,  mode 1: every wave its own D requests behind MFMA steps 1, 5, 9 ...,
// mode 2: a FIFTH wave issues all 4 D of them (the four MFMA waves issue no VMEM at all).
// Build: hipcc --offload-arch=gfx950 -O3 dma_issue_probe.hip -o dma_issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) unsigned u4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
typedef __attribute__((ext_vector_type(16))) float f16v;

__device__ __forceinline__ int swz(int row) { return (((row & 31) >> 1) ^ ((row & 31) >> 4)) & 7; }
__device__ __forceinline__ void glds16(const void* gptr, char* lds_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gptr, (__attribute__((address_space(3))) void*)lds_base, 16, 0, 0);
}

template <int MODE, int D>
__global__ __launch_bounds__(320) void probe(int iters, const char* src, unsigned long long* cycles, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int t = threadIdx.x, lane = t & 63, hf = lane >> 5, lq = lane & 31;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    for (int i = t; i < 24576 / 4; i += blockDim.x) reinterpret_cast<unsigned*>(lds)[i] = 0x3f803f80u;
    __syncthreads();
    unsigned base = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)lds;
    unsigned off[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) off[kk] = base + (unsigned)(lq * 128 + (((2 * kk + hf) ^ swz(lq)) * 16));
    char* ring = lds + 24576;                                  // 4 x 24 KiB of DMA targets, never read
    const char* my_src = src + (size_t)(blockIdx.x % 64) * 98304 + lane * 16;
    f16v acc[6];
#pragma unroll
    for (int j = 0; j < 6; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const u4 b0 = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (w < 4) {
        for (int it = 0; it < iters; ++it) {
            const int slot = it & 3;
            u4 fr[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fr[j]) : "v"(off[j / 6]), "n"((j % 6) * 4096));
#pragma unroll
            for (int k = 0; k < 24; ++k) {
                asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(fr[k % 6]) : "n"(k < 19 ? 5 : 23 - k));
                acc[k % 6] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, fr[k % 6]), __builtin_bit_cast(bf8, b0), acc[k % 6], 0, 0, 0);
                if (k + 6 < 24) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fr[k % 6]) : "v"(off[(k + 6) / 6]), "n"(((k + 6) % 6) * 4096));
                if (MODE == 1 && k % 4 == 1 && k / 4 < D) glds16(my_src + (k / 4) * 4096 + w * 1024, ring + slot * 24576 + (k / 4) * 4096 + w * 1024);
            }
            if (MODE == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * D) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
    } else {
        for (int it = 0; it < iters; ++it) {
            const int slot = it & 3;
            if (MODE == 2) {
#pragma unroll
                for (int i = 0; i < 4 * D; ++i) glds16(my_src + i * 1024, ring + slot * 24576 + i * 1024);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * D < 32 ? 4 * D : 32) : "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cycles[blockIdx.x * 8 + w] = t1 - t0;
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 6; ++j) s += acc[j][0];
    if (s == 12345.f) sink[0] = s;
}

template <int MODE, int D>
static void run(const char* name, const char* src, unsigned long long* cyc_d, float* sink) {
    const int iters = 2000, cus = 256, smem = 24576 + 4 * 24576;
    hipFuncSetAttribute((const void*)probe<MODE, D>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        probe<MODE, D><<<cus, 320, smem>>>(iters, src, cyc_d, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    // shader clock from the event time: the kernel is compute-bound, every SIMD issues 24 MFMAs per window
    printf("{\"probe\": \"%s\", \"dma_per_wave_and_window\": %d, \"ms\": %.4f, \"ns_per_mfma\": %.2f, \"cycles_per_mfma_at_2.4GHz\": %.1f}\n",
           name, D, ms, ms * 1e6 / (24.0 * iters), ms * 1e6 / (24.0 * iters) * 2.4);
    fflush(stdout);
}

int main() {
    unsigned long long* cyc; float* sink; char* src;
    hipMalloc(&cyc, 256 * 8 * 8); hipMalloc(&sink, 64); hipMalloc(&src, 64 * 98304 + 65536); hipMemset(src, 0, 64 * 98304 + 65536);
    run<0, 0>("no DMA", src, cyc, sink);
    run<1, 2>("every wave issues its own", src, cyc, sink);
    run<1, 4>("every wave issues its own", src, cyc, sink);
    run<1, 6>("every wave issues its own", src, cyc, sink);
    run<2, 2>("a fifth wave issues all", src, cyc, sink);
    run<2, 4>("a fifth wave issues all", src, cyc, sink);
    run<2, 6>("a fifth wave issues all", src, cyc, sink);
    return 0;
}
