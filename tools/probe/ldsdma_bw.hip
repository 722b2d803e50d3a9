// How many bytes per clock does a CU move into LDS, by path?  (gfx950)
//   global_load_lds_dwordx4 (LDS-DMA) from all 8 waves, nothing else running.
// Measured (round 2): L2-resident 125-127 GB/s per CU = 60 B / cycle (32 TB/s over the chip) - one 1-KiB piece per ~17 cycles;
// from HBM 23.6 GB/s per CU = 6.0 TB/s over the chip.
// Each workgroup (512 threads, one per CU) streams its own contiguous range; `span` bytes per workgroup are re-read `reps` times
// (span small = L2 hits, span large = HBM).  Build: hipcc --offload-arch=gfx950 -O3 ldsdma_bw.hip -o ldsdma_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) unsigned u4;
constexpr int PIECES = 4;        // 1-KiB pieces per wave and step
constexpr int DEPTH = 3;         // steps in flight
__global__ __launch_bounds__(512, 1) void stream(const char* src, long span, int reps, int mode, unsigned* sink) {
    extern __shared__ char smem[];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const char* base = src + (long)blockIdx.x * span;
    const long step_bytes = 8L * PIECES * 1024;                 // per workgroup and step: 32 KiB
    const long steps = span / step_bytes;
    const bool dma = mode == 0 || (mode == 2 && w < 4);
    unsigned acc = 0;
    for (int r = 0; r < reps; ++r) {
        if (dma) {
            for (long s = 0; s < steps; ++s) {
                const char* g = base + s * step_bytes + (long)w * PIECES * 1024 + lane * 16;
                char* l = smem + ((s & 3) * 8 + w) * PIECES * 1024;
#pragma unroll
                for (int i = 0; i < PIECES; ++i)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + i * 1024),
                                                     (__attribute__((address_space(3))) void*)(l + i * 1024), 16, 0, 0);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES * (DEPTH - 1)) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    __syncthreads();
    acc += *reinterpret_cast<unsigned*>(smem + t * 4);
    if (acc == 0x12345678u) sink[0] = acc;
}
int main(int argc, char** argv) {
    int cus = 256;
    char* buf; unsigned* sink;
    const long total = 2L << 30;
    if (hipMalloc(&buf, total) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) { printf("malloc failed\n"); return 1; }
    if (hipMemset(buf, 1, total) != hipSuccess) { printf("memset failed\n"); return 1; }
    hipDeviceSynchronize(); printf("allocated\n"); fflush(stdout);
    hipFuncSetAttribute((const void*)stream, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int hbm = 0; hbm < 2; ++hbm) {
        const long span = hbm ? (total / cus) & ~32767L : 98304;      // 8 MiB per workgroup (HBM) or 96 KiB (L2-resident: 24 MiB per chip)
        const int reps = hbm ? 2 : 160;
        for (int mode = 0; mode < 1; ++mode) {
            for (int it = 0; it < 3; ++it) {
                hipEventRecord(e0);
                stream<<<cus, 512, 131072>>>(buf, span, reps, mode, sink);
                hipEventRecord(e1); hipEventSynchronize(e1);
                { hipError_t e = hipGetLastError(); if (e != hipSuccess) { printf("launch: %s\n", hipGetErrorString(e)); return 1; } }
                float ms; hipEventElapsedTime(&ms, e0, e1);
                const double bytes = (double)span * reps * cus;
                if (it == 2) { printf("%s mode %d: %.3f ms  %.2f TB/s chip  %.1f GB/s per CU  (%.1f B/clk at 2.1 GHz)\n", hbm ? "HBM" : "L2 ", mode, ms,
                                    bytes / ms / 1e9, bytes / ms / 1e6 / cus, bytes / ms / 1e6 / cus / 2.1); fflush(stdout); }
            }
        }
    }
    return 0;
}
