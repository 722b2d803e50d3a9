// scratch_gap_probe.hip - does a kernel's private-segment (scratch) size cost a pipeline drain around its launch?
// The step's launch-by-launch trace shows ~6 us of idle before AND after every launch of the one kernel with ~300 B of scratch per lane
// (the fused block half, 72 - 78 spilled registers) and none around kernels with 0 - 16 B.  This probe alternates a scratch-free kernel A with a
// kernel B<N> that keeps N bytes per lane in scratch (dynamically indexed private array), both ~50 us on every CU, and reports the time per
// (A, B) pair against (A, B<0>).   hipcc --offload-arch=gfx950 -O3 scratch_gap_probe.hip -o scratch_gap_probe && ./scratch_gap_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ __launch_bounds__(256) void busy_a(float* out, int iters) {
    float v = threadIdx.x * 1e-3f;
    for (int i = 0; i < iters; ++i) v = fmaf(v, 1.0001f, 1e-4f);
    if (v == 123.456f) out[0] = v;
}

template <int N>
__global__ __launch_bounds__(256) void busy_b(float* out, int iters, int sel) {
    float v = threadIdx.x * 1e-3f;
    if constexpr (N > 0) {
        float buf[N / 4];
#pragma unroll 1
        for (int i = 0; i < N / 4; ++i) buf[i] = v + i;
#pragma unroll 1
        for (int i = 0; i < iters; ++i) {
            v = fmaf(v, 1.0001f, buf[(i + sel) % (N / 4)]);
            buf[(i * 7 + sel) % (N / 4)] = v;
        }
    } else {
        for (int i = 0; i < iters; ++i) v = fmaf(v, 1.0001f, 1e-4f);
    }
    if (v == 123.456f) out[0] = v;
}

// a second scratch kernel (another code object entry) for the "both launches keep scratch" pairs
template <int N>
__global__ __launch_bounds__(256) void busy_c(float* out, int iters, int sel) {
    float v = threadIdx.x * 2e-3f;
    float buf[N / 4];
#pragma unroll 1
    for (int i = 0; i < N / 4; ++i) buf[i] = v - i;
#pragma unroll 1
    for (int i = 0; i < iters; ++i) {
        v = fmaf(v, 0.9999f, buf[(i + sel) % (N / 4)]);
        buf[(i * 5 + sel) % (N / 4)] = v;
    }
    if (v == 123.456f) out[1] = v;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int N>
static void run(float* d, hipStream_t s, int ia, int ib) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = 256 * 4, pairs = 200;
    for (int i = 0; i < 20; ++i) { busy_a<<<grid, 256, 0, s>>>(d, ia); busy_b<N><<<grid, 256, 0, s>>>(d, ib, i); }
    CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < pairs; ++i) { busy_a<<<grid, 256, 0, s>>>(d, ia); busy_b<N><<<grid, 256, 0, s>>>(d, ib, i); }
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    // B alone, back to back
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < pairs; ++i) busy_b<N><<<grid, 256, 0, s>>>(d, ib, i);
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    float msb; CK(hipEventElapsedTime(&msb, e0, e1));
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < pairs; ++i) busy_a<<<grid, 256, 0, s>>>(d, ia);
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    float msa; CK(hipEventElapsedTime(&msa, e0, e1));
    printf("{\"scratch_bytes_per_lane\": %d, \"us_per_pair_A_B\": %.2f, \"us_A_alone\": %.2f, \"us_B_alone\": %.2f, \"pair_minus_sum_us\": %.2f}\n", N,
           1e3f * ms / pairs, 1e3f * msa / pairs, 1e3f * msb / pairs, 1e3f * (ms - msa - msb) / pairs);
    fflush(stdout);
}

template <int NB, int NC>
static void run2(float* d, hipStream_t s, int ib) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = 256 * 4, pairs = 200;
    float ms[3];
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            if (rep) CK(hipEventRecord(e0, s));
            for (int i = 0; i < (rep ? pairs : 20); ++i) {
                if (mode != 2) busy_b<NB><<<grid, 256, 0, s>>>(d, ib, i);
                if (mode != 1) busy_c<NC><<<grid, 256, 0, s>>>(d, ib, i);
            }
            if (rep) CK(hipEventRecord(e1, s));
            CK(hipStreamSynchronize(s));
        }
        CK(hipEventElapsedTime(&ms[mode], e0, e1));
    }
    printf("{\"scratch_B\": %d, \"scratch_C\": %d, \"us_per_pair_B_C\": %.2f, \"us_B_alone\": %.2f, \"us_C_alone\": %.2f, \"pair_minus_sum_us\": %.2f}\n", NB, NC,
           1e3f * ms[0] / pairs, 1e3f * ms[1] / pairs, 1e3f * ms[2] / pairs, 1e3f * (ms[0] - ms[1] - ms[2]) / pairs);
    fflush(stdout);
}

int main() {
    float* d; CK(hipMalloc(&d, 4096));
    hipStream_t s; CK(hipStreamCreate(&s));
    const int ia = 20000, ib = 2000;
    run<0>(d, s, ia, ia);
    run<16>(d, s, ia, ib);
    run<64>(d, s, ia, ib);
    run<128>(d, s, ia, ib);
    run<144>(d, s, ia, ib);
    run<160>(d, s, ia, ib);
    run<176>(d, s, ia, ib);
    run<192>(d, s, ia, ib);
    run<208>(d, s, ia, ib);
    run<224>(d, s, ia, ib);
    run<240>(d, s, ia, ib);
    run<256>(d, s, ia, ib);
    run<272>(d, s, ia, ib);
    run<304>(d, s, ia, ib);
    run<512>(d, s, ia, ib);
    run<1024>(d, s, ia, ib);
    run<0>(d, s, ia, ia);
    run2<304, 304>(d, s, ib);
    run2<304, 320>(d, s, ib);
    run2<304, 512>(d, s, ib);
    run2<304, 144>(d, s, ib);
    run2<304, 128>(d, s, ib);
    run2<304, 16>(d, s, ib);
    run2<64, 128>(d, s, ib);
    run2<304, 304>(d, s, ib);
    return 0;
}
