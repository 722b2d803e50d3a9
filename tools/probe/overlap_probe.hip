// Can one MI355X run an MFMA-bound and an HBM-bound kernel AT THE SAME TIME at (close to) their stand-alone rates?  (gfx950)
// The row-owner kernels of the step (mlp_fused.h, rowgemm.h) are the SUM of an MFMA phase and a streaming phase; whether a
// design that overlaps the two can reach max(phase) instead depends on what the chip sustains when both run together (power
// budget, fabric) - measured here before such a kernel is written.
//   mfma_body:  4 waves per workgroup (one per SIMD), 8 independent 32x32x16 bf16 accumulators per wave; mode 0 = operands in
//               registers (matrix pipe 100 % busy), mode 1 = one conflict-free ds_read_b128 per MFMA (the row-owner kernels'
//               fragment traffic), small LDS and register footprint so that streaming workgroups fit on the same CU beside it
//   stream:     grid-stride copy, 4 x 16 B loads then 4 x 16 B stores per thread and trip (the yardstick: torch's relu, 6.2 TB/s)
// Each alone, then both at once on two streams (both launch orders).  Build: hipcc --offload-arch=gfx950 -O3 overlap_probe.hip -o overlap_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
typedef __attribute__((ext_vector_type(16))) float f16v;
typedef __attribute__((ext_vector_type(4))) unsigned u4;

template <int MODE>
__global__ __launch_bounds__(256) void mfma_body(int iters, float* sink) {
    __shared__ __attribute__((aligned(16))) char lds[32768];
    const int t = threadIdx.x, lane = t & 63;
    for (int i = t; i < 32768 / 4; i += 256) reinterpret_cast<unsigned*>(lds)[i] = 0x3f803f80u;
    __syncthreads();
    f16v acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    u4 a0 = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, b0 = a0;
    const unsigned addr = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)lds + (unsigned)((lane & 31) * 128 + (((lane >> 5) ^ ((lane >> 1) & 7)) * 16));
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, a0), __builtin_bit_cast(bf8, b0), acc[j], 0, 0, 0);
        } else {
            u4 fr[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fr[j]) : "v"(addr), "n"(0));
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(fr[j]) : "n"(7 - j));
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, fr[j]), __builtin_bit_cast(bf8, b0), acc[j], 0, 0, 0);
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += acc[j][0] + acc[j][15];
    if (s == 12345.f) sink[0] = s;
}

__global__ __launch_bounds__(256) void stream(const u4* __restrict__ in, u4* __restrict__ out, long n) {
    const long stride = (long)gridDim.x * 256 * 4;
    for (long i = (long)blockIdx.x * 1024 + threadIdx.x; i < n; i += stride) {
        u4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = i + 256 * k < n ? in[i + 256 * k] : u4{0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < 4; ++k) if (i + 256 * k < n) out[i + 256 * k] = v[k];
    }
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    const long n = (1L << 30) / 16;                  // 1 GiB in, 1 GiB out
    u4 *in, *out; float* sink;
    if (hipMalloc(&in, n * 16) != hipSuccess || hipMalloc(&out, n * 16) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) { printf("malloc failed\n"); return 1; }
    hipMemset(in, 1, n * 16); hipMemset(out, 0, n * 16);
    hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
    hipDeviceSynchronize();
    const int cus = 256;
    for (int mode = 0; mode < 2; ++mode) {
        for (int wgs_per_cu = 1; wgs_per_cu <= 2; ++wgs_per_cu) {
            const int iters = (mode == 0 ? 3400 : 2600) / wgs_per_cu;
            auto run_mfma = [&](hipStream_t s) {
                if (mode == 0) mfma_body<0><<<cus * wgs_per_cu, 256, 0, s>>>(iters, sink);
                else mfma_body<1><<<cus * wgs_per_cu, 256, 0, s>>>(iters, sink);
            };
            auto run_stream = [&](hipStream_t s) { stream<<<cus * 8, 256, 0, s>>>(in, out, n); };
            double t_m = 1e30, t_s = 1e30, t_b1 = 1e30, t_b2 = 1e30;
            for (int rep = 0; rep < 40; ++rep) { run_mfma(s1); run_stream(s2); }       // warm clocks
            hipDeviceSynchronize();
            for (int rep = 0; rep < 6; ++rep) {
                double t0 = now_us(); run_mfma(s1); hipDeviceSynchronize(); double t1 = now_us(); if (t1 - t0 < t_m) t_m = t1 - t0;
                t0 = now_us(); run_stream(s2); hipDeviceSynchronize(); t1 = now_us(); if (t1 - t0 < t_s) t_s = t1 - t0;
                t0 = now_us(); run_mfma(s1); run_stream(s2); hipDeviceSynchronize(); t1 = now_us(); if (t1 - t0 < t_b1) t_b1 = t1 - t0;
                t0 = now_us(); run_stream(s2); run_mfma(s1); hipDeviceSynchronize(); t1 = now_us(); if (t1 - t0 < t_b2) t_b2 = t1 - t0;
            }
            // longer trains (10 of each, alternating launches) amortise the launch and sync cost of the wall clock
            auto train = [&](int what) {
                double best = 1e30;
                for (int rep = 0; rep < 3; ++rep) {
                    const double t0 = now_us();
                    for (int k = 0; k < 10; ++k) { if (what & 1) run_mfma(s1); if (what & 2) run_stream(s2); }
                    hipDeviceSynchronize();
                    const double t1 = now_us();
                    if (t1 - t0 < best) best = t1 - t0;
                }
                return best / 10;
            };
            const double tr_m = train(1), tr_s = train(2), tr_b = train(3);
            const double flops = 2.0 * 32 * 32 * 16 * 8.0 * iters * 4 * cus * wgs_per_cu, bytes = 2.0 * n * 16;
            printf("{\"mode\": %d, \"wgs_per_cu\": %d, \"single_us\": {\"mfma\": %.1f, \"stream\": %.1f, \"both_mfma_first\": %.1f, \"both_stream_first\": %.1f}, "
                   "\"train_us\": {\"mfma\": %.1f, \"stream\": %.1f, \"both\": %.1f, \"sum\": %.1f, \"max\": %.1f}, "
                   "\"mfma_tflops_alone\": %.0f, \"stream_tbs_alone\": %.2f, \"both_tflops\": %.0f, \"both_tbs\": %.2f}\n",
                   mode, wgs_per_cu, t_m, t_s, t_b1, t_b2, tr_m, tr_s, tr_b, tr_m + tr_s, tr_m > tr_s ? tr_m : tr_s,
                   flops / tr_m / 1e6, bytes / tr_s / 1e6, flops / tr_b / 1e6, bytes / tr_b / 1e6);
            fflush(stdout);
        }
    }
    return 0;
}
