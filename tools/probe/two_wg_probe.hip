// Round 6, VERDICT item 1: "two INDEPENDENT wave sets per CU, out of phase" - a go / no-go model of the teacher's fused block half
// (mlp_fused_kernel<384, false, true>: 348 GF per launch, 108 weight pieces of 24 KiB per 128-row tile, a row pass that reads 48 KiB
// and writes 72 KiB per 32 rows) BEFORE anybody rewrites the kernel.  The model keeps what bounds the real kernel and nothing else:
//   * the product loop: hand-issued ds_read_b128 fragment reads 6 ahead, counted lgkmcnt waits, one MFMA per fragment, V VALU fillers
//     per MFMA (the real kernel carries 6.0 VALU + 1.7 LDS instructions per 32x32x16 MFMA: GELU, packing, ring bookkeeping);
//   * the weight ring: every wave issues its share of the next piece by LDS-DMA from an L2-resident source (counted vmcnt), one
//     LDS-only barrier per piece;
//   * the row pass: after the last piece of a tile every wave streams its rows in (16-byte loads, 8 in flight) and out (16-byte stores).
// Geometries:
//   G0  today's: ONE workgroup per CU, 4 waves x 32 rows, v_mfma_f32_32x32x16_bf16, 24-KiB pieces (24 MFMAs per piece and wave)
//   G1  the verdict's: TWO workgroups per CU, 4 waves x 16 rows each, v_mfma_f32_16x16x32_bf16, 12-KiB pieces (12 MFMAs per piece and
//       wave), <= 256 registers, 60 KiB of LDS each.  Same rows per CU, same flops, 2 x the fragment reads and 2 x the weight DMA per flop.
//       phase = 1: the second workgroup to arrive on a CU (an atomic ticket per (XCC, SE, CU)) starts half a tile out of phase - its
//       first and last tiles are half tiles - so that one workgroup's row pass runs under the other's products.
// Output: one JSON line per configuration: ms per launch, chip TFLOP/s, MFMA-busy share at the event-time clock, and (lab) the per-phase
// cycle totals of wave 0 of BOTH resident workgroups of CU 0 / XCC 0 (products, ring waits + barrier, row pass).
// Build: hipcc --offload-arch=gfx950 -O3 two_wg_probe.hip -o two_wg_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) unsigned u4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
typedef __attribute__((ext_vector_type(16))) float f16v;
typedef __attribute__((ext_vector_type(4))) float f4v;

__device__ __forceinline__ void glds16(const void* gptr, char* lds_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gptr, (__attribute__((address_space(3))) void*)lds_base, 16, 0, 0);
}
template <int V>
__device__ __forceinline__ void valu_fill(float (&f)[4], float c) {
#pragma unroll
    for (int i = 0; i < V; ++i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[i & 3]) : "v"(c));
}

struct Args {
    const char* wsrc;          // L2-resident weight bytes (2.7 MB, read over and over)
    const char* xin;           // row-pass input stream  [rows][1536 B]
    char* xout;                // row-pass output stream [rows][2304 B]
    unsigned long long* stamps;
    unsigned* tickets;         // [8 XCC][256] arrival counters per CU
    int tiles_per_wg;          // 4 at 131 072 rows
    int pieces;                // per tile: 108 (G0) / 216 (G1)
    int rowpass;               // 0 = products only
    int phase;                 // G1: 1 = the second workgroup of a CU runs half a tile out of phase
};

// ------------------------------------------------------------------------------------------------------------------------- G0
template <int V>
__global__ __launch_bounds__(256, 1) void g0_kernel(Args a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    constexpr int PIECE = 24576, NSLOT = 5;
    for (int i = t; i < NSLOT * PIECE / 4; i += 256) reinterpret_cast<unsigned*>(lds)[i] = 0x3f803f80u;
    __syncthreads();
    const unsigned base = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)lds;
    const unsigned lane_off = base + lane * 16;                       // fragment j of a piece: 1 KiB at j * 1024, lane-linear (conflict-free)
    const char* my_src = a.wsrc + lane * 16 + w * 6144;
    f16v acc[6];
#pragma unroll
    for (int j = 0; j < 6; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const u4 b0 = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    float fil[4] = {1.f, 1.f, 1.f, 1.f};
    unsigned long long ph[3] = {0, 0, 0}, tp = __builtin_amdgcn_s_memtime();
#define STAMP(i) { const unsigned long long tn = __builtin_amdgcn_s_memtime(); ph[i] += tn - tp; tp = tn; }
    int piece_no = 0;
    // prologue: 4 pieces in flight
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 6; ++i) glds16(my_src + ((piece_no + j) % 96) * PIECE + i * 1024, lds + ((piece_no + j) % NSLOT) * PIECE + w * 6144 + i * 1024);
    for (int tile = 0; tile < a.tiles_per_wg; ++tile) {
        for (int p = 0; p < a.pieces; ++p, ++piece_no) {
            asm volatile("s_waitcnt vmcnt(18)" ::: "memory");         // my share of this piece has landed, three younger pieces in flight
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            STAMP(1)
            const unsigned sb = lane_off + (unsigned)((piece_no % NSLOT) * PIECE);
            const char* nsrc = my_src + ((piece_no + 4) % 96) * PIECE;
            char* ndst = lds + ((piece_no + 4) % NSLOT) * PIECE + w * 6144;
            u4 fr[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fr[j]) : "v"(sb), "n"(j * 1024));
#pragma unroll
            for (int k = 0; k < 24; ++k) {
                asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(fr[k % 6]) : "n"(k < 19 ? 5 : 23 - k));
                acc[k % 6] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, fr[k % 6]), __builtin_bit_cast(bf8, b0), acc[k % 6], 0, 0, 0);
                if (k + 6 < 24) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fr[k % 6]) : "v"(sb), "n"((k + 6) * 1024));
                if (k % 4 == 1) glds16(nsrc + (k / 4) * 1024, ndst + (k / 4) * 1024);
                valu_fill<V>(fil, 1.0f);
            }
            STAMP(0)
        }
        if (a.rowpass) {
            // the wave's 32 rows: 48 KiB in (3 x 16 loads of 1 KiB), 72 KiB out
            const size_t wt = ((size_t)blockIdx.x * a.tiles_per_wg + tile) * 4 + w;
            const char* xi = a.xin + wt * 49152 + lane * 16;
            char* xo = a.xout + wt * 73728 + lane * 16;
            u4 sum = {0, 0, 0, 0};
            for (int c = 0; c < 6; ++c) {
                u4 v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const u4*>(xi + (c * 8 + i) * 1024);
#pragma unroll
                for (int i = 0; i < 8; ++i) sum += v[i];
#pragma unroll
                for (int i = 0; i < 12; ++i) *reinterpret_cast<u4*>(xo + (c * 12 + i) * 1024) = sum;
            }
            STAMP(2)
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0 && w == 0) { a.stamps[blockIdx.x * 4 + 0] = ph[0]; a.stamps[blockIdx.x * 4 + 1] = ph[1]; a.stamps[blockIdx.x * 4 + 2] = ph[2]; a.stamps[blockIdx.x * 4 + 3] = 0; }
    float s = fil[0] + fil[1] + fil[2] + fil[3];
#pragma unroll
    for (int j = 0; j < 6; ++j) s += acc[j][0];
    if (s == 12345.f) a.xout[0] = 1;
}

// ------------------------------------------------------------------------------------------------------------------------- G1
template <int V>
__global__ __launch_bounds__(256, 2) void g1_kernel(Args a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    constexpr int PIECE = 12288, NSLOT = 5;
    for (int i = t; i < NSLOT * PIECE / 4; i += 256) reinterpret_cast<unsigned*>(lds)[i] = 0x3f803f80u;
    // which of the CU's two workgroups am I?  (ticket per (XCC, SE/SH/CU) - read from the hardware id registers)
    __shared__ int second;
    if (t == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        const unsigned key = (xcc & 7u) * 256u + ((hw >> 8) & 255u);
        second = (int)(atomicAdd(a.tickets + key, 1u) & 1u);
    }
    __syncthreads();
    const int late = a.phase ? second : 0;
    const unsigned base = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)lds;
    const unsigned lane_off = base + lane * 16;
    const char* my_src = a.wsrc + lane * 16 + w * 3072;
    f4v acc[12];
#pragma unroll
    for (int j = 0; j < 12; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[j][r] = 0.f;
    const u4 b0 = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    float fil[4] = {1.f, 1.f, 1.f, 1.f};
    unsigned long long ph[3] = {0, 0, 0}, tp = __builtin_amdgcn_s_memtime();
    int piece_no = 0;
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 3; ++i) glds16(my_src + ((piece_no + j) % 192) * PIECE + i * 1024, lds + ((piece_no + j) % NSLOT) * PIECE + w * 3072 + i * 1024);
    // the late workgroup: half a tile, row pass, full tiles ..., half a tile (no row pass behind it: same products, same row passes)
    const int ntile = a.tiles_per_wg + late;
    for (int tile = 0; tile < ntile; ++tile) {
        const int np = late && (tile == 0 || tile == ntile - 1) ? a.pieces / 2 : a.pieces;
        for (int p = 0; p < np; ++p, ++piece_no) {
            asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            STAMP(1)
            const unsigned sb = lane_off + (unsigned)((piece_no % NSLOT) * PIECE);
            const char* nsrc = my_src + ((piece_no + 4) % 192) * PIECE;
            char* ndst = lds + ((piece_no + 4) % NSLOT) * PIECE + w * 3072;
            u4 fr[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fr[j]) : "v"(sb), "n"(j * 1024));
#pragma unroll
            for (int k = 0; k < 12; ++k) {
                asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(fr[k % 6]) : "n"(k < 7 ? 5 : 11 - k));
                acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, fr[k % 6]), __builtin_bit_cast(bf8, b0), acc[k], 0, 0, 0);
                if (k + 6 < 12) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fr[k % 6]) : "v"(sb), "n"((k + 6) * 1024));
                if (k % 4 == 1) glds16(nsrc + (k / 4) * 1024, ndst + (k / 4) * 1024);
                valu_fill<V>(fil, 1.0f);
            }
            STAMP(0)
        }
        if (a.rowpass && !(late && tile == ntile - 1)) {
            // the wave's 16 rows: 24 KiB in, 36 KiB out
            const size_t wt = ((size_t)blockIdx.x * a.tiles_per_wg + tile) * 4 + w;
            const char* xi = a.xin + wt * 24576 + lane * 16;
            char* xo = a.xout + wt * 36864 + lane * 16;
            u4 sum = {0, 0, 0, 0};
            for (int c = 0; c < 3; ++c) {
                u4 v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const u4*>(xi + (c * 8 + i) * 1024);
#pragma unroll
                for (int i = 0; i < 8; ++i) sum += v[i];
#pragma unroll
                for (int i = 0; i < 12; ++i) *reinterpret_cast<u4*>(xo + (c * 12 + i) * 1024) = sum;
            }
            STAMP(2)
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0 && w == 0) { a.stamps[blockIdx.x * 4 + 0] = ph[0]; a.stamps[blockIdx.x * 4 + 1] = ph[1]; a.stamps[blockIdx.x * 4 + 2] = ph[2]; a.stamps[blockIdx.x * 4 + 3] = (unsigned long long)second; }
    float s = fil[0] + fil[1] + fil[2] + fil[3];
#pragma unroll
    for (int j = 0; j < 12; ++j) s += acc[j][0];
    if (s == 12345.f) a.xout[0] = 1;
}

static Args g_args;
static unsigned long long* g_stamps_h;

template <typename K>
static void run(const char* name, K kernel, int grid, int smem, int pieces, int rowpass, int phase, int valu) {
    hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    Args a = g_args;
    a.pieces = pieces; a.rowpass = rowpass; a.phase = phase; a.tiles_per_wg = 4;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0, best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
        hipMemset(a.tickets, 0, 8 * 256 * 4);
        hipEventRecord(e0);
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), smem, 0, a);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 2 && ms < best) best = ms;
    }
    hipMemcpy(g_stamps_h, a.stamps, grid * 4 * 8, hipMemcpyDeviceToHost);
    // per-phase cycle totals: the mean over the early workgroups and over the late ones
    double s[2][3] = {{0, 0, 0}, {0, 0, 0}}; int n[2] = {0, 0};
    for (int b = 0; b < grid; ++b) {
        const int late = (int)g_stamps_h[b * 4 + 3] & 1;
        for (int i = 0; i < 3; ++i) s[late][i] += (double)g_stamps_h[b * 4 + i];
        ++n[late];
    }
    const double flops = 348.0e9;
    printf("{\"config\": \"%s\", \"valu_per_mfma\": %d, \"rowpass\": %d, \"phase\": %d, \"ms\": %.4f, \"tflops\": %.0f, \"frac_of_2.5PF\": %.3f, "
           "\"kcycles_wave0_early\": {\"n\": %d, \"products\": %.0f, \"ring_wait_barrier\": %.0f, \"rowpass\": %.0f}, "
           "\"kcycles_wave0_late\": {\"n\": %d, \"products\": %.0f, \"ring_wait_barrier\": %.0f, \"rowpass\": %.0f}}\n",
           name, valu, rowpass, phase, best, flops / best / 1e9, flops / best / 1e9 / 2500.0,
           n[0], n[0] ? s[0][0] / n[0] / 1e3 : 0, n[0] ? s[0][1] / n[0] / 1e3 : 0, n[0] ? s[0][2] / n[0] / 1e3 : 0,
           n[1], n[1] ? s[1][0] / n[1] / 1e3 : 0, n[1] ? s[1][1] / n[1] / 1e3 : 0, n[1] ? s[1][2] / n[1] / 1e3 : 0);
    fflush(stdout);
}

int main() {
    char *wsrc, *xin, *xout;
    const size_t nin = (size_t)131072 * 1536, nout = (size_t)131072 * 2304;
    hipMalloc(&wsrc, 192 * 12288 + 65536 + 98304); hipMemset(wsrc, 0, 192 * 12288 + 65536 + 98304);
    hipMalloc(&xin, nin + 65536); hipMemset(xin, 0, nin + 65536);
    hipMalloc(&xout, nout + 65536);
    hipMalloc(&g_args.stamps, 512 * 4 * 8);
    hipMalloc(&g_args.tickets, 8 * 256 * 4);
    g_stamps_h = (unsigned long long*)malloc(512 * 4 * 8);
    g_args.wsrc = wsrc; g_args.xin = xin; g_args.xout = xout;
    const int s0 = 5 * 24576, s1 = 5 * 12288;
    // warm the clocks
    for (int i = 0; i < 3; ++i) run("warm-up (ignore)", g0_kernel<6>, 256, s0, 108, 1, 0, 6);
    run("G0 one workgroup per CU, 32x32x16", g0_kernel<0>, 256, s0, 108, 0, 0, 0);
    run("G0 one workgroup per CU, 32x32x16", g0_kernel<6>, 256, s0, 108, 0, 0, 6);
    run("G0 one workgroup per CU, 32x32x16", g0_kernel<6>, 256, s0, 108, 1, 0, 6);
    run("G1 two workgroups per CU, 16x16x32", g1_kernel<0>, 512, s1, 216, 0, 0, 0);
    run("G1 two workgroups per CU, 16x16x32", g1_kernel<3>, 512, s1, 216, 0, 0, 3);
    run("G1 two workgroups per CU, 16x16x32", g1_kernel<3>, 512, s1, 216, 1, 0, 3);
    run("G1 two workgroups per CU, 16x16x32", g1_kernel<3>, 512, s1, 216, 1, 1, 3);
    run("G0 one workgroup per CU, 32x32x16", g0_kernel<6>, 256, s0, 108, 1, 0, 6);
    run("G1 two workgroups per CU, 16x16x32", g1_kernel<3>, 512, s1, 216, 1, 1, 3);
    return 0;
}
