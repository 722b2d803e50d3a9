// Probe of ds_read_b64_tr_b16 (gfx950): which LDS elements does each lane receive for a given per-lane address pattern?
// LDS holds u16 value = element index.  Build: hipcc --offload-arch=gfx950 -O2 tr_probe.hip -o tr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(const unsigned* addr, unsigned short* out) {
    __shared__ unsigned short lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned short*)lds;
    const unsigned a = base + addr[threadIdx.x];
    typedef __attribute__((ext_vector_type(2))) unsigned u2;
    u2 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(a));
    out[threadIdx.x * 4 + 0] = r.x & 0xffff; out[threadIdx.x * 4 + 1] = r.x >> 16;
    out[threadIdx.x * 4 + 2] = r.y & 0xffff; out[threadIdx.x * 4 + 3] = r.y >> 16;
}
int main() {
    unsigned* d_addr; unsigned short* d_out;
    hipMalloc(&d_addr, 64 * 4); hipMalloc(&d_out, 64 * 4 * 2);
    // pattern A: lane l -> byte l*8 (lane-linear 4 elements each)
    // pattern B: lanes of a 16-group address a [4 rows][16 cols] block of a row-major matrix with 64 columns:
    //            lane j -> row j/4, col 4*(j%4); group gidx -> cols +16*gidx
    // pattern C: same with rows = j%4, cols 4*(j/4)
    for (int pat = 0; pat < 3; ++pat) {
        std::vector<unsigned> a(64);
        for (int l = 0; l < 64; ++l) {
            const int j = l & 15, grp = l >> 4;
            if (pat == 0) a[l] = l * 8;
            else if (pat == 1) a[l] = ((j / 4) * 64 + 4 * (j % 4) + 16 * grp) * 2;
            else a[l] = ((j % 4) * 64 + 4 * (j / 4) + 16 * grp) * 2;
        }
        hipMemcpy(d_addr, a.data(), 256, hipMemcpyHostToDevice);
        probe<<<1, 64>>>(d_addr, d_out);
        std::vector<unsigned short> o(256);
        hipMemcpy(o.data(), d_out, 512, hipMemcpyDeviceToHost);
        printf("pattern %d\n", pat);
        for (int l = 0; l < 64; ++l) printf("lane %2d addr_elem %4u -> %4u %4u %4u %4u\n", l, a[l] / 2, o[4 * l], o[4 * l + 1], o[4 * l + 2], o[4 * l + 3]);
    }
    return 0;
}
