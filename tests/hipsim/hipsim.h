// hipsim.h - TEST INFRASTRUCTURE ONLY.
//
// A small host-side SIMT executor so that the *same* kernel sources and C-ABI launchers that ship in
// ccd_amd/csrc can be executed on a CPU at tiny problem sizes (this build container has no GPU and GPU
// time is scarce).  It provides the handful of names ccd_amd/csrc/prelude_hip.h provides on the device:
// __global__/__shared__/threadIdx/.../__syncthreads, wave64 shuffles, atomics and an emulation of the
// MFMA instructions with the documented gfx950 operand/accumulator layouts (MI355X guide section 3).
// Every lane is a fiber (a hand-rolled x86-64 context switch: no signal-mask system calls, unlike swapcontext); a
// workgroup is run by one OS thread, workgroups run in parallel.
// It is NOT a compatibility layer of the product: ccd_amd never loads anything built from this file.
#pragma once

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static thread_local
#define __launch_bounds__(...)

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
typedef void* hipStream_t;

namespace sim {
constexpr int WAVE = 64;
struct Block;
struct Ctx { void* rsp = nullptr; };                 // callee-saved registers live on the fiber's own stack
extern "C" void sim_switch(Ctx* from, Ctx* to);
struct Lane {
    dim3 tid;
    int linear = 0;
    int wave = 0;
    int lane = 0;
    Ctx ctx;
    bool done = false;
    int wait_kind = 0;       // 0 runnable, 1 waiting wave barrier, 2 waiting block barrier
    unsigned wait_gen = 0;
    Block* blk = nullptr;
    void* stack = nullptr;
    // LDS-DMA requests of this lane that have not been retired by a counted wait yet (late mode, see glds16 below)
    struct PendingDma { char* dst; const void* src; int bytes; };
    std::vector<PendingDma> dma;
};
struct WaveState {
    int alive = 0, arrived = 0;
    unsigned gen = 0;
    alignas(16) unsigned char xchg[WAVE][64];   // per-lane exchange slot (up to 64 bytes)
};
struct Block {
    dim3 bid, bdim, gdim;
    std::vector<Lane> lanes;
    std::vector<WaveState> waves;
    int alive = 0, arrived = 0;
    unsigned gen = 0;
    Ctx sched;
    char* dyn_smem = nullptr;
    const std::function<void()>* body = nullptr;
};
extern thread_local Lane* cur;
extern thread_local Block* curblk;

inline void yield_to_scheduler() { sim_switch(&cur->ctx, &cur->blk->sched); }

inline void wave_sync() {
    Lane* l = cur;
    WaveState& w = l->blk->waves[l->wave];
    if (++w.arrived >= w.alive) {
        w.arrived = 0;
        ++w.gen;
        return;
    }
    l->wait_kind = 1;
    l->wait_gen = w.gen;
    yield_to_scheduler();
}
inline void block_sync() {
    Lane* l = cur;
    Block* b = l->blk;
    if (++b->arrived >= b->alive) {
        b->arrived = 0;
        ++b->gen;
        return;
    }
    l->wait_kind = 2;
    l->wait_gen = b->gen;
    yield_to_scheduler();
}
void launch(dim3 grid, dim3 block, size_t dyn_smem, const std::function<void()>& body);
}  // namespace sim

#define threadIdx (sim::cur->tid)
#define blockIdx (sim::curblk->bid)
#define blockDim (sim::curblk->bdim)
#define gridDim (sim::curblk->gdim)
inline bool sim_dma_late() {
    static const bool late = [] { const char* v = getenv("CCD_SIM_DMA"); return v && v[0] == 'l'; }();
    return late;
}
inline void sim_dma_retire(size_t leave) {
    auto& q = sim::cur->dma;
    const size_t n = q.size() > leave ? q.size() - leave : 0;
    for (size_t i = 0; i < n; ++i)
        if (q[i].bytes) {
            if (q[i].src) std::memcpy(q[i].dst, q[i].src, (size_t)q[i].bytes);
            else std::memset(q[i].dst, 0, (size_t)q[i].bytes);      // a late register load of an out-of-range address
        }
    q.erase(q.begin(), q.begin() + (long)n);
}
inline void __syncthreads() { sim_dma_retire(0); sim::block_sync(); }      // (the compiler drains vmcnt in front of s_barrier)

// ------------------------------------------------------------------------------------------- atomics
template <typename T>
inline T sim_atomic_rmw(T* p, T v, T (*op)(T, T)) {
    static_assert(sizeof(T) == 4 || sizeof(T) == 8, "");
    using U = typename std::conditional<sizeof(T) == 4, uint32_t, uint64_t>::type;
    U* up = reinterpret_cast<U*>(p);
    U old = __atomic_load_n(up, __ATOMIC_RELAXED);
    for (;;) {
        T o;
        std::memcpy(&o, &old, sizeof(T));
        T n = op(o, v);
        U nu;
        std::memcpy(&nu, &n, sizeof(T));
        if (__atomic_compare_exchange_n(up, &old, nu, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) return o;
    }
}
inline float atomicAdd(float* p, float v) { return sim_atomic_rmw<float>(p, v, [](float a, float b) { return a + b; }); }
inline int atomicAdd(int* p, int v) { return sim_atomic_rmw<int>(p, v, [](int a, int b) { return a + b; }); }
inline unsigned atomicAdd(unsigned* p, unsigned v) { return sim_atomic_rmw<unsigned>(p, v, [](unsigned a, unsigned b) { return a + b; }); }
inline int atomicMin(int* p, int v) { return sim_atomic_rmw<int>(p, v, [](int a, int b) { return a < b ? a : b; }); }
inline int atomicMax(int* p, int v) { return sim_atomic_rmw<int>(p, v, [](int a, int b) { return a > b ? a : b; }); }
inline unsigned atomicMin(unsigned* p, unsigned v) { return sim_atomic_rmw<unsigned>(p, v, [](unsigned a, unsigned b) { return a < b ? a : b; }); }

// --------------------------------------------------------------------------- the ccd:: device prelude
namespace ccd {
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

inline char* dynamic_smem() { return sim::curblk->dyn_smem; }
// LDS-DMA (prelude_hip.h): the executor copies synchronously, lane by lane
// LDS-DMA (prelude_hip.h).  Two models of WHEN the bytes land, selected by CCD_SIM_DMA:
//   early (default): at issue, lane by lane - the earliest the hardware could deliver them: a DMA into a buffer that other waves
//                    still read shows up as wrong results;
//   late:            only when a counted wait of the issuing lane retires the request (`glds_wait<N>` leaves the N youngest in
//                    flight, in issue order; every lane runs the same instruction stream, so the per-lane queue is the wave's
//                    vmcnt) - the latest the hardware may deliver them: a fragment read in front of a sufficient wait sees stale
//                    LDS.  Loads that return to registers (vmem_pad_load) are NOT counted, i.e. they are modelled as returning at
//                    once: a wait window that relies on such loads to keep its count is too lax here, as it was on the GPU.
inline void glds16(const void* gptr, char* lds_base) {
    char* dst = lds_base + 16 * sim::cur->lane;
    if (::sim_dma_late()) sim::cur->dma.push_back({dst, gptr, 16});
    else std::memcpy(dst, gptr, 16);
}
inline void glds4(const void* gptr, char* lds_base) {
    char* dst = lds_base + 4 * sim::cur->lane;
    if (::sim_dma_late()) sim::cur->dma.push_back({dst, gptr, 4});
    else std::memcpy(dst, gptr, 4);
}
inline void glds_wait_all() { ::sim_dma_retire(0); }
template <int N>
inline void glds_wait() { ::sim_dma_retire((size_t)N); }
inline unsigned opaque_u32(unsigned x) { return x; }
inline int opaque_vgpr(int x) { return x; }
inline float opaque_f32(float x) { return x; }
inline void vmem_pad_load(unsigned& sink) { (void)sink; }
// ds_read_b64_tr_b16 (prelude_hip.h): every lane deposits the 4 bf16 it addresses, lane i of a 16-lane group collects element
// i % 4 of lanes i / 4, 4 + i / 4, 8 + i / 4, 12 + i / 4
typedef __attribute__((ext_vector_type(2))) unsigned tr_u32x2;
inline tr_u32x2 lds_read_tr16(const char* p) {
    sim::WaveState& w = sim::curblk->waves[sim::cur->wave];
    const int lane = sim::cur->lane, grp = lane & ~15, i = lane & 15;
    std::memcpy(w.xchg[lane], p, 8);
    sim::wave_sync();
    unsigned short e[4];
    for (int r = 0; r < 4; ++r) std::memcpy(&e[r], w.xchg[grp + 4 * r + (i >> 2)] + 2 * (i & 3), 2);
    sim::wave_sync();
    tr_u32x2 out;
    out.x = (unsigned)e[0] | ((unsigned)e[1] << 16);
    out.y = (unsigned)e[2] | ((unsigned)e[3] << 16);
    return out;
}
inline unsigned perm_b32(unsigned hi, unsigned lo, unsigned sel) {
    const unsigned long long v = ((unsigned long long)hi << 32) | lo;
    unsigned r = 0;
    for (int i = 0; i < 4; ++i) r |= (unsigned)((v >> (8 * ((sel >> (8 * i)) & 7))) & 0xffu) << (8 * i);
    return r;
}
// hand-issued LDS fragment reads (prelude_hip.h): synchronous here; LDS "addresses" are offsets from the block's dynamic LDS
inline unsigned lds_addr_of(const void* p) { return (unsigned)(reinterpret_cast<const char*>(p) - sim::curblk->dyn_smem); }
template <int OFF>
inline void lds_read_frag(bf16x8& dst, unsigned lds_addr) { std::memcpy(&dst, sim::curblk->dyn_smem + lds_addr + OFF, 16); }
template <int N>
inline void lds_wait_frag(bf16x8&) {}
inline void lds_landed4(bf16x8 (&)[4]) {}
inline void lds_gather_f32(float& dst, unsigned lds_addr) { std::memcpy(&dst, sim::curblk->dyn_smem + lds_addr, 4); }
inline void lds_landed(float&, float&) {}
typedef float f32x2 __attribute__((ext_vector_type(2)));
inline void lds_gather_f32x2(f32x2& dst, unsigned lds_addr) { std::memcpy(&dst, sim::curblk->dyn_smem + lds_addr, 8); }
inline void lds_landed8(f32x2 (&)[8]) {}
inline void lds_drain() {}
template <int P>
inline void wave_prio() {}
inline void lds_barrier() { sim::block_sync(); }                           // LDS-only barrier: an LDS-DMA in flight spans it
// buffer addressing: out-of-range lanes read zeros (prelude_hip.h)
typedef __attribute__((ext_vector_type(4))) unsigned buf_u32x4;
template <int OFF>
inline void lds_read_tr(tr_u32x2& dst, unsigned lds_addr) { dst = lds_read_tr16(sim::curblk->dyn_smem + lds_addr + OFF); }
inline bf16x8 frag_from_tr(tr_u32x2 lo, tr_u32x2 hi) {
    const buf_u32x4 v = {lo.x, lo.y, hi.x, hi.y};
    bf16x8 r;
    std::memcpy(&r, &v, 16);
    return r;
}
struct buf_rsrc { const char* base; unsigned bytes; };
constexpr unsigned BUF_OOB = 0x80000000u;
inline buf_rsrc make_rsrc(const void* base, unsigned bytes) { return buf_rsrc{reinterpret_cast<const char*>(base), bytes}; }
// (late-DMA mode: a buffer load of real data keeps its place in the lane's in-order queue - its data is here at once, but a
// counted wait still has to get past it - while vmem_pad_load, out of range by construction, holds no place: see glds16)
inline void sim_vmem_slot() { if (::sim_dma_late()) sim::cur->dma.push_back({nullptr, nullptr, 0}); }
inline buf_u32x4 buf_load16(buf_rsrc r, unsigned byte_offset) {
    sim_vmem_slot();
    buf_u32x4 v = {0u, 0u, 0u, 0u};
    if ((unsigned long long)byte_offset + 16ull <= (unsigned long long)r.bytes) std::memcpy(&v, r.base + byte_offset, 16);
    return v;
}
inline buf_u32x4 buf_load16(buf_rsrc r, unsigned lane_offset, unsigned uniform_offset) {
    sim_vmem_slot();
    const unsigned long long o = (unsigned long long)lane_offset + uniform_offset;
    buf_u32x4 v = {0u, 0u, 0u, 0u};
    if (o + 16ull <= (unsigned long long)r.bytes) std::memcpy(&v, r.base + o, 16);
    return v;
}
typedef __attribute__((ext_vector_type(2))) unsigned buf_u32x2;
inline buf_u32x2 buf_load8(buf_rsrc r, unsigned lane_offset, unsigned uniform_offset) {
    sim_vmem_slot();
    const unsigned long long o = (unsigned long long)lane_offset + uniform_offset;
    buf_u32x2 v = {0u, 0u};
    if (o + 8ull <= (unsigned long long)r.bytes) std::memcpy(&v, r.base + o, 8);
    return v;
}
inline void bufdma16(buf_rsrc r, unsigned lane_offset, unsigned uniform_offset, char* lds_base) {
    const unsigned long long o = (unsigned long long)lane_offset + uniform_offset;
    const bool in_range = o + 16ull <= (unsigned long long)r.bytes;
    char* dst = lds_base + 16 * sim::cur->lane;
    if (::sim_dma_late()) sim::cur->dma.push_back({dst, in_range ? r.base + o : nullptr, 16});
    else if (in_range) std::memcpy(dst, r.base + o, 16);
    else std::memset(dst, 0, 16);
}
inline void buf_store16(buf_rsrc r, unsigned lane_offset, unsigned uniform_offset, buf_u32x4 v) {
    const unsigned long long o = (unsigned long long)lane_offset + uniform_offset;
    if (o + 16ull <= (unsigned long long)r.bytes) std::memcpy(const_cast<char*>(r.base) + o, &v, 16);
}
inline buf_u32x4 buf_load16_nt(buf_rsrc r, unsigned a, unsigned b) { return buf_load16(r, a, b); }
inline buf_u32x4 buf_load16_coherent(buf_rsrc r, unsigned a, unsigned b) { return buf_load16(r, a, b); }
constexpr int NT_RG_A = 0, NT_RG_X = 1, NT_RG_G = 2, NT_MLP_U = 3, NT_MLP_Y = 4, NT_MLP_X = 5, NT_MLP_OUT = 6, NT_RP_A = 7, NT_RP_OUT = 8,
              NT_DGELU = 9, NT_TN = 10, NT_RG_GB = 11, NT_ATTB_OUT = 12;
template <int B> inline buf_u32x4 stream_load16(buf_rsrc r, unsigned a, unsigned b) { return buf_load16(r, a, b); }
template <typename T>
inline void needed_here(T&) {}
inline float scalar_load_f32(const float* p) { return *p; }
inline const float* uniform_ptr(const float* p) { return p; }
// hand-tracked buffer loads (prelude_hip.h: buf_load16_late / vm_arrived).  Late mode: the destination is poisoned (bf16 NaNs)
// at issue and receives its data only when a counted wait of this lane retires the request - the latest the hardware may
// deliver it, in the same in-order queue as the LDS-DMA requests: a use in front of a sufficient wait sees NaNs.
typedef buf_rsrc buf_desc;
inline buf_desc make_desc(const void* base, unsigned bytes) { return make_rsrc(base, bytes); }
inline void buf_load16_late(buf_u32x4& dst, buf_desc r, unsigned lane_offset, unsigned uniform_offset) {
    const unsigned long long o = (unsigned long long)lane_offset + uniform_offset;
    const bool in_range = o + 16ull <= (unsigned long long)r.bytes;
    if (::sim_dma_late()) {
        dst = buf_u32x4{0x7FC07FC0u, 0x7FC07FC0u, 0x7FC07FC0u, 0x7FC07FC0u};
        sim::cur->dma.push_back({reinterpret_cast<char*>(&dst), in_range ? r.base + o : nullptr, 16});
    } else {
        dst = buf_u32x4{0u, 0u, 0u, 0u};
        if (in_range) std::memcpy(&dst, r.base + o, 16);
    }
}
template <int N>
inline void vm_arrived(buf_u32x4&) { ::sim_dma_retire((size_t)N); }
template <int OFF = 0>
inline void global_load16_late(buf_u32x4& dst, const void* p0) {
    const void* p = reinterpret_cast<const char*>(p0) + OFF;
    if (::sim_dma_late()) {
        dst = buf_u32x4{0x7FC07FC0u, 0x7FC07FC0u, 0x7FC07FC0u, 0x7FC07FC0u};
        sim::cur->dma.push_back({reinterpret_cast<char*>(&dst), p, 16});
    } else {
        std::memcpy(&dst, p, 16);
    }
}
inline void vm_landed4(buf_u32x4 (&)[4]) {}
inline void buf_store16_nt(buf_rsrc r, unsigned a, unsigned b, buf_u32x4 v) { buf_store16(r, a, b, v); }
inline void lds_write16(unsigned lds_addr, const buf_u32x4& v) { std::memcpy(sim::curblk->dyn_smem + lds_addr, &v, 16); }
inline void lds_read16(buf_u32x4& dst, unsigned lds_addr) { std::memcpy(&dst, sim::curblk->dyn_smem + lds_addr, 16); }
template <int B> inline void stream_store16(buf_rsrc r, unsigned a, unsigned b, buf_u32x4 v) { buf_store16(r, a, b, v); }
template <int B> inline void stream_glds16(const void* gptr, char* lds_base) { glds16(gptr, lds_base); }
template <int B> inline void stream_bufdma16(buf_rsrc r, unsigned a, unsigned b, char* lds_base) { bufdma16(r, a, b, lds_base); }
inline void wave_sleep(int) {}
inline void wave_nap(int) {}
inline int lane_id() { return sim::cur->lane; }
inline int wave_id() { return sim::cur->wave; }

template <typename T>
inline T shfl(T v, int src_lane) {
    static_assert(sizeof(T) <= 64, "");
    sim::WaveState& w = sim::curblk->waves[sim::cur->wave];
    std::memcpy(w.xchg[sim::cur->lane], &v, sizeof(T));
    sim::wave_sync();
    T r;
    std::memcpy(&r, w.xchg[src_lane & 63], sizeof(T));
    sim::wave_sync();
    return r;
}
inline void lane32_swap(unsigned& a, unsigned& b) {           // prelude_hip.h: v_permlane32_swap
    const int lane = sim::cur->lane;
    const unsigned pa = shfl(a, lane ^ 32), pb = shfl(b, lane ^ 32);
    if (lane < 32) b = pa; else a = pb;
}
template <typename T>
inline T shfl_xor(T v, int mask) { return shfl(v, sim::cur->lane ^ mask); }
template <int M>
inline float lane_xor(float v) { return shfl(v, sim::cur->lane ^ M); }
template <typename T>
inline T shfl_down(T v, int d) { int s = sim::cur->lane + d; return shfl(v, s > 63 ? sim::cur->lane : s); }
inline unsigned long long ballot(bool p) {
    unsigned long long bit = p ? 1ull : 0ull;
    unsigned long long all = 0;
    // gather through shuffles (slow but simple)
    sim::WaveState& w = sim::curblk->waves[sim::cur->wave];
    std::memcpy(w.xchg[sim::cur->lane], &bit, 8);
    sim::wave_sync();
    for (int i = 0; i < 64; ++i) {
        unsigned long long b;
        std::memcpy(&b, w.xchg[i], 8);
        // lanes that already exited keep their last slot content; treat missing lanes as 0
        all |= (b & 1ull) << i;
    }
    sim::wave_sync();
    return all;
}
// lanes of a wave are separate fibers here: a wave-level LDS fence is a wave rendezvous
inline void wave_lds_fence() { (void)ballot(true); }
inline float bf16_bits_to_f32(unsigned short h) { uint32_t u = (uint32_t)h << 16; float f; std::memcpy(&f, &u, 4); return f; }

// v_mfma_f32_32x32x16_bf16: A[i=l&31][k=8*(l>>5)+e], B[k=8*(l>>5)+e][j=l&31],
// D: col = l&31, row = (r&3) + 8*(r>>2) + 4*(l>>5)
inline f32x16 mfma_32x32x16_bf16(bf16x8 a, bf16x8 b, f32x16 c) {
    struct AB { short a[8]; short b[8]; };
    sim::WaveState& w = sim::curblk->waves[sim::cur->wave];
    AB me;
    for (int e = 0; e < 8; ++e) { me.a[e] = a[e]; me.b[e] = b[e]; }
    std::memcpy(w.xchg[sim::cur->lane], &me, sizeof(me));
    sim::wave_sync();
    int l = sim::cur->lane;
    int col = l & 31;
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        float acc = c[r];
        for (int k = 0; k < 16; ++k) {
            AB la, lb;
            std::memcpy(&la, w.xchg[row + 32 * (k >> 3)], sizeof(AB));
            std::memcpy(&lb, w.xchg[col + 32 * (k >> 3)], sizeof(AB));
            acc += bf16_bits_to_f32((unsigned short)la.a[k & 7]) * bf16_bits_to_f32((unsigned short)lb.b[k & 7]);
        }
        c[r] = acc;
    }
    sim::wave_sync();
    return c;
}
// v_mfma_f32_16x16x32_bf16: A[i=l&15][k=8*(l>>4)+e], B[k=8*(l>>4)+e][j=l&15], D: col=l&15, row=4*(l>>4)+r
inline f32x4 mfma_16x16x32_bf16(bf16x8 a, bf16x8 b, f32x4 c) {
    struct AB { short a[8]; short b[8]; };
    sim::WaveState& w = sim::curblk->waves[sim::cur->wave];
    AB me;
    for (int e = 0; e < 8; ++e) { me.a[e] = a[e]; me.b[e] = b[e]; }
    std::memcpy(w.xchg[sim::cur->lane], &me, sizeof(me));
    sim::wave_sync();
    int l = sim::cur->lane;
    int col = l & 15;
    for (int r = 0; r < 4; ++r) {
        int row = 4 * (l >> 4) + r;
        float acc = c[r];
        for (int k = 0; k < 32; ++k) {
            AB la, lb;
            std::memcpy(&la, w.xchg[row + 16 * (k >> 3)], sizeof(AB));
            std::memcpy(&lb, w.xchg[col + 16 * (k >> 3)], sizeof(AB));
            acc += bf16_bits_to_f32((unsigned short)la.a[k & 7]) * bf16_bits_to_f32((unsigned short)lb.b[k & 7]);
        }
        c[r] = acc;
    }
    sim::wave_sync();
    return c;
}
inline unsigned short cvt_bf16(float f) {          // round-to-nearest-even, NaN preserved
    uint32_t u; std::memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40u);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
inline unsigned cvt_pk_bf16(float lo, float hi) { return (unsigned)cvt_bf16(lo) | ((unsigned)cvt_bf16(hi) << 16); }
inline float fast_exp(float x) { return std::exp(x); }
inline float fast_exp2(float x) { return std::exp2(x); }
inline float fast_rcp(float x) { return 1.0f / x; }
#define CCD_SGB_MFMA(n)
#define CCD_SGB_DS_READ(n)
#define CCD_SGB_VALU(n)
#define CCD_SCHED_FENCE()
inline int uniform_i32(int x) { return x; }
inline float fast_rsqrt(float x) { return 1.0f / std::sqrt(x); }
}  // namespace ccd

// <cmath> already declares ::erff, ::expf, ::fabsf, ::fmaxf, ::fminf, ::logf, ::sqrtf in the global namespace
inline float rsqrtf(float x) { return 1.0f / std::sqrt(x); }
inline float __fmul_rn(float a, float b) { volatile float r = a * b; return r; }
inline float __fadd_rn(float a, float b) { volatile float r = a + b; return r; }
inline float __fsub_rn(float a, float b) { volatile float r = a - b; return r; }
inline float __fdiv_rn(float a, float b) { volatile float r = a / b; return r; }

// launch + runtime shims used by abi_impl.h
#define CCD_LAUNCH(kernel, grid, block, smem, stream, ...) \
    sim::launch((grid), (block), (smem), [=]() { kernel(__VA_ARGS__); })
inline int ccd_rt_memset_async(void* p, int v, size_t n, hipStream_t) { std::memset(p, v, n); return 0; }
inline int ccd_rt_last_error() { return 0; }
// size of the emulated chip: 4 "CUs" by default (persistent kernels then run 4-8 workgroups on as many host
// threads); tests that want every workgroup to walk several work items set CCD_SIM_CUS=1
inline int ccd_rt_num_cus() {
    const char* v = getenv("CCD_SIM_CUS");
    const int n = v ? atoi(v) : 4;
    return n > 0 ? n : 1;
}
