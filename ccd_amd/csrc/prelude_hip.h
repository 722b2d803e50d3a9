// prelude_hip.h - device-side vocabulary of the ccd kernels on gfx950 (CDNA4, wave64).
// Kernel headers under kernels/ include nothing themselves; they are written against the few names
// defined here (ccd:: MFMA wrappers, wave shuffles, dynamic LDS accessor) plus plain HIP builtins.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

namespace ccd {
typedef __attribute__((ext_vector_type(8))) short bf16x8;    // 8 raw bf16 = one MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;   // 32x32 accumulator fragment
typedef __attribute__((ext_vector_type(4))) float f32x4;     // 16x16 accumulator fragment
typedef __attribute__((ext_vector_type(8))) __bf16 hw_bf16x8;

extern __shared__ __attribute__((aligned(16))) char dyn_smem_base[];
__device__ __forceinline__ char* dynamic_smem() { return dyn_smem_base; }
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

template <typename T>
__device__ __forceinline__ T shfl(T v, int src_lane) { return __shfl(v, src_lane, 64); }
template <typename T>
__device__ __forceinline__ T shfl_xor(T v, int mask) { return __shfl_xor(v, mask, 64); }
template <typename T>
__device__ __forceinline__ T shfl_down(T v, int d) { return __shfl_down(v, d, 64); }
__device__ __forceinline__ unsigned long long ballot(bool p) { return __ballot(p); }

// D = A*B + C on one wave64.  A[i=l&31][k=8*(l>>5)+e], B[k][j=l&31]; D col=l&31, row=(r&3)+8*(r>>2)+4*(l>>5)
__device__ __forceinline__ f32x16 mfma_32x32x16_bf16(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(hw_bf16x8, a), __builtin_bit_cast(hw_bf16x8, b),
                                                   c, 0, 0, 0);
}
// A[i=l&15][k=8*(l>>4)+e], B[k][j=l&15]; D col=l&15, row=4*(l>>4)+r
__device__ __forceinline__ f32x4 mfma_16x16x32_bf16(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(hw_bf16x8, a), __builtin_bit_cast(hw_bf16x8, b),
                                                   c, 0, 0, 0);
}
// Buffer addressing (guide T8/T20): a 128-bit resource descriptor built from wave-uniform values + a 32-bit per-lane
// byte offset.  A lane whose offset is >= the descriptor's byte count reads zeros WITHOUT touching memory - the GEMM
// loaders turn every row / tap / tail predicate into that (offset = BUF_OOB), so their loads need no exec-mask branches.
typedef __attribute__((ext_vector_type(4))) unsigned buf_u32x4;
typedef __amdgpu_buffer_rsrc_t buf_rsrc;
constexpr unsigned BUF_OOB = 0x80000000u;       // offsets of valid elements stay below 2 GiB (checked by the ABI)
__device__ __forceinline__ buf_rsrc make_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ buf_u32x4 buf_load16(buf_rsrc r, unsigned byte_offset) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_offset, 0, 0);
}
// the same with a wave-uniform (SGPR) part of the offset: address = base + lane_offset + uniform_offset, the bounds check
// (reads return zeros, writes are dropped) applies to the sum
__device__ __forceinline__ buf_u32x4 buf_load16(buf_rsrc r, unsigned lane_offset, unsigned uniform_offset) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, (int)lane_offset, (int)uniform_offset, 0);
}
// 8-byte form (four bf16 of a lane's row: the bf16 gradient stream of the LayerNorm-backward epilogues, round 6)
typedef __attribute__((ext_vector_type(2))) unsigned buf_u32x2;
__device__ __forceinline__ buf_u32x2 buf_load8(buf_rsrc r, unsigned lane_offset, unsigned uniform_offset) {
    return __builtin_amdgcn_raw_buffer_load_b64(r, (int)lane_offset, (int)uniform_offset, 0);
}
// STORE-DATA HAZARD (found in round 4 on rowgemm8.h, two waves per SIMD): a 16-byte buffer store reads its data registers a few
// cycles AFTER it issues.  LLVM's hazard recogniser inserts the wait state only when the store has no SGPR soffset (GCNHazardRecognizer:
// "this hazard only exists if the instruction is not using a register in the soffset field") - ours always has one, and on gfx950 a
// VALU write to the first data register right behind the store (`v_mov_b32 v18, 0`, an address computation) reached memory instead
// of the data in the last lanes of each half wave: run-to-run different values in rows 25 / 27 / 29 / 31 of a tile, lab_r04 notes in
// docs/LAB_NOTEBOOK.md section 4d.  The empty-bodied `s_nop` below takes the data as an INPUT: the registers stay allocated until two wait
// states have passed.
__device__ __forceinline__ void buf_store_data_hold(buf_u32x4 v) { asm volatile("s_nop 1" : : "v"(v)); }
__device__ __forceinline__ void buf_store16(buf_rsrc r, unsigned lane_offset, unsigned uniform_offset, buf_u32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)lane_offset, (int)uniform_offset, 0);
    buf_store_data_hold(v);
}
// Buffer loads whose ARRIVAL THE KERNEL TRACKS (rowgemm.h's activation blocks).  The compiler's own bookkeeping turns every
// wait for a builtin load inside a loop that also carries LDS-DMA requests and (from the previous tile's epilogue) stores
// into `s_waitcnt vmcnt(0)` - a full drain of the weight ring and of every prefetched block, twice per six windows in
// rowgemm.h's main loop (found in the ISA; it was 2/3 of that loop's time).  These are issued by hand instead: the compiler
// sees no VMEM operation (its own counted waits only get stricter), the destination is an ACCUMULATOR register quadruple (an
// MFMA B operand can come from there: no copy that could run ahead of the data), and the value may only be used behind
// vm_arrived<N>(dst), N = the number of loads / LDS-DMA requests issued after this one (loads return in issue order; stores
// are counted by vmcnt too but can only make the wait longer).  Raw descriptor words: what make_rsrc builds.
typedef buf_u32x4 buf_desc;
__device__ __forceinline__ buf_desc make_desc(const void* base, unsigned bytes) {
    const unsigned long long a = (unsigned long long)(uintptr_t)base;
    return buf_desc{(unsigned)a, (unsigned)(a >> 32) & 0xFFFFu, bytes, 0x00020000u};
}
__device__ __forceinline__ void buf_load16_late(buf_u32x4& dst, buf_desc r, unsigned lane_offset, unsigned uniform_offset) {
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=a"(dst) : "v"(lane_offset), "s"(r), "s"(uniform_offset) : "memory");
}
template <int N>
__device__ __forceinline__ void vm_arrived(buf_u32x4& a) { asm volatile("s_waitcnt vmcnt(%1)" : "+a"(a) : "n"(N)); }
// a 16-byte global load whose arrival the kernel tracks itself (attention_bwd1.h: the next block's v / o rows, requested under the
// current block's last products): the compiler sees no VMEM operation, the value may only be used behind a vmcnt wait of the
// kernel's own followed by vm_landed4 (no instruction: the dependency that keeps every use behind the wait)
template <int OFF = 0>
__device__ __forceinline__ void global_load16_late(buf_u32x4& dst, const void* p) {
    // ("+v": the load writes IN PLACE - with a plain output the allocator may give the result a fresh register and copy it into the
    // loop-carried one at the join, i.e. read it before the data has arrived)
    asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "+v"(dst) : "v"(p), "n"(OFF) : "memory");
}
__device__ __forceinline__ void vm_landed4(buf_u32x4 (&a)[4]) { asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3])); }
// "this value is needed HERE": the compiler places its wait for a load in front of the first instruction that reads the result.
// In an unrolled loop that consumes several prefetched rows and stores after each, that puts a wait for row i + 1 BEHIND the
// stores of row i - and with loads and stores pending on the one vmcnt the compiler makes it `s_waitcnt vmcnt(0)`: the store
// round trip is paid once per row.  Touching every prefetched value before the first store moves all those waits in front.
template <typename T>
__device__ __forceinline__ void needed_here(T& v) {
#ifndef CCD_LAB_NO_NEEDED_HERE      // (lab build: the A/B of this placement)
    asm volatile("" : "+v"(v));
#endif
}
// A wave-uniform fp32 through the SCALAR cache: retired by lgkmcnt, so its wait does not drain the vector memory queue (the weight
// ring, the previous tile's stores) the way a vector load's `vmcnt(0)` does.  The address must be wave-uniform; the data must
// not be written by this kernel (the scalar cache is invalidated at kernel boundaries only).
__device__ __forceinline__ float scalar_load_f32(const float* p) {
    float v;
    asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
    return v;
}
// a pointer the compiler must treat as wave-uniform (both halves through v_readfirstlane): scalar_load_f32's "s" operand
__device__ __forceinline__ const float* uniform_ptr(const float* p) {
    const unsigned long long a = (unsigned long long)(uintptr_t)p;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));
    return reinterpret_cast<const float*>((uintptr_t)(((unsigned long long)hi << 32) | lo));
}
// streaming variants: non-temporal cache policy (aux = 2, "nt"): data that is touched once must not push the L2-resident
// operands (weights) of the same kernel out of the XCD's 4-MiB L2
__device__ __forceinline__ buf_u32x4 buf_load16_nt(buf_rsrc r, unsigned lane_offset, unsigned uniform_offset) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, (int)lane_offset, (int)uniform_offset, 2);
}
// system-coherent read (aux bit 0, "sc0"/glc): past the CU's vector L1 - data this kernel itself wrote earlier
__device__ __forceinline__ buf_u32x4 buf_load16_coherent(buf_rsrc r, unsigned lane_offset, unsigned uniform_offset) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, (int)lane_offset, (int)uniform_offset, 1);
}
__device__ __forceinline__ void buf_store16_nt(buf_rsrc r, unsigned lane_offset, unsigned uniform_offset, buf_u32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)lane_offset, (int)uniform_offset, 2);
    buf_store_data_hold(v);
}
// One-touch streams of the large kernels: which of them carry the non-temporal policy is a compile-time table (bit B of CCD_NT;
// lab builds override it with -DCCD_NT=mask, tools/lab/nt_ab.sh), so that a kernel names its stream once: stream_load16<B> etc.
#ifndef CCD_NT
#define CCD_NT CCD_NT_DEFAULT
#endif
constexpr int NT_RG_A = 0, NT_RG_X = 1, NT_RG_G = 2, NT_MLP_U = 3, NT_MLP_Y = 4, NT_MLP_X = 5, NT_MLP_OUT = 6, NT_RP_A = 7, NT_RP_OUT = 8,
              NT_DGELU = 9, NT_TN = 10, NT_RG_GB = 11, NT_ATTB_OUT = 12;
constexpr unsigned CCD_NT_DEFAULT = 1u << NT_DGELU;
template <int B>
__device__ __forceinline__ buf_u32x4 stream_load16(buf_rsrc r, unsigned lane_offset, unsigned uniform_offset) {
    if constexpr ((CCD_NT >> B) & 1) return buf_load16_nt(r, lane_offset, uniform_offset);
    else return buf_load16(r, lane_offset, uniform_offset);
}
template <int B>
__device__ __forceinline__ void stream_store16(buf_rsrc r, unsigned lane_offset, unsigned uniform_offset, buf_u32x4 v) {
    if constexpr ((CCD_NT >> B) & 1) buf_store16_nt(r, lane_offset, uniform_offset, v);
    else buf_store16(r, lane_offset, uniform_offset, v);
}
__device__ __forceinline__ void wave_nap(int n) { for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(1); }        // ~n * 64 cycles
__device__ __forceinline__ void wave_sleep(int n) { for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(127); }   // ~n * 8 k cycles
// LDS-DMA: one wave instruction copies 64 x 16 B from per-lane global addresses straight into LDS at
// (wave-uniform lds_base) + lane * 16 - no VGPR round trip, counted on vmcnt like any VMEM load.
__device__ __forceinline__ void glds16(const void* gptr, char* lds_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gptr,
                                     (__attribute__((address_space(3))) void*)lds_base, 16, 0, 0);
}
// the same through a buffer descriptor (buffer_load_dwordx4 ... lds): address = base + lane_offset + uniform_offset, a lane whose
// offset is out of range moves ZEROS into its 16 bytes of LDS (checked on the GPU by the LayerNorm-backward tests: rows behind M)
__device__ __forceinline__ void bufdma16(buf_rsrc r, unsigned lane_offset, unsigned uniform_offset, char* lds_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_base, 16, (int)lane_offset, (int)uniform_offset, 0, 0);
}
template <int B>
__device__ __forceinline__ void stream_glds16(const void* gptr, char* lds_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gptr,
                                     (__attribute__((address_space(3))) void*)lds_base, 16, 0, ((CCD_NT >> B) & 1) ? 2 : 0);
}
template <int B>
__device__ __forceinline__ void stream_bufdma16(buf_rsrc r, unsigned lane_offset, unsigned uniform_offset, char* lds_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_base, 16, (int)lane_offset, (int)uniform_offset, 0,
                                             ((CCD_NT >> B) & 1) ? 2 : 0);
}
// 4-byte form: lane l's dword lands at lds_base + 4 l
__device__ __forceinline__ void glds4(const void* gptr, char* lds_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gptr,
                                     (__attribute__((address_space(3))) void*)lds_base, 4, 0, 0);
}
__device__ __forceinline__ void glds_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
template <int N>
__device__ __forceinline__ void glds_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }   // <= N VMEM ops outstanding
// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt while an LDS-DMA is in flight
// (the DMA is a pending LDS write on the VM counter); this one lets a DMA span the barrier.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// Orders LDS traffic between the LANES OF ONE WAVE (a wave that stages an image for itself needs no workgroup barrier:
// its LDS operations execute in program order; this only stops the compiler from moving the reads above the writes).
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// LDS fragment reads the compiler must not wait for with lgkmcnt(0): issued by hand, retired by a counted wait that names
// the fragment (guide 5.7 form (ii): the consumer cannot be scheduled above the wait).  LDS returns in order, so with D
// reads issued behind the one needed, lds_wait_frag<D> is exact; compiler-issued LDS operations in between only make it
// conservative.  lds_addr = byte address inside the workgroup's LDS.
__device__ __forceinline__ unsigned lds_addr_of(const void* p) {
    return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)p;
}
template <int OFF>
__device__ __forceinline__ void lds_read_frag(bf16x8& dst, unsigned lds_addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(lds_addr), "n"(OFF));
}
// 16-byte LDS write / read issued by hand on the same in-order queue (the caller's counted waits cover them: mlp_product's Extra)
__device__ __forceinline__ void lds_write16(unsigned lds_addr, const buf_u32x4& v) {
    asm volatile("ds_write_b128 %0, %1" : : "v"(lds_addr), "v"(v) : "memory");
}
__device__ __forceinline__ void lds_read16(buf_u32x4& dst, unsigned lds_addr) {
    asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(lds_addr));
}
template <int N>
__device__ __forceinline__ void lds_wait_frag(bf16x8& frag) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(frag) : "n"(N)); }
// a 4-byte LDS gather issued by hand (same in-order queue as the fragment reads), and the two ways its result becomes
// visible to the compiler: after a counted wait that some LATER hand-issued read has passed (lds_landed: no instruction,
// only a dependency), or after draining the queue (lds_drain)
__device__ __forceinline__ void lds_gather_f32(float& dst, unsigned lds_addr) {
    asm volatile("ds_read_b32 %0, %1" : "=v"(dst) : "v"(lds_addr));
}
__device__ __forceinline__ void lds_landed(float& a, float& b) { asm volatile("" : "+v"(a), "+v"(b)); }
typedef __attribute__((ext_vector_type(2))) float f32x2;
__device__ __forceinline__ void lds_gather_f32x2(f32x2& dst, unsigned lds_addr) {
    asm volatile("ds_read_b64 %0, %1" : "=v"(dst) : "v"(lds_addr));
}
__device__ __forceinline__ void lds_landed8(f32x2 (&f)[8]) {
    asm volatile("" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]));
}
__device__ __forceinline__ void lds_drain() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// four hand-issued fragment reads become visible to the compiler (no instruction: a counted or draining wait has passed)
__device__ __forceinline__ void lds_landed4(bf16x8 (&f)[4]) { asm volatile("" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3])); }
// the value lane (l ^ M) holds, M in {1, 2, 4, 8, 16}: DPP where gfx950 has a pattern for it (no LDS crossbar traffic), the
// swizzle unit otherwise
template <int M>
__device__ __forceinline__ float lane_xor(float v) {
    int x = __builtin_bit_cast(int, v);
    if constexpr (M == 1) x = __builtin_amdgcn_update_dpp(x, x, 0xB1, 0xF, 0xF, false);          // quad_perm:[1,0,3,2]
    else if constexpr (M == 2) x = __builtin_amdgcn_update_dpp(x, x, 0x4E, 0xF, 0xF, false);     // quad_perm:[2,3,0,1]
    else if constexpr (M == 8) x = __builtin_amdgcn_update_dpp(x, x, 0x128, 0xF, 0xF, false);    // row_ror:8
    else x = __builtin_amdgcn_ds_swizzle(x, (M << 10) | 0x1F);                                    // swizzle(SWAP, M)
    return __builtin_bit_cast(float, x);
}
// static wave priority (guide T5): 0..3, arbitration between the waves sharing a SIMD
template <int P>
__device__ __forceinline__ void wave_prio() { __builtin_amdgcn_s_setprio(P); }
// value the optimiser must treat as unknown (keeps `base ^ constant` address arithmetic inside a loop instead of
// hoisting one register per combination)
__device__ __forceinline__ unsigned opaque_u32(unsigned x) { asm volatile("" : "+s"(x)); return x; }
// the same for a per-lane value, and NOT hoistable out of a loop: address arithmetic derived from it is redone where it is
// used instead of living in a register (or, worse, in scratch) across a register-starved main loop
__device__ __forceinline__ int opaque_vgpr(int x) { asm volatile("" : "+v"(x)); return x; }
// A VMEM load that moves nothing (empty descriptor: every offset is out of range, no memory access) but occupies a slot
// of the wave's vmcnt queue - keeps "VMEM instructions per window" a compile-time constant for counted waits.  The zero it
// returns lands in `sink` whenever the load retires: `sink` must stay a live, otherwise unused register.
__device__ __forceinline__ void vmem_pad_load(unsigned& sink) {
    const buf_u32x4 none = {0u, 0u, 0u, 0x00020000u};
    asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "+v"(sink) : "v"(BUF_OOB), "s"(none));
}
// ds_read_b64_tr_b16 (gfx950): within each group of 16 lanes, lane j addresses 4 contiguous bf16 D[j][0..3]; lane i receives
// {D[i / 4][i % 4], D[4 + i / 4][i % 4], D[8 + i / 4][i % 4], D[12 + i / 4][i % 4]} - with lane j pointing at (row j / 4, columns
// 4 (j % 4) ..) of a row-major 4 x 16 block that is column i of the four rows (tools/probe/tr_probe.hip).  The compiler
// counts it in lgkmcnt like any LDS read.
typedef __attribute__((ext_vector_type(4))) short tr_v4s;
typedef __attribute__((ext_vector_type(2))) unsigned tr_u32x2;
__device__ __forceinline__ tr_u32x2 lds_read_tr16(const char* p) {
    const tr_v4s v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tr_v4s*)p);
    return __builtin_bit_cast(tr_u32x2, v);
}
// hand-issued form (same in-order queue and counted waits as lds_read_frag): the builtin's LDS memory operand makes the compiler
// put `s_waitcnt vmcnt(0)` in front of it whenever an LDS-DMA is in flight (it cannot tell the DMA's buffer from the one being read)
template <int OFF>
__device__ __forceinline__ void lds_read_tr(tr_u32x2& dst, unsigned lds_addr) {
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(lds_addr), "n"(OFF));
}
__device__ __forceinline__ bf16x8 frag_from_tr(tr_u32x2 lo, tr_u32x2 hi) {
    const buf_u32x4 v = {lo.x, lo.y, hi.x, hi.y};
    return __builtin_bit_cast(bf16x8, v);
}
// v_permlane32_swap (gfx950): lanes l < 32 keep a and receive lane l + 32's a in b; lanes l >= 32 keep b and receive lane
// l - 32's b in a - two half-wave exchanges in one instruction
__device__ __forceinline__ void lane32_swap(unsigned& a, unsigned& b) {
    const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    a = r[0];
    b = r[1];
}
// v_perm_b32: result byte i = byte sel.byte[i] of the 8-byte value {hi : lo} (0..3 = lo's bytes, 4..7 = hi's bytes)
__device__ __forceinline__ unsigned perm_b32(unsigned hi, unsigned lo, unsigned sel) { return __builtin_amdgcn_perm(hi, lo, sel); }
__device__ __forceinline__ float opaque_f32(float x) { asm volatile("" : "+v"(x)); return x; }
// hardware float -> bf16 (RNE): clang lowers the __bf16 casts to v_cvt_pk_bf16_f32 on gfx950
typedef __attribute__((ext_vector_type(2))) __bf16 hw_bf16x2;
__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
    hw_bf16x2 v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ unsigned short cvt_bf16(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }
__device__ __forceinline__ float fast_exp(float x) { return __expf(x); }
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }   // bare v_exp_f32
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }   // bare v_rcp_f32 (1 ulp); __frcp_rn expands to an 11-instruction IEEE division
// scheduling hints (guide T19): pin the issue order of an unrolled block as groups of instruction kinds
#define CCD_SGB_MFMA(n) __builtin_amdgcn_sched_group_barrier(0x8, (n), 0)
#define CCD_SGB_DS_READ(n) __builtin_amdgcn_sched_group_barrier(0x100, (n), 0)
#define CCD_SGB_VALU(n) __builtin_amdgcn_sched_group_barrier(0x2, (n), 0)
#define CCD_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
__device__ __forceinline__ int uniform_i32(int x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ float fast_rsqrt(float x) { return rsqrtf(x); }
}  // namespace ccd

template <typename F>
static inline const void* ccd_fn_ptr(F* f) { return reinterpret_cast<const void*>(f); }
// dynamic LDS above 64 KiB has to be opted into once per kernel and device (gfx950 allows up to 160 KiB per workgroup)
#define CCD_LAUNCH(kernel, grid, block, smem, stream, ...)                                                      \
    do {                                                                                                        \
        if ((smem) > 65536) {                                                                                   \
            /* the attribute is per function AND per device; a call site whose smem varies between launches (rowproj +N*4,   \
               gemm256 with column sums, augment_spatial by image size) must raise it again: keep the largest set so far */   \
            static int ccd_max_[64] = {0};                                                                      \
            int ccd_dev_ = 0;                                                                                   \
            (void)hipGetDevice(&ccd_dev_);                                                                      \
            if ((int)(smem) > ccd_max_[ccd_dev_ & 63]) {                                                        \
                (void)hipFuncSetAttribute(ccd_fn_ptr(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,       \
                                          (int)(smem));                                                         \
                ccd_max_[ccd_dev_ & 63] = (int)(smem);                                                          \
            }                                                                                                   \
        }                                                                                                       \
        hipLaunchKernelGGL(kernel, (grid), (block), (smem), (hipStream_t)(stream), __VA_ARGS__);                \
    } while (0)
static inline int ccd_rt_memset_async(void* p, int v, size_t n, void* stream) {
    return (int)hipMemsetAsync(p, v, n, (hipStream_t)stream);
}
static inline int ccd_rt_last_error() { return (int)hipGetLastError(); }
static inline int ccd_rt_num_cus() {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    return cus;
}
