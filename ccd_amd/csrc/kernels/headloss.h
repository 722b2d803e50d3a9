// headloss.h - the DINO head's last layer and the distillation loss WITHOUT the logits round trip
// (Dino/modules/vision_transformer.py:324-328: x = normalize(x); x = last_layer(x)  +  Dino/loss/Dino_loss.py:81-105):
//     s[i, k] = zs[i, :] . Ws[k, :]      t[j, k] = zt[j, :] . Wt[k, :]        (K = 65 536 columns, D = 256)
//     loss = mean_i ( lse(s[i] / Ts) - sum_k softmax((t[partner(i)] - c) / Tt)[k] * s[i, k] / Ts )
// The unfused path writes both [2M, K] fp32 logit matrices (2 x 865 MB at 256 images per GPU), reads them back in dino_loss_fwd
// and again in dino_loss_bwd.  Here a logit lives in an accumulator register for the time it takes to fold it into the row's
// running (max, sum, cross term); the backward pass recomputes the same products (bit-identical: same kernel body, same order)
// and writes only the bf16 logit gradient that the head's weight / data gradient products read.
//
// Structure: rowproj.h's ("row owners", E = D = 256).  A wave keeps 32 student rows AND their 32 partner teacher rows in registers
// as MFMA B operands (2 x 64 VGPRs); only the weights move - pieces of 64 output columns x one K half (16 KiB) of Ws and Wt in
// turn through the 5-slot LDS ring by LDS-DMA (counted vmcnt, one LDS-only barrier per piece).  The products are computed
// transposed (S^T[64 columns][32 rows]) so that a lane owns a row: the online-softmax state of a row is five registers of the two
// lanes that hold its column halves, merged once per work item.
// Work item = (128-row tile, column split): the K columns are cut into CS splits (a multiple of 8) and XCD x works on the splits
// congruent to x modulo 8, row tiles innermost - the workgroups of an XCD stream the same ~1 MB of weights through its L2 at the
// same time.  A split leaves one partial (m_s, l_s, m_t, l_t, dot) per row; head_loss_finish_kernel merges the CS partials of a row,
// adds the row's loss and keeps (m_s, 1 / l_s, m_t, 1 / l_t) for the backward pass.  All exponentials are base 2 (v_exp_f32): the
// temperatures are folded into ks = log2(e) / Ts, kt = log2(e) / Tt and the loss is scaled back by ln 2.
#pragma once

namespace ccd {

struct HeadLossParams {
    const bf16_t* zs;       // [rows, 256] bf16: L2-normalised bottleneck rows of the student
    long ld_zs;
    const bf16_t* zt;       // the teacher's
    long ld_zt;
    const bf16_t* ws;       // [K, 256] bf16: weight-normed last layer of the student
    long ld_ws;
    const bf16_t* wt;
    long ld_wt;
    const float* center;    // [K]
    const int* d_m;         // M (rows per view) in device memory; rows = 2 M
    int max_rows;           // capacity of zs / zt / d_logits in rows
    int K, CS, chunks;      // columns, column splits (CS % 8 == 0), 64-column chunks per split: K == CS * chunks * 64
    float ks, kt;           // log2(e) / student_temp, log2(e) / teacher_temp
    float* part;            // forward: [CS][max_rows][8] partials
    const float* stats;     // backward: [max_rows][4] = m_s, 1 / l_s, m_t, 1 / l_t (base-2 domain)
    float grad_scale;
    const float* d_grad_scale;
    bf16_t* d_logits;       // backward: [rows, K] bf16
    long ld_d;
};

constexpr int HL_THREADS = 256, HL_D = 256, HL_BM = 128, HL_SCRATCH = 4096, HL_NSLOT = 3, HL_PIECE = 32 * HL_D * 2;
__host__ __device__ inline int hl_smem_bytes(int chunks) { return HL_NSLOT * HL_PIECE + 4 * HL_SCRATCH + 2 * chunks * 64 * 4; }

struct HlState { float ms, ls, mt, lt, dot; };
__device__ __forceinline__ void hl_merge(HlState& a, float oms, float ols, float omt, float olt, float odot) {
    const float nms = fmaxf(a.ms, oms);
    a.ls = a.ls * fast_exp2(a.ms - nms) + ols * fast_exp2(oms - nms);
    a.ms = nms;
    const float nmt = fmaxf(a.mt, omt);
    const float f0 = fast_exp2(a.mt - nmt), f1 = fast_exp2(omt - nmt);
    a.lt = a.lt * f0 + olt * f1;
    a.dot = a.dot * f0 + odot * f1;
    a.mt = nmt;
}

template <bool BWD>
__global__ __launch_bounds__(HL_THREADS, 2) void head_loss_kernel(HeadLossParams p) {
    constexpr int E = HL_D, KT = E / 64, KJ = E / 16, PIECE = HL_PIECE, NSLOT = HL_NSLOT, AHEAD = NSLOT - 1, DEPTH = 6;
    static_assert(KJ == 16 && KT == 4, "a piece = 64 weight rows x one K half of 128 = 16 MFMA steps");
    const int M = uniform_i32(p.d_m[0]);
    const int rows = 2 * M < p.max_rows ? 2 * M : p.max_rows;
    const int RT = (rows + HL_BM - 1) / HL_BM;
    const int G = gridDim.x, Gx = G >> 3, xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int items = (p.CS >> 3) * RT;                  // of this XCD: (split j, row tile), row tiles innermost
    if (q >= items) return;
    char* smem = dynamic_smem();
    const int t = threadIdx.x, lane = t & 63, hf = lane >> 5, lq = lane & 31;
    const int w = uniform_i32(t >> 6);
    char* scratch = smem + NSLOT * PIECE + w * HL_SCRATCH;
    float* ctab = reinterpret_cast<float*>(smem + NSLOT * PIECE + 4 * HL_SCRATCH);      // [2][chunks * 64]: kt * centre of the item's columns
    const int ccols = p.chunks * 64;
    const int NP = 4 * p.chunks;                         // pieces per item: chunk c = Ws(c, 0) Ws(c, 1) Wt(c, 0) Wt(c, 1)

    // ---- weight ring (rowproj.h): piece = [2 k-tiles of one K half][64 weight rows][128 B]; wave w moves instruction i = rows
    // 32 (i & 1) + 8 w .. + 7 of k-tile i >> 1
    const int dr = lane >> 3, dp = lane & 7, drow = 8 * w + dr;
    const unsigned swz16 = (unsigned)((dp ^ mlp_swz(drow)) * 16);
    const unsigned req_lane_s = (unsigned)(2 * drow) * (unsigned)p.ld_ws + swz16, req_lane_t = (unsigned)(2 * drow) * (unsigned)p.ld_wt + swz16;
    int slot_i = 0, slot_c = 0, pos_i = 0, n_i = q;
    const char* req_base = nullptr;
    char* req_lds = nullptr;
    long step_a = 0;
    unsigned req_lane = 0;
    // the issue stream's item: (split j_i, row tile rt_i) of item n_i, kept incrementally (no division per piece), and where the
    // item's columns start in either weight matrix
    int j_i = q / RT, rt_i = q - j_i * RT;
    const unsigned cstride_s = (unsigned)(64 * p.ld_ws * 2), cstride_t = (unsigned)(64 * p.ld_wt * 2);     // bytes per 64-column chunk
    const char* item_s = reinterpret_cast<const char*>(p.ws) + (unsigned long long)((xcd + 8 * j_i) * p.chunks) * cstride_s;
    const char* item_t = reinterpret_cast<const char*>(p.wt) + (unsigned long long)((xcd + 8 * j_i) * p.chunks) * cstride_t;
    auto issue_prepare = [&]() __attribute__((always_inline)) {
        const int chunk = pos_i >> 2, net = (pos_i >> 1) & 1, half = pos_i & 1;
        req_base = (net ? item_t : item_s) + (unsigned)chunk * (net ? cstride_t : cstride_s) + half * E;      // (E / 2 elements = E bytes)
        step_a = 64 * (net ? p.ld_wt : p.ld_ws);
        req_lane = net ? req_lane_t : req_lane_s;
        req_lds = smem + slot_i * PIECE + w * 1024;
        slot_i = slot_i + 1 == NSLOT ? 0 : slot_i + 1;
        pos_i = pos_i + 1;
        if (pos_i == NP) {
            pos_i = 0;
            if (n_i + Gx < items) {        // behind the last item: the same pieces again (drained at the end)
                n_i += Gx;
                rt_i += Gx;
                while (rt_i >= RT) { rt_i -= RT; ++j_i; }
                item_s = reinterpret_cast<const char*>(p.ws) + (unsigned long long)((xcd + 8 * j_i) * p.chunks) * cstride_s;
                item_t = reinterpret_cast<const char*>(p.wt) + (unsigned long long)((xcd + 8 * j_i) * p.chunks) * cstride_t;
            }
        }
    };
    auto issue_one = [&](int i) __attribute__((always_inline)) {
        glds16(req_base + ((i & 1) * step_a + (i >> 1) * 128) + req_lane, req_lds + 4096 * i);
    };
    const unsigned smem_addr = lds_addr_of(smem);
    auto acquire = [&]() __attribute__((always_inline)) -> unsigned {
        glds_wait<(AHEAD - 1) * KT>();
        lds_barrier();
        issue_prepare();
        const unsigned sb = smem_addr + (unsigned)(slot_c * PIECE);
        slot_c = slot_c + 1 == NSLOT ? 0 : slot_c + 1;
        return sb;
    };
#pragma unroll
    for (int j = 0; j < AHEAD; ++j) {
        issue_prepare();
#pragma unroll
        for (int i = 0; i < KT; ++i) issue_one(i);
    }
    // weight row lq of a 32-row tile is fed with bits 2 and 3 of the row index swapped: accumulator registers 8 s + (0 .. 7) of
    // tile tt hold columns 32 tt + 16 s + 8 hf + (0 .. 7) of the chunk (rowproj.h)
    const int prow = (lq & 19) | ((lq & 4) << 1) | ((lq & 8) >> 1);
    unsigned off1[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) off1[kk] = (unsigned)(prow * 128 + (((2 * kk + hf) ^ mlp_swz(prow)) * 16));

    const float gs = BWD ? p.grad_scale * (p.d_grad_scale ? p.d_grad_scale[0] : 1.0f) * (p.ks * 0.6931471805599453f) / (float)(2 * M) : 0.f;
    int it = 0;
    for (int n = q; n < items; n += Gx, ++it) {
        const int cs = xcd + 8 * (n / RT), rt = n % RT;
        const int r0 = rt * HL_BM + 32 * w, row = r0 + lq;
        const bool live = row < rows;
        const int prt = row < M ? row + M : row - M;     // the partner row of the other view (Dino_loss.py:88-102: the two cross terms)
        // ---- this item's centre columns, times kt, into the table of this parity (the other parity may still be read by a wave
        // that has not passed the item's first barrier)
        float* ct = ctab + (it & 1) * ccols;
        {
            const float* cg = p.center + (long)cs * ccols;
            for (int i = 4 * t; i < ccols; i += 4 * HL_THREADS) {
                const f32x4v c = *reinterpret_cast<const f32x4v*>(cg + i);
                *reinterpret_cast<f32x4v*>(ct + i) = f32x4v{c.x * p.kt, c.y * p.kt, c.z * p.kt, c.w * p.kt};
            }
        }
        // ---- the wave's rows: row lq of the student tile, its partner of the teacher; k = 16 j + 8 hf .. + 7
        bf16x8 as_[KJ], at_[KJ];
        {
            const buf_rsrc rs_s = make_rsrc(p.zs, (unsigned)(((long)rows - 1) * p.ld_zs + E) * 2u);
            const buf_rsrc rs_t = make_rsrc(p.zt, (unsigned)(((long)rows - 1) * p.ld_zt + E) * 2u);
            const unsigned lo_s = live ? (unsigned)(((long)row * p.ld_zs + 8 * hf) * 2) : BUF_OOB;
            const unsigned lo_t = live ? (unsigned)(((long)prt * p.ld_zt + 8 * hf) * 2) : BUF_OOB;
#pragma unroll
            for (int j = 0; j < KJ; ++j) {
                as_[j] = __builtin_bit_cast(bf16x8, buf_load16(rs_s, lo_s, 32 * j));
                at_[j] = __builtin_bit_cast(bf16x8, buf_load16(rs_t, lo_t, 32 * j));
            }
        }
        HlState st;
        st.ms = -3.0e38f; st.ls = 0.f; st.mt = -3.0e38f; st.lt = 0.f; st.dot = 0.f;
        float b_ms = 0.f, b_ils = 0.f, b_mt = 0.f, b_ilt = 0.f;
        if constexpr (BWD) {
            const f32x4v sv = *reinterpret_cast<const f32x4v*>(p.stats + 4L * (live ? row : 0));
            b_ms = sv.x; b_ils = sv.y * gs; b_mt = sv.z; b_ilt = sv.w * gs;
        }
        // the wave's 32 rows of d_logits through a descriptor of its own (offsets stay small whatever rows x K is)
        const int rows_left = rows - r0 > 32 ? 32 : (rows - r0 > 0 ? rows - r0 : 0);
        const buf_rsrc rs_d = make_rsrc(BWD ? p.d_logits + (long)r0 * p.ld_d : nullptr,
                                        BWD && rows_left > 0 ? (unsigned)((((long)rows_left - 1) * p.ld_d + p.K) * 2) : 0u);
        // one chunk = four pieces: Ws(c, 0), Ws(c, 1) -> hs;  Wt(c, 0), Wt(c, 1) -> ht;  filler(piece index, MFMA step) rides between the MFMAs
        auto chunk_products = [&](f32x16 (&hs)[2], f32x16 (&ht)[2], auto&& filler) __attribute__((always_inline)) {
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int r = 0; r < 16; ++r) { hs[tt][r] = 0.f; ht[tt][r] = 0.f; }
            auto piece = [&](auto NET, auto KH) __attribute__((always_inline)) {
                constexpr int net = decltype(NET)::value, kh = decltype(KH)::value;
                const unsigned sb = acquire();
                const unsigned areg[4] = {sb + off1[0], sb + off1[1], sb + off1[2], sb + off1[3]};
                mlp_product<KJ, DEPTH, MlpMapP1<KT / 2>, MlpNoExtra>(
                    areg,
                    [&](auto Kk, const bf16x8& a) {
                        constexpr int k = decltype(Kk)::value;
                        if constexpr (net == 0) hs[k & 1] = mfma_32x32x16_bf16(a, as_[(KJ / 2) * kh + (k >> 1)], hs[k & 1]);
                        else ht[k & 1] = mfma_32x32x16_bf16(a, at_[(KJ / 2) * kh + (k >> 1)], ht[k & 1]);
                    },
                    [&](auto Kk) {
                        constexpr int k = decltype(Kk)::value, stride = KJ / KT;
                        if constexpr (k % stride == 1 && k / stride < KT) issue_one(k / stride);
                        filler(std::integral_constant<int, 2 * net + kh>{}, Kk);
                    });
            };
            piece(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
            piece(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
            piece(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
            piece(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
        };
        // register r of tile tt = column 64 c + 32 tt + 16 (r >> 3) + 8 hf + (r & 7) of the split, this lane's row
        if constexpr (!BWD) {
            // (Round 6 also built the software-pipelined form - the online-softmax update of chunk c - 1 riding between the MFMAs of chunk
            // c, one logit pair per MFMA step - and measured it SLOWER: 396 against 338 us per launch.  With one wave per SIMD a VALU
            // instruction costs its issue slot wherever it stands (tools/probe/two_wg_probe.hip: 6 fillers per MFMA take the product loop from
            // 33.5 to 54 cycles per MFMA), and between the MFMAs it also delays the next fragment wait.  The update runs behind its chunk.)
            f32x16 hA[2], tA[2];
            const float* cvp = ct;
            float cms = -3.0e38f, cmt = -3.0e38f, ls0 = 0.f, ls1 = 0.f, lt0 = 0.f, lt1 = 0.f, d0 = 0.f, d1 = 0.f;
            auto load_cv = [&](int c_of) __attribute__((always_inline)) {           // where the centre columns of chunk c_of start for this half-wave
                cvp = ct + 64 * c_of + 8 * hf;
                cms = -3.0e38f; cmt = -3.0e38f;
            };
            auto stage_a = [&](auto EI, f32x16 (&hP)[2], f32x16 (&tP)[2]) __attribute__((always_inline)) {
                constexpr int e = decltype(EI)::value, tt = e >> 4, r = e & 15;
                const float xt = fmaf(tP[tt][r], p.kt, -cvp[32 * tt + 16 * (r >> 3) + (r & 7)]);
                tP[tt][r] = xt;
                cmt = fmaxf(cmt, xt);
                cms = fmaxf(cms, hP[tt][r]);
            };
            auto rescale = [&]() __attribute__((always_inline)) {
                const float nms = fmaxf(st.ms, cms * p.ks), nmt = fmaxf(st.mt, cmt);
                const float fs = fast_exp2(st.ms - nms), ft = fast_exp2(st.mt - nmt);
                st.ls *= fs; st.lt *= ft; st.dot *= ft;
                st.ms = nms; st.mt = nmt;
                ls0 = ls1 = lt0 = lt1 = d0 = d1 = 0.f;
            };
            auto stage_b = [&](auto EI, f32x16 (&hP)[2], f32x16 (&tP)[2]) __attribute__((always_inline)) {
                constexpr int e = decltype(EI)::value, tt = e >> 4, r = e & 15;
                const float es = fast_exp2(fmaf(hP[tt][r], p.ks, -st.ms)), pt = fast_exp2(tP[tt][r] - st.mt);
                if constexpr (e & 1) { ls1 += es; lt1 += pt; d1 = fmaf(pt, hP[tt][r], d1); }       // (dot carries sum p * s: times ks when it is published)
                else { ls0 += es; lt0 += pt; d0 = fmaf(pt, hP[tt][r], d0); }
            };
            auto finish_b = [&]() __attribute__((always_inline)) { st.ls += ls0 + ls1; st.lt += lt0 + lt1; st.dot += d0 + d1; };
#pragma unroll 1
            for (int c = 0; c < p.chunks; ++c) {
                chunk_products(hA, tA, [](auto, auto) {});
                load_cv(c);
                mlp_static_for<0, 32>([&](auto EI) { stage_a(EI, hA, tA); });
                rescale();
                mlp_static_for<0, 32>([&](auto EI) { stage_b(EI, hA, tA); });
                finish_b();
            }
        } else {
#pragma unroll 1
            for (int c = 0; c < p.chunks; ++c) {
                f32x16 hs[2], ht[2];
                chunk_products(hs, ht, [](auto, auto) {});
                const float* cc = ct + 64 * c + 8 * hf;
                // d s[i, k] = gs * (softmax_s - softmax_t): packed to bf16, out through the wave's scratch image as 128-byte row segments
                const int ln = opaque_vgpr(t) & 63, lhf = ln >> 5, llq = ln & 31, ldr_ = ln >> 3, ldp = ln & 7;
#pragma unroll
                for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        const f32x4v ca = *reinterpret_cast<const f32x4v*>(cc + 32 * tt + 16 * s), cb = *reinterpret_cast<const f32x4v*>(cc + 32 * tt + 16 * s + 4);
                        const float c8[8] = {ca.x, ca.y, ca.z, ca.w, cb.x, cb.y, cb.z, cb.w};
                        float d[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const int r = 8 * s + e;
                            const float ps = fast_exp2(fmaf(hs[tt][r], p.ks, -b_ms)) * b_ils;
                            const float pt = fast_exp2(fmaf(ht[tt][r], p.kt, -c8[e]) - b_mt) * b_ilt;
                            d[e] = ps - pt;
                        }
                        u32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = pack_bf2(d[2 * e], d[2 * e + 1]);
                        const int slot16 = 4 * tt + 2 * s + lhf;             // columns 8 * slot16 .. + 7
                        *reinterpret_cast<u32x4*>(scratch + llq * 128 + ((slot16 ^ (llq & 7)) * 16)) = v;
                    }
                wave_lds_fence();
                const unsigned lo_o = (unsigned)(ldr_ * p.ld_d * 2 + ldp * 16);
                const unsigned col_b = (unsigned)((((long)cs * p.chunks + c) * 64) * 2);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const u32x4 v = *reinterpret_cast<const u32x4*>(scratch + (ldr_ + 8 * i) * 128 + ((ldp ^ ldr_) * 16));
                    buf_store16(rs_d, lo_o, (unsigned)(8 * i) * (unsigned)(p.ld_d * 2) + col_b, v);
                }
                wave_lds_fence();
            }
        }
        if constexpr (!BWD) {
            // the row's two column halves meet; the lower half-wave publishes the split's partial
            st.dot *= p.ks;
            hl_merge(st, shfl_xor(st.ms, 32), shfl_xor(st.ls, 32), shfl_xor(st.mt, 32), shfl_xor(st.lt, 32), shfl_xor(st.dot, 32));
            if (live && hf == 0) {
                float* o = p.part + ((long)cs * p.max_rows + row) * 8;
                *reinterpret_cast<f32x4v*>(o) = f32x4v{st.ms, st.ls, st.mt, st.lt};
                o[4] = st.dot;
            }
        }
    }
    glds_wait_all();                       // requested pieces that no item consumed must not outlive the workgroup's LDS
}

// merges the CS partials of a row: stats[row] = (m_s, 1 / l_s, m_t, 1 / l_t) in the base-2 domain, loss += row loss / (2 M)
__global__ __launch_bounds__(256) void head_loss_finish_kernel(const float* __restrict__ part, const int* __restrict__ d_m, int max_rows, int CS,
                                                               float* __restrict__ stats, float* __restrict__ loss_out) {
    __shared__ float red[4];
    const int M = d_m[0];
    const int rows = 2 * M < max_rows ? 2 * M : max_rows;
    const int row = blockIdx.x * 256 + threadIdx.x;
    float row_loss = 0.f;
    if (row < rows) {
        HlState st;
        const float* o = part + (long)row * 8;
        const f32x4v v0 = *reinterpret_cast<const f32x4v*>(o);
        st.ms = v0.x; st.ls = v0.y; st.mt = v0.z; st.lt = v0.w; st.dot = o[4];
        int cs = 1;
        for (; cs + 7 < CS; cs += 8) {         // (eight splits' partials in flight: one per trip was a chain of CS L2 round trips, 28 us)
            f32x4v v[8];
            float d[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float* oc = part + ((long)(cs + u) * max_rows + row) * 8;
                v[u] = *reinterpret_cast<const f32x4v*>(oc);
                d[u] = oc[4];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) hl_merge(st, v[u].x, v[u].y, v[u].z, v[u].w, d[u]);
        }
        for (; cs < CS; ++cs) {
            const float* oc = part + ((long)cs * max_rows + row) * 8;
            const f32x4v v = *reinterpret_cast<const f32x4v*>(oc);
            hl_merge(st, v.x, v.y, v.z, v.w, oc[4]);
        }
        *reinterpret_cast<f32x4v*>(stats + 4L * row) = f32x4v{st.ms, 1.0f / st.ls, st.mt, 1.0f / st.lt};
        // -sum_k p_t log_softmax(s):  ln 2 * ((m_s + log2 l_s) - dot / l_t)
        row_loss = 0.6931471805599453f * ((st.ms + log2f(st.ls)) - st.dot / st.lt) / (float)(2 * M);
    }
    row_loss = wave_sum(row_loss);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = row_loss;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float s = red[0] + red[1] + red[2] + red[3];
        if (s != 0.f) atomicAdd(loss_out, s);
    }
}

}  // namespace ccd
