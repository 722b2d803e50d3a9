// gemm_row384.h - NT GEMM for the N <= 384 products of the transformer (proj / fc2 with the fp32 residual epilogue,
// the data gradients of qkv / fc1 / proj): one workgroup owns FULL output rows - a 128 x 384 tile, 8 waves (2 x 4,
// 64 x 96 per wave), BK = 64, one workgroup per CU.
//
// Why: with 128-wide column tiles (gemm.h) the A row-panel of such a product is streamed three times, by three
// workgroups that drift apart in the persistent schedule, so the re-reads miss the XCD's L2 (PMC: 859 MB moved per
// launch for 655 MB of algorithmic traffic on the residual GEMMs, the step's dominant and HBM-bound kernel kind).
// Here A is read exactly once, there is no column-partial tile, and an epilogue that owns whole rows can later take
// the following LayerNorm with it.
//
// Structure = gemm.h's: operands by buffer loads (branch-free predicates, zero VALU in the loop), two register sets
// hold k-tiles t+1 / t+2 (2 + 6 sixteen-byte pieces per thread and set) while tile t is multiplied out of one of two
// 64-KiB LDS stages, persistent work list with the next tile's first k-tiles in flight under the epilogue.  The
// products are accumulated transposed (lane = one row, 4 consecutive columns), staged through a swizzled LDS image
// (bf16 64 rows / fp32 32 rows per pass) and written out by the shared row epilogue.
#pragma once

namespace ccd {

constexpr int GR_BM = 128, GR_BN = 384, GR_BK = 64, GR_THREADS = 512;
constexpr int GR_A_BYTES = GR_BM * GR_BK * 2, GR_B_BYTES = GR_BN * GR_BK * 2;   // 16 KiB + 48 KiB per stage
constexpr int GR_STAGE_BYTES = GR_A_BYTES + GR_B_BYTES;                          // 64 KiB
constexpr int GR_VEC_BYTES = 3 * GR_BN * 4;                                      // bias, LayerNorm gamma / beta (fp32)
constexpr int GR_SMEM_BYTES = 2 * GR_STAGE_BYTES + 2 * GR_VEC_BYTES;             // 128 KiB + vectors + EPI_LNBWD column sums

template <int EPI>
__global__ __launch_bounds__(GR_THREADS, 1) void gemm_row384_kernel(GemmParams p) {
    const int m_static = p.M;
    if (p.d_rows) {
        const int dyn = p.d_rows[0] * p.rows_mul;
        p.M = dyn < p.M ? dyn : p.M;
    }
    char* smem = dynamic_smem();
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, hf = lane >> 5, lq = lane & 31;
    const int wm = w & 1, wn = w >> 1;                       // wm fastest: any column range covers all four SIMDs
    const unsigned total = (unsigned)((p.M + GR_BM - 1) / GR_BM), G = gridDim.x;       // live row tiles only
    const unsigned ng = G < 8u ? G : 8u;
    const unsigned xcd = blockIdx.x % ng, slot = blockIdx.x / ng;
    const unsigned nx = G / ng + (xcd < G % ng ? 1u : 0u);
    const unsigned q8 = total / ng, r8 = total % ng;
    const unsigned base_x = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const unsigned cnt_x = q8 + (xcd < r8 ? 1u : 0u);
    if (slot >= cnt_x) return;
    const int nk_full = p.K / GR_BK;
    // the per-column vectors of the epilogue live in LDS for the whole (persistent) kernel: fetched from global memory
    // inside the row sweep, every one of them was a serialised L2 round trip behind `s_waitcnt vmcnt(0)`
    float* vbias = reinterpret_cast<float*>(smem + 2 * GR_STAGE_BYTES);
    float* vgamma = vbias + GR_BN;
    float* vbeta = vgamma + GR_BN;
    float* lds_colsum = vbeta + GR_BN;                    // EPI_LNBWD: [3][GR_BN] column sums of the whole workgroup
    for (int i = t; i < 3 * GR_BN; i += GR_THREADS) lds_colsum[i] = 0.f;
    for (int i = t; i < GR_BN; i += GR_THREADS) {
        vbias[i] = (p.bias && i < p.N) ? p.bias[i] : 0.f;
        vgamma[i] = ((EPI == EPI_RESID_LN || EPI == EPI_LNBWD) && i < p.N) ? p.ln_gamma[i] : 0.f;
        vbeta[i] = (EPI == EPI_RESID_LN && i < p.N) ? p.ln_beta[i] : 0.f;
    }
    __syncthreads();

    // staging pieces of this thread: A rows (t >> 3) + 64 i, B rows (t >> 3) + 64 i, 16-byte slot t & 7
    unsigned offa[2], offb[6];
    const int srow = t >> 3, sslot = t & 7;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int row = srow + 64 * i;
        offb[i] = row < p.N ? (unsigned)(((long)row * p.ldb + sslot * 8) * 2) : BUF_OOB;
    }
    int m0, nk;
    auto setup = [&](unsigned item) {
        m0 = (int)(base_x + item) * GR_BM;
        nk = m0 < p.M ? nk_full : 0;                        // tiles past a device-side row count do nothing
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = m0 + srow + 64 * i;
            offa[i] = row < p.M ? (unsigned)(((long)row * p.lda + sslot * 8) * 2) : BUF_OOB;
        }
    };
    u32x4 ra0[2], rb0[6], ra1[2], rb1[6];
    auto load = [&](int kt, u32x4 (&ra)[2], u32x4 (&rb)[6]) {
        const unsigned whole = kt < nk ? BUF_OOB : 0u;
        const buf_rsrc rsa = make_rsrc(p.A + kt * GR_BK, whole), rsb = make_rsrc(p.B + kt * GR_BK, whole);
#pragma unroll
        for (int i = 0; i < 2; ++i) ra[i] = buf_load16(rsa, offa[i]);
#pragma unroll
        for (int i = 0; i < 6; ++i) rb[i] = buf_load16(rsb, offb[i]);
    };
    auto store = [&](int stage, const u32x4 (&ra)[2], const u32x4 (&rb)[6]) {
        char* as = smem + stage * GR_STAGE_BYTES;
        char* bs = as + GR_A_BYTES;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = srow + 64 * i;
            *reinterpret_cast<u32x4*>(as + row * 128 + gemm_swz(row, sslot) * 16) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int row = srow + 64 * i;
            *reinterpret_cast<u32x4*>(bs + row * 128 + gemm_swz(row, sslot) * 16) = rb[i];
        }
    };
    // fragment addresses: offset(kk) = base ^ (kk << 5) (see gemm256.h), stage bit 1 << 16 folded into the XOR
    unsigned base_a[2], base_b[3];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = 64 * wm + 32 * i + lq, f = ((row >> 1) ^ (row >> 4)) & 7;
        base_a[i] = (unsigned)(row * 128 + ((f >> 1) << 5) + ((hf ^ (f & 1)) << 4));
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int row = 96 * wn + 32 * j + lq, f = ((row >> 1) ^ (row >> 4)) & 7;
        base_b[j] = (unsigned)(GR_A_BYTES + row * 128 + ((f >> 1) << 5) + ((hf ^ (f & 1)) << 4));
    }
    f32x16 acc[2][3];
    auto compute = [&](unsigned stagebit) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const unsigned x = stagebit | (unsigned)(kk << 5);
            bf16x8 a[2], b[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) b[j] = *reinterpret_cast<const bf16x8*>(smem + (base_b[j] ^ x));
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const bf16x8*>(smem + (base_a[i] ^ x));
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) acc[i][j] = mfma_32x32x16_bf16(b[j], a[i], acc[i][j]);   // D^T[n][m]
        }
    };

    unsigned item = slot;
    setup(item);
    load(0, ra0, rb0);
    load(1, ra1, rb1);
    while (true) {
        store(0, ra0, rb0);
        load(2, ra0, rb0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        __syncthreads();
        for (int kt = 0; kt < nk; kt += 2) {                // branch-free steady state, as gemm.h
            const unsigned s0 = opaque_u32(0u), s1 = opaque_u32(1u << 16);
            compute(s0);
            store(1, ra1, rb1);
            load(kt + 3, ra1, rb1);
            __syncthreads();
            compute(s1);
            store(0, ra0, rb0);
            load(kt + 4, ra0, rb0);
            __syncthreads();
        }
        const int em0 = m0;
        const bool elive = nk > 0;
        const unsigned next = item + nx;
        const bool has_next = next < cnt_x;
        if (has_next) {                                      // next tile's first two k-tiles fly under the epilogue
            setup(next);
            if (EPI != EPI_RESID_LN && EPI != EPI_LNBWD) {   // (the row-wise epilogues request them themselves, below)
                load(0, ra0, rb0);
                load(1, ra1, rb1);
            }
        }
        bool prefetched = false;                             // EPI_RESID_LN: next tile requested from inside the epilogue
        if (elive) {
            // ---- epilogue: staged rows per pass: bf16 64 (both wm halves), fp32 32 (one wm half) -> <= 48 KiB
            constexpr bool STAGE_BF16 = (EPI == EPI_BF16);
            constexpr int ROWB = STAGE_BF16 ? GR_BN * 2 : GR_BN * 4;
            constexpr int WMP = STAGE_BF16 ? 2 : 1;          // wm halves staged together
            constexpr int SROWS = 32 * WMP;
            constexpr int CT = GR_BN / 8, RSTEP = GR_THREADS / CT;       // 48 column threads, 10 rows per step
            char* stg = smem;
            if (p.alpha != 1.0f) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[i][j][r] *= p.alpha;
            }
            // EPI_LNBWD: per-lane column sums (this lane's 3 x 4 columns) of this tile's rows; added to the workgroup's LDS
            // sums when the tile is done (live across the main loop they pushed the kernel over its 256 registers)
            f32x4v cs_dg[3], cs_db[3], cs_dbi[3];
#pragma unroll
            for (int c3 = 0; c3 < 3; ++c3) cs_dg[c3] = cs_db[c3] = cs_dbi[c3] = f32x4v{0.f, 0.f, 0.f, 0.f};
            const bool want_stats = EPI == EPI_BF16 && p.colsum != nullptr;
            float csum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            const int ct = t % CT, rr = t / CT;
            const int gn = 8 * ct;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
#pragma unroll
                for (int h = 0; h < 2 / WMP; ++h) {
                    const int sr = (WMP == 2 ? 32 * wm : 0) + lq;
                    // fused-LayerNorm epilogue: the next tile's first two k-tiles are requested after the LAST staging write
                    // below (all accumulators dead: their registers hold the pieces) - before the global stores of the last
                    // pass, so the loads do not queue behind those (gfx950 counts stores and loads on one in-order vmcnt).
                    // (Requesting the first k-tile two passes earlier, when half of the accumulators are dead, spills.)
                    // the residual rows of this pass are requested before the staging writes and the barrier, so their
                    // latency is not paid serially inside the row sweep
                    f32x4v rpre[2][3];
                    f32x4v gpre[2][3];                       // EPI_LNBWD: the rows of g that dx is added to
                    float mu_pre[2] = {0.f, 0.f}, rs_pre[2] = {0.f, 0.f};
                    if (EPI == EPI_LNBWD) {
#pragma unroll
                        for (int step = 0; step < 2; ++step) {
                            const int gm = em0 + 64 * h + 32 * q + 16 * step + 2 * w + hf;
                            if (gm < p.M) { mu_pre[step] = p.ln_mean[gm]; rs_pre[step] = p.ln_rstd[gm]; }
#pragma unroll
                            for (int c3 = 0; c3 < 3; ++c3) {
                                const int gnc = 4 * ((lane & 31) + 32 * c3);
                                rpre[step][c3] = gpre[step][c3] = f32x4v{0.f, 0.f, 0.f, 0.f};
                                if (gm < p.M && gnc < p.N) {
                                    rpre[step][c3] = *reinterpret_cast<const f32x4v*>(p.resid + (long)gm * p.ldr + gnc);
                                    if (p.lnb_accumulate)
                                        gpre[step][c3] = *reinterpret_cast<const f32x4v*>(reinterpret_cast<const float*>(p.C) + (long)gm * p.ldc + gnc);
                                }
                            }
                        }
                    }
                    float sc_pre[2] = {1.0f, 1.0f};                      // DropPath scale of the two rows (RESID_LN)
                    if (EPI == EPI_RESID_LN) {
#pragma unroll
                        for (int step = 0; step < 2; ++step) {
                            const int gm = em0 + 64 * h + 32 * q + 16 * step + 2 * w + hf;
                            if (gm < p.M && p.rowscale) sc_pre[step] = p.rowscale[p.rps_shift >= 0 ? gm >> p.rps_shift : gm / p.rows_per_sample];
#pragma unroll
                            for (int c3 = 0; c3 < 3; ++c3) {
                                const int gnc = 4 * ((lane & 31) + 32 * c3);
                                rpre[step][c3] = f32x4v{0.f, 0.f, 0.f, 0.f};
                                if (gm < p.M && gnc < p.N)
                                    rpre[step][c3] = *reinterpret_cast<const f32x4v*>(p.resid + (long)gm * p.ldr + gnc);
                            }
                        }
                    }
                    if (WMP == 2 || wm == h) {
#pragma unroll
                        for (int j = 0; j < 3; ++j)
#pragma unroll
                            for (int g = 0; g < 4; ++g) {
                                const int nl = 96 * wn + 32 * j + 8 * g + 4 * hf;
                                float v0 = acc[q][j][4 * g], v1 = acc[q][j][4 * g + 1], v2 = acc[q][j][4 * g + 2],
                                      v3 = acc[q][j][4 * g + 3];
                                if (STAGE_BF16) {
                                    {
                                        const f32x4v b = *reinterpret_cast<const f32x4v*>(vbias + nl);
                                        v0 += b.x; v1 += b.y; v2 += b.z; v3 += b.w;
                                    }
                                    u32x2 o;
                                    o.x = pack_bf2(v0, v1);
                                    o.y = pack_bf2(v2, v3);
                                    *reinterpret_cast<u32x2*>(stg + sr * ROWB + (((nl >> 3) ^ (sr & 15)) * 16) +
                                                              ((nl >> 2) & 1) * 8) = o;
                                } else {
                                    const f32x4v o = {v0, v1, v2, v3};
                                    *reinterpret_cast<f32x4v*>(stg + sr * ROWB + (((nl >> 2) ^ (sr & 15)) * 16)) = o;
                                }
                            }
                    }
                    constexpr bool fuse_ln = EPI == EPI_RESID_LN;
                    lds_barrier();
                    // (EPI_LNBWD requests nothing here: its row sweep needs the registers the next tile's k-tiles would occupy)
                    if (fuse_ln && q == 1 && h == 1 && has_next) {       // every accumulator has been staged
                        load(0, ra0, rb0);
                        prefetched = true;
                    }
                    if (EPI == EPI_LNBWD) {
                        // LayerNorm backward of the finished rows (layernorm.h: ln_bwd_kernel, same arithmetic): half a wave
                        // owns one row, the two row means by shuffles, column sums kept per lane until the kernel ends
                        const int L = lane & 31;
                        const float inv_n = 1.0f / (float)p.N;
#pragma unroll
                        for (int step = 0; step < 2; ++step) {
                            const int s2 = 16 * step + 2 * w + hf;
                            const int gm = em0 + 64 * h + 32 * q + s2;
                            const bool row_ok = gm < p.M;
                            const float mu = mu_pre[step], rs = rs_pre[step];
                            f32x4v d[3];                 // (xhat is recomputed in the second sweep: 12 registers matter here)
                            float s1 = 0.f, sq = 0.f;
#pragma unroll
                            for (int c3 = 0; c3 < 3; ++c3) {
                                const int chunk = L + 32 * c3, gnc = 4 * chunk;
                                d[c3] = f32x4v{0.f, 0.f, 0.f, 0.f};
                                if (row_ok && gnc < p.N) {
                                    const f32x4v dyv = *reinterpret_cast<const f32x4v*>(stg + s2 * ROWB + ((chunk ^ (s2 & 15)) * 16));
                                    const f32x4v ga = *reinterpret_cast<const f32x4v*>(vgamma + gnc);
                                    const f32x4v xh = (rpre[step][c3] - mu) * rs;
                                    d[c3] = dyv * ga;
                                    cs_dg[c3] += dyv * xh;
                                    cs_db[c3] += dyv;
                                    s1 += (d[c3].x + d[c3].y) + (d[c3].z + d[c3].w);
                                    const f32x4v e = d[c3] * xh;
                                    sq += (e.x + e.y) + (e.z + e.w);
                                }
                            }
#pragma unroll
                            for (int msk = 16; msk >= 1; msk >>= 1) { s1 += shfl_xor(s1, msk); sq += shfl_xor(sq, msk); }
                            s1 *= inv_n;
                            sq *= inv_n;
                            const float sc = (row_ok && p.lnb_gb && p.rowscale)
                                                 ? p.rowscale[p.rps_shift >= 0 ? gm >> p.rps_shift : gm / p.rows_per_sample] : 1.0f;
#pragma unroll
                            for (int c3 = 0; c3 < 3; ++c3) {
                                const int gnc = 4 * (L + 32 * c3);
                                if (row_ok && gnc < p.N) {
                                    const f32x4v xh = (rpre[step][c3] - mu) * rs;
                                    const f32x4v dx = (d[c3] - s1 - xh * sq) * rs + gpre[step][c3];
                                    *reinterpret_cast<f32x4v*>(reinterpret_cast<float*>(p.C) + (long)gm * p.ldc + gnc) = dx;
                                    if (p.lnb_gb) {
                                        const f32x4v o = dx * sc;
                                        u32x2 pk;
                                        pk.x = pack_bf2(o.x, o.y);
                                        pk.y = pack_bf2(o.z, o.w);
                                        *reinterpret_cast<u32x2*>(p.lnb_gb + (long)gm * p.ld_gb + gnc) = pk;
                                        // sum what the GEMMs will actually read (the bf16-rounded values)
                                        cs_dbi[c3] += f32x4v{bf_lo(pk.x), bf_hi(pk.x), bf_lo(pk.y), bf_hi(pk.y)};
                                    }
                                }
                            }
                        }
                    } else
                    if (fuse_ln) {
                        // residual epilogue + LayerNorm of the finished rows.  Half a wave owns one row (32 lanes x three
                        // 16-byte chunks = 384 columns): the row sums are five shuffles, no LDS traffic, one sweep.
                        const int L = lane & 31;
                        const float inv_n = 1.0f / (float)p.N;
                        // everything this pass loaded has arrived BEFORE its first store (prelude: needed_here): a wait behind a
                        // store is `vmcnt(0)` - the store's round trip, once per row (and the scale was loaded right there)
#pragma unroll
                        for (int step = 0; step < 2; ++step) {
                            needed_here(sc_pre[step]);
#pragma unroll
                            for (int c3 = 0; c3 < 3; ++c3) needed_here(rpre[step][c3]);
                        }
#pragma unroll
                        for (int step = 0; step < 2; ++step) {
                            const int s2 = 16 * step + 2 * w + hf;
                            const int gm = em0 + 64 * h + 32 * q + s2;
                            const bool row_ok = gm < p.M;
                            const float sc = sc_pre[step];
                            f32x4v o[3];
                            float s1 = 0.f, sq = 0.f;
#pragma unroll
                            for (int c3 = 0; c3 < 3; ++c3) {
                                const int chunk = L + 32 * c3, gnc = 4 * chunk;
                                o[c3] = f32x4v{0.f, 0.f, 0.f, 0.f};
                                if (row_ok && gnc < p.N) {
                                    const f32x4v v = *reinterpret_cast<const f32x4v*>(stg + s2 * ROWB + ((chunk ^ (s2 & 15)) * 16));
                                    const f32x4v r = rpre[step][c3];
                                    const f32x4v b = *reinterpret_cast<const f32x4v*>(vbias + gnc);
                                    o[c3] = r + (v + b) * sc;
                                    *reinterpret_cast<f32x4v*>(reinterpret_cast<float*>(p.C) + (long)gm * p.ldc + gnc) = o[c3];
                                    s1 += (o[c3].x + o[c3].y) + (o[c3].z + o[c3].w);
                                    sq += (o[c3].x * o[c3].x + o[c3].y * o[c3].y) + (o[c3].z * o[c3].z + o[c3].w * o[c3].w);
                                }
                            }
#pragma unroll
                            for (int msk = 16; msk >= 1; msk >>= 1) { s1 += shfl_xor(s1, msk); sq += shfl_xor(sq, msk); }
                            const float mean = s1 * inv_n;
                            float var = sq * inv_n - mean * mean;
                            var = var > 0.f ? var : 0.f;
                            const float rstd = 1.0f / sqrtf(var + p.ln_eps);
#pragma unroll
                            for (int c3 = 0; c3 < 3; ++c3) {
                                const int gnc = 4 * (L + 32 * c3);
                                if (row_ok && gnc < p.N) {
                                    const f32x4v ga = *reinterpret_cast<const f32x4v*>(vgamma + gnc);
                                    const f32x4v be = *reinterpret_cast<const f32x4v*>(vbeta + gnc);
                                    const f32x4v yv = (o[c3] - mean) * rstd * ga + be;
                                    u32x2 pk;
                                    pk.x = pack_bf2(yv.x, yv.y);
                                    pk.y = pack_bf2(yv.z, yv.w);
                                    *reinterpret_cast<u32x2*>(p.ln_y + (long)gm * p.ld_y + gnc) = pk;
                                }
                            }
                            if (L == 0 && row_ok) { p.ln_mean[gm] = mean; p.ln_rstd[gm] = rstd; }
                        }
                    } else
                    if (rr < RSTEP) {
                        for (int s2 = rr; s2 < SROWS; s2 += RSTEP) {
                            const int gm = em0 + 64 * (WMP == 2 ? (s2 >> 5) : h) + 32 * q + (s2 & 31);
                            if (gm < p.M && gn < p.N) {
                                if (STAGE_BF16) {
                                    const u32x4 wv = *reinterpret_cast<const u32x4*>(stg + s2 * ROWB + ((ct ^ (s2 & 15)) * 16));
                                    *reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(p.C) + (long)gm * p.ldc + gn) = wv;
                                    if (want_stats) {
                                        float v[8];
                                        unpack8(wv, v);
#pragma unroll
                                        for (int e = 0; e < 8; ++e) csum[e] += v[e];
                                    }
                                } else {
                                    float v[8];
                                    const f32x4v c0 = *reinterpret_cast<const f32x4v*>(stg + s2 * ROWB + (((2 * ct) ^ (s2 & 15)) * 16));
                                    const f32x4v c1 = *reinterpret_cast<const f32x4v*>(stg + s2 * ROWB + (((2 * ct + 1) ^ (s2 & 15)) * 16));
                                    v[0] = c0.x; v[1] = c0.y; v[2] = c0.z; v[3] = c0.w;
                                    v[4] = c1.x; v[5] = c1.y; v[6] = c1.z; v[7] = c1.w;
                                    gemm_epilogue_row8<(EPI == EPI_RESID_LN ? EPI_RESID : EPI)>(p, gm, gn, v);
                                }
                            }
                        }
                    }
                    lds_barrier();
                }
            }
            if (EPI == EPI_LNBWD) {
                // this tile's column sums -> the workgroup's LDS sums: the two half-waves of a wave by a shuffle, the eight
                // waves through the (now idle) staging image, then every thread owns a few (quantity, column) cells.
                // (LDS float atomics from all 16 half-waves onto the same 1152 cells measured 40 us per tile.)
                const int L = lane & 31;
                float* red = reinterpret_cast<float*>(stg);                      // [8 waves][3][GR_BN]
#pragma unroll
                for (int c3 = 0; c3 < 3; ++c3) {
                    f32x4v* q3[3] = {&cs_dg[c3], &cs_db[c3], &cs_dbi[c3]};
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        f32x4v v = *q3[k];
                        v.x += shfl_xor(v.x, 32); v.y += shfl_xor(v.y, 32); v.z += shfl_xor(v.z, 32); v.w += shfl_xor(v.w, 32);
                        if (hf == 0) *reinterpret_cast<f32x4v*>(red + (w * 3 + k) * GR_BN + 4 * (L + 32 * c3)) = v;
                    }
                }
                lds_barrier();
                for (int i = t; i < 3 * GR_BN; i += GR_THREADS) {
                    float a = 0.f;
#pragma unroll
                    for (int wv = 0; wv < 8; ++wv) a += red[wv * 3 * GR_BN + i];
                    lds_colsum[i] += a;
                }
                lds_barrier();                               // the staging image is rewritten by the next tile's first k-tile
            }
            if (want_stats) {
                float* red = reinterpret_cast<float*>(stg);
                if (rr < RSTEP) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) red[rr * GR_BN + 8 * ct + e] = csum[e];
                }
                lds_barrier();
                if (t < GR_BN && t < p.N) {
                    float a = 0.f;
#pragma unroll
                    for (int r = 0; r < RSTEP; ++r) a += red[r * GR_BN + t];
                    atomicAdd(p.colsum + t, a);
                }
                lds_barrier();
            }
        }
        if (!has_next) break;
        if (EPI == EPI_RESID_LN || EPI == EPI_LNBWD) {
            if (!prefetched) load(0, ra0, rb0);              // tile without work: nothing was requested in its epilogue
            load(1, ra1, rb1);
        }
        item = next;
    }
    if (EPI == EPI_LNBWD) {
        lds_barrier();                                       // every wave's LDS additions are in
        for (int c = t; c < p.N; c += GR_THREADS) {
            atomicAdd(p.lnb_dgamma + c, lds_colsum[c]);
            atomicAdd(p.lnb_dbeta + c, lds_colsum[GR_BN + c]);
            if (p.lnb_gb && p.lnb_dbias) atomicAdd(p.lnb_dbias + c, lds_colsum[2 * GR_BN + c]);
        }
    }
}

}  // namespace ccd
