// conv_gemm.hip -- implicit-GEMM convolution / linear layer on the fp32 matrix cores of gfx950.
//
//   C[out_row(m), n] = epilogue( sum_k A(m,k) * W[n,k] )          m < M, n < N, k < K
//
// A(m,k) is gathered on the fly from a channels-last activation tensor: m -> (batch, oy, ox),
// k -> (tap, channel); tap -> (dy,dx) with stride / padding / dilation / nearest-upsample of the
// input grid.  A linear layer is the 1-tap case.  This one kernel serves every dense contraction
// on the path (SURVEY K2,K3,K5,K6,K11,K12): U-Net / VAE 3x3 and 1x1 convolutions, all Linear
// layers, HiFi-GAN dilated Conv1d and (phase-decomposed) ConvTranspose1d, the STFT-as-DFT and the
// mel filterbank product.
//
// Arithmetic: v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain, 157 TF peak) -- the parity
// configuration of the reference, which runs strict fp32 (code/utils.py:113-116).
//
// Tiling: 256 threads = 4 waves; block tile BM x BN, K-chunk BKT staged through TWO LDS stages with a
// 4-float row pad (conflict-free ds_read_b128 fragment reads): chunk k+1 is written to the other stage
// while the MFMAs of chunk k run (one barrier per chunk), DEPTH further chunks are in flight in
// registers, and the fragments of k-block kb+1 are read before the MFMAs of kb issue.  Each lane reads 4 consecutive k per ds_read_b128;
// MFMA step s of a k-block uses k = {s, 4+s} (A and B agree, the sum over k is order-free).
//
// At the batch sizes of the edit loop (U-Net batch 2) most launches are a few microseconds of MFMA
// work, so the kernel is written for LATENCY: 32-bit offsets, no per-chunk integer divisions, one
// batch of residual loads in the epilogue (measured with the in-kernel s_memtime timeline, p.dbg).
#include "cg_params.h"

// PLAIN: the A operand needs no transform (no loader activation, no LayerNorm statistics) -- the common case.
// MINW: waves per SIMD the register allocation must leave room for (2 blocks per CU for the 128x128 tile).
template <int BM, int BN, int WROWS, int WCOLS, bool GENERIC, int DEPTH, int BKT, bool PLAIN>
__global__ __launch_bounds__(256, (BM * BN >= 128 * 128) ? 2 : 1) void conv_gemm_kernel(CGParams p) {
    constexpr int WM = BM / WROWS, WN = BN / WCOLS;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int LDS_LD = BKT + 4;
    constexpr int STAGE = (BM + BN) * LDS_LD;      // one operand stage (A tile then W tile)
    constexpr int TPR = BKT / 4;            // loader threads per row (float4 each)
    constexpr int RPP = 256 / TPR;          // rows per loader pass
    constexpr int PA = BM / RPP, PB = BN / RPP;
    constexpr int NKB = BKT / 8;
    static_assert(WROWS * WCOLS == 4, "4 waves");
    static_assert(TM >= 1 && TN >= 1 && PA >= 1 && PB >= 1, "tile");

    // two operand stages: chunk k+1 is written while the MFMAs of chunk k run -> one barrier per chunk
    __shared__ __attribute__((aligned(16))) float lds[2 * STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave / WCOLS, wc = wave % WCOLS;
    // XCD-aware tile order.  The dispatcher places workgroup b on XCD b % 8 (observed; a speed assumption only), and
    // each XCD has its own L2: with the plain order the N-tiles that share an A row panel land on 8 different L2s and
    // the panel is fetched 8 times (rocprofv3 FETCH_SIZE: 4-8x the algorithmic bytes, profiles/r01b_pmc_forward_B40.md).
    // Bijective remap (any grid size): XCD x gets a contiguous range of tile ids, n fastest.
    int tile_x = blockIdx.x, tile_y = blockIdx.y;
    if (gridDim.z == 1) {
        const unsigned nx = gridDim.x, nwg = nx * gridDim.y;
        const unsigned orig = blockIdx.y * nx + blockIdx.x;
        const unsigned xcd = orig & 7u, q = nwg >> 3, r = nwg & 7u;
        const unsigned id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
        tile_y = (int)(id / nx);
        tile_x = (int)(id - (unsigned)tile_y * nx);
    }
    const int m0 = tile_y * BM;
    const int n0 = tile_x * BN;
    const bool dbg_on = p.dbg && tid == 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0;
    int dbg_n = 0;
#define STAMP() do { if (dbg_on && dbg_n < 30) p.dbg[dbg_n++] = __builtin_amdgcn_s_memtime(); } while (0)
    STAMP();

    int kc_begin = 0, kc_end = p.nchunks;
    if (p.ksplit > 1) {
        int per = (p.nchunks + p.ksplit - 1) / p.ksplit;
        kc_begin = blockIdx.z * per;
        kc_end = min(p.nchunks, kc_begin + per);
    }

    const int lrow = tid / TPR;
    const int lcol = (tid % TPR) * 4;

    // per-row gather state (32-bit element offsets: every tensor on the path is < 2^31 elements)
    int ay0[PA], ax0[PA];
    unsigned abase[PA], abase2[PA];
    bool avalid[PA];
#pragma unroll
    for (int q = 0; q < PA; ++q) {
        const int m = m0 + lrow + RPP * q;
        avalid[q] = m < p.M;
        const int mm = avalid[q] ? m : 0;
        const int b = mm / p.rpb;
        const int r = mm - b * p.rpb;
        const int oy = r / p.OW;
        const int ox = r - oy * p.OW;
        ay0[q] = oy * p.stride - p.pad_h;
        ax0[q] = ox * p.stride - p.pad_w;
        abase[q] = (unsigned)b * (unsigned)p.a_bs + lcol;
        abase2[q] = (unsigned)b * (unsigned)p.a_bs2 + lcol;
    }
    unsigned wbase[PB];
    bool wvalid[PB];
#pragma unroll
    for (int q = 0; q < PB; ++q) {
        const int n = n0 + lrow + RPP * q;
        wvalid[q] = n < p.N;
        wbase[q] = (unsigned)(wvalid[q] ? n : 0) * (unsigned)p.K + lcol;
    }
    const int vIH = p.vIH, vIW = p.vIW;
    // running (tap, channel) position of the NEXT chunk to prefetch (prefetch() is always called with
    // consecutive kc): no per-chunk integer divisions
    // [group of p.kgroup channels][tap][chunk within the group] (cg_params.h): pf_cg = first channel of the group
    int pf_cg = 0, pf_sub = 0, pf_ty = 0, pf_tx = 0;
    const int gq = p.kgroup / BKT;            // chunks per (group, tap)
    if constexpr (!GENERIC) {
        const int per_group = p.KH * p.KW * gq;
        const int g = kc_begin / per_group;
        const int rem = kc_begin - g * per_group;
        const int tap = rem / gq;
        pf_cg = g * p.kgroup;
        pf_sub = rem - tap * gq;
        pf_ty = tap / p.KW;
        pf_tx = tap - pf_ty * p.KW;
    }

    float4 rbuf_a[DEPTH][PA], rbuf_b[DEPTH][PB];
    float ln_s1[PA], ln_s2[PA];   // fused LayerNorm: running sum / sum of squares of this thread's A rows
#pragma unroll
    for (int q = 0; q < PA; ++q) { ln_s1[q] = 0.f; ln_s2[q] = 0.f; }
    unsigned rmask[DEPTH];      // per stage: bit q set = A row q of that chunk is in-bounds (else zero padding)

    auto prefetch = [&](int kc, float4 (&ra)[PA], float4 (&rb)[PB], unsigned& mask) {
        const int k0 = min(kc, p.nchunks - 1) * BKT;      // dead prefetches past the end stay in bounds (generic path)
        if constexpr (!GENERIC) {
            const bool live = kc < p.nchunks;                 // a dead prefetch reads clamped addresses and is never staged
            int c0 = pf_cg + pf_sub * BKT;
            const int dy = pf_ty * p.dil_h, dx = pf_tx * p.dil_w;
            const unsigned kw = live ? (unsigned)((pf_ty * p.KW + pf_tx) * p.Cin + c0) : 0u;    // W column of this chunk
            if (++pf_sub == gq) {
                pf_sub = 0;
                if (++pf_tx == p.KW) {
                    pf_tx = 0;
                    if (++pf_ty == p.KH) { pf_ty = 0; pf_cg += p.kgroup; }
                }
            }
            // two-source A (an up-block concat that is never materialised): block-uniform select per chunk
            const float* src = p.A;
            unsigned ld = (unsigned)p.lda;
            const bool second = p.C1 > 0 && c0 >= p.C1;
            if (second) { src = p.A2; ld = (unsigned)p.lda2; c0 -= p.C1; }
            unsigned mk = 0;
#pragma unroll
            for (int q = 0; q < PA; ++q) {
                const int iy = ay0[q] + dy, ix = ax0[q] + dx;
                const bool ok = avalid[q] & ((unsigned)iy < (unsigned)vIH) & ((unsigned)ix < (unsigned)vIW) & live;
                // clamped, always-valid address; the zero padding is applied at the LDS write so that the
                // raw load stays in flight (nothing consumes it here)
                const int cy = ok ? (iy >> p.up) : 0, cx = ok ? (ix >> p.up) : 0;
                const unsigned base = second ? abase2[q] : abase[q];
                const unsigned off = base + (unsigned)(cy * p.IW + cx) * ld + c0;
                ra[q] = *reinterpret_cast<const float4*>(src + (ok ? off : (unsigned)lcol));
                mk |= ok ? (1u << q) : 0u;
            }
            mask = mk;
#pragma unroll
            for (int q = 0; q < PB; ++q) rb[q] = *reinterpret_cast<const float4*>(p.W + wbase[q] + kw);
        } else {
#pragma unroll
            for (int q = 0; q < PA; ++q) {
                float tmp[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int k = k0 + lcol + e;
                    float v = 0.f;
                    if (k < p.K && avalid[q]) {
                        const int tap = k / p.Cin;
                        const int c = k - tap * p.Cin;
                        const int ty = tap / p.KW;
                        const int tx = tap - ty * p.KW;
                        const int iy = ay0[q] + ty * p.dil_h, ix = ax0[q] + tx * p.dil_w;
                        if ((unsigned)iy < (unsigned)vIH && (unsigned)ix < (unsigned)vIW)
                            v = p.A[(size_t)(abase[q] - lcol) +
                                    ((size_t)(iy >> p.up) * p.IW + (size_t)(ix >> p.up)) * p.lda + c];
                    }
                    tmp[e] = v;
                }
                ra[q] = make_float4(tmp[0], tmp[1], tmp[2], tmp[3]);
            }
            mask = 0xffffffffu;
#pragma unroll
            for (int q = 0; q < PB; ++q) {
                float tmp[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int k = k0 + lcol + e;
                    tmp[e] = (wvalid[q] && k < p.K) ? p.W[(size_t)(wbase[q] - lcol) + k] : 0.f;
                }
                rb[q] = make_float4(tmp[0], tmp[1], tmp[2], tmp[3]);
            }
        }
    };

    // registers of one prefetched chunk -> an LDS stage (zero padding, loader transforms, LayerNorm statistics)
    auto stage_write = [&](float* st, const float4 (&ra)[PA], const float4 (&rb)[PB], unsigned mask) {
        float* As_ = st + lrow * LDS_LD + lcol;
        float* Ws_ = st + (BM + lrow) * LDS_LD + lcol;
#pragma unroll
        for (int q = 0; q < PA; ++q) {
            float4 v = ra[q];
            if (!((mask >> q) & 1u)) v = make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (!PLAIN) {
                if (p.ln_mode) {
                    ln_s1[q] += (v.x + v.y) + (v.z + v.w);
                    ln_s2[q] += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
                }
                if (p.in_act) {         // SiLU / LeakyReLU of the A operand; f(0) = 0 keeps the zero padding
                    v.x = in_transform(v.x, p.in_act, p.in_slope);
                    v.y = in_transform(v.y, p.in_act, p.in_slope);
                    v.z = in_transform(v.z, p.in_act, p.in_slope);
                    v.w = in_transform(v.w, p.in_act, p.in_slope);
                }
            }
            *reinterpret_cast<float4*>(As_ + RPP * q * LDS_LD) = v;
        }
#pragma unroll
        for (int q = 0; q < PB; ++q)
            *reinterpret_cast<float4*>(Ws_ + RPP * q * LDS_LD) = wvalid[q] ? rb[q] : make_float4(0.f, 0.f, 0.f, 0.f);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int fi = lane & 31;        // fragment row (A: m, B: n)
    const int fh = lane >> 5;        // k half
    const int a_frag_off = (wr * WM + fi) * LDS_LD + 4 * fh;
    const int b_frag_off = (BM + wc * WN + fi) * LDS_LD + 4 * fh;

    // the MFMAs of one staged chunk; the fragments of k-block kb+1 are fetched before the MFMAs of kb issue
    auto mfma_chunk = [&](const float* st) {
        const float* a_frag = st + a_frag_off;
        const float* b_frag = st + b_frag_off;
        float4 af[2][TM], bf[2][TN];
#pragma unroll
        for (int a = 0; a < TM; ++a) af[0][a] = *reinterpret_cast<const float4*>(a_frag + a * 32 * LDS_LD);
#pragma unroll
        for (int b = 0; b < TN; ++b) bf[0][b] = *reinterpret_cast<const float4*>(b_frag + b * 32 * LDS_LD);
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            const int c = kb & 1, nx = c ^ 1;
            if (kb + 1 < NKB) {
#pragma unroll
                for (int a = 0; a < TM; ++a)
                    af[nx][a] = *reinterpret_cast<const float4*>(a_frag + a * 32 * LDS_LD + (kb + 1) * 8);
#pragma unroll
                for (int b = 0; b < TN; ++b)
                    bf[nx][b] = *reinterpret_cast<const float4*>(b_frag + b * 32 * LDS_LD + (kb + 1) * 8);
            }
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[c][a].x, bf[c][b].x, acc[a][b], 0, 0, 0);
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[c][a].y, bf[c][b].y, acc[a][b], 0, 0, 0);
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[c][a].z, bf[c][b].z, acc[a][b], 0, 0, 0);
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[c][a].w, bf[c][b].w, acc[a][b], 0, 0, 0);
        }
    };

    STAMP();
    // every prefetch is issued unconditionally (chunks past the end read clamped addresses and are never staged):
    // the number of loads in flight is then static and the compiler's s_waitcnt keeps the newer chunks in flight
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) prefetch(kc_begin + d, rbuf_a[d], rbuf_b[d], rmask[d]);
    STAMP();
    // chunk kc_begin -> stage 0; its register slot takes chunk kc_begin + DEPTH
    stage_write(lds, rbuf_a[0], rbuf_b[0], rmask[0]);
    prefetch(kc_begin + DEPTH, rbuf_a[0], rbuf_b[0], rmask[0]);
    __syncthreads();
    STAMP();
    int cur = 0;
    for (int kc0 = kc_begin; kc0 < kc_end; kc0 += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int kc = kc0 + d;
            if (kc >= kc_end) break;
            // chunk kc+1 sits in register slot (d+1) % DEPTH: write it to the other stage, refill the slot
            {
                const int sl = (d + 1) % DEPTH;
                if (kc + 1 < kc_end) stage_write(lds + (cur ^ 1) * STAGE, rbuf_a[sl], rbuf_b[sl], rmask[sl]);
                prefetch(kc + 1 + DEPTH, rbuf_a[sl], rbuf_b[sl], rmask[sl]);
            }
            mfma_chunk(lds + cur * STAGE);
            __syncthreads();
            STAMP();
            cur ^= 1;
        }
    }

    // ---- fused LayerNorm: finish the per-row statistics (the loader threads of a row are TPR adjacent lanes)
    float* ln_stat = lds;                 // [BM][2] (mean, rstd); the operand tiles are dead after the last barrier
    if (p.ln_mode) {
#pragma unroll
        for (int q = 0; q < PA; ++q) {
            float s1 = ln_s1[q], s2 = ln_s2[q];
#pragma unroll
            for (int o = TPR / 2; o > 0; o >>= 1) {
                s1 += __shfl_xor(s1, o, 64);
                s2 += __shfl_xor(s2, o, 64);
            }
            if ((tid % TPR) == 0) {
                const float mean = s1 / (float)p.K;
                const float var = fmaxf(s2 / (float)p.K - mean * mean, 0.f);
                ln_stat[2 * (lrow + RPP * q)] = mean;
                ln_stat[2 * (lrow + RPP * q) + 1] = 1.0f / sqrtf(var + p.ln_eps);
            }
        }
        __syncthreads();
    }

    // ---- epilogue: acc[a][b][r] is C[row = (r&3) + 8*(r>>2) + 4*fh][col = fi] of the 32x32 tile.
    // Straight-line: every flag is kernel-uniform, every address is clamped in-bounds, so the residual /
    // previous-value loads of a tile issue as one batch (C may alias the residual: a per-element
    // load->store chain costs ~25k cycles).
    if constexpr (TN % 2 == 0) {
        if (p.geglu) {
            // FF1 of a transformer block with the GEGLU gate fused: the host packs W rows as [32 value | 32 gate] per 32
            // output features, so sub-tiles (b, b+1) of one wave hold value and gate of the same features
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; b += 2) {
                    const int nv = n0 + wc * WN + b * 32 + fi, ng = nv + 32;
                    const int mbase = m0 + wr * WM + a * 32 + 4 * fh;
                    if (ng >= p.N) continue;
                    const float bv = p.bias ? p.bias[nv] : 0.f, bg = p.bias ? p.bias[ng] : 0.f;
                    const float sv = p.ln_mode ? p.rowvec[nv] : 0.f, sg = p.ln_mode ? p.rowvec[ng] : 0.f;
                    const int nf = ((n0 + wc * WN + b * 32) >> 1) + fi;      // output feature column
                    const int gmb = min(mbase, p.M - 1), gb0 = gmb / p.rpb, gq0 = gmb - gb0 * p.rpb;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int dm = (r & 3) + 8 * (r >> 2);
                        const int m = mbase + dm;
                        float val = acc[a][b][r], gate = acc[a][b + 1][r];
                        if (p.ln_mode) {
                            const int lr = wr * WM + a * 32 + 4 * fh + dm;
                            const float mean = ln_stat[2 * lr], rstd = ln_stat[2 * lr + 1];
                            val = rstd * (val - mean * sv);
                            gate = rstd * (gate - mean * sg);
                        }
                        val += bv;
                        gate += bg;
                        if (m < p.M) {
                            int bb, q;                      // (no division per element: see conv_gemm_x6.hip)
                            if (p.rpb >= 32) {
                                q = gq0 + dm;
                                const bool wrap = q >= p.rpb;
                                bb = wrap ? gb0 + 1 : gb0;
                                q = wrap ? q - p.rpb : q;
                            } else {
                                bb = m / p.rpb;
                                q = m - bb * p.rpb;
                            }
                            const unsigned row = (unsigned)bb * (unsigned)p.out_bs + (unsigned)q;
                            p.C[row * (unsigned)p.ldc + nf] = val * glu_gate(gate, p.geglu);
                        }
                    }
                }
            return;
        }
    }
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const int n = n0 + wc * WN + b * 32 + fi;
            const int mbase = m0 + wr * WM + a * 32 + 4 * fh;
            if (n >= p.N) continue;
            if (p.ksplit > 1) {
                float* wsp = p.ws + ((size_t)blockIdx.z * p.M + mbase) * p.N + n;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int dm = (r & 3) + 8 * (r >> 2);
                    if (mbase + dm < p.M) wsp[(unsigned)dm * (unsigned)p.N] = acc[a][b][r];
                }
                continue;
            }
            const float bias_v = p.bias ? p.bias[n] : 0.f;
            unsigned rows[16];
            bool ok[16];
            {
                const int mb = min(mbase, p.M - 1);
                const int b0 = mb / p.rpb;
                const int q0 = mb - b0 * p.rpb;
                const int bmax = (p.M - 1) / p.rpb;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int dm = (r & 3) + 8 * (r >> 2);
                    int bb, q;
                    if (p.rpb >= 32) {                 // at most one batch-item wrap inside a 32-row tile
                        q = q0 + dm;
                        const bool wrap = q >= p.rpb;
                        bb = wrap ? b0 + 1 : b0;
                        q = wrap ? q - p.rpb : q;
                    } else {
                        const int mm = min(mbase + dm, p.M - 1);
                        bb = mm / p.rpb;
                        q = mm - bb * p.rpb;
                    }
                    const int o = q * p.o_mul + p.o_add;
                    ok[r] = (mbase + dm) < p.M && (unsigned)o < (unsigned)p.o_len;
                    rows[r] = (unsigned)min(bb, bmax) * (unsigned)p.out_bs + (unsigned)min(max(o, 0), p.o_len - 1);
                }
            }
            float val[16];
            if (p.ln_mode) {       // LN(x).W = rstd*(x.W' - mean*sum_k W') + W.beta   (W' = W*gamma, folded on the host)
                const float sn = p.rowvec[n];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int lr = wr * WM + a * 32 + 4 * fh + (r & 3) + 8 * (r >> 2);
                    val[r] = ln_stat[2 * lr + 1] * (acc[a][b][r] - ln_stat[2 * lr] * sn) + bias_v;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) val[r] = acc[a][b][r] + bias_v;
            }
            if (p.rowvec && !p.ln_mode) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    val[r] += p.rowvec[(rows[r] / (unsigned)p.out_bs) * (unsigned)p.ld_rv + n];
            }
            if (p.res) {
                float rv[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) rv[r] = p.res[rows[r] * (unsigned)p.ldr + n];
#pragma unroll
                for (int r = 0; r < 16; ++r) val[r] += rv[r];
            }
            if (p.out_act != AED_ACT_NONE) {
#pragma unroll
                for (int r = 0; r < 16; ++r) val[r] = aed_apply_act(val[r], p.out_act, p.out_p);
            }
            if (p.accumulate) {
                float pv[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) pv[r] = p.C[rows[r] * (unsigned)p.ldc + n];
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    val[r] = (p.accumulate == 1) ? val[r] + pv[r] : (pv[r] + val[r]) / p.out_div;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (ok[r]) p.C[rows[r] * (unsigned)p.ldc + n] = val[r];
        }
    STAMP();
    if (dbg_on) p.dbg[31] = dbg_n;
#undef STAMP
}

__global__ __launch_bounds__(256) void splitk_reduce_kernel(CGParams p) {
    size_t total = (size_t)p.M * p.N;
    if ((p.N & 3) == 0) {
        // four consecutive columns per thread, the slabs' loads issued back to back (round 5: the scalar loop below was a chain of
        // dependent 4-byte loads -- 5-8 us for a 128 x 640 output, as long as the split GEMM it finishes).  Same summation
        // order per element (z ascending), so the two paths are bit-identical.
        const size_t q4 = total >> 2;
        const float4* ws4 = reinterpret_cast<const float4*>(p.ws);
        for (size_t e4 = (size_t)blockIdx.x * 256 + threadIdx.x; e4 < q4; e4 += (size_t)gridDim.x * 256) {
            float4 v = ws4[e4];
#pragma unroll 4
            for (int z = 1; z < p.ksplit; ++z) {
                const float4 w = ws4[(size_t)z * q4 + e4];
                v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
            }
            const size_t e = e4 << 2;
            const int m = (int)(e / p.N);
            const int n = (int)(e - (size_t)m * p.N);
            store_out(p, m, n, v.x);
            store_out(p, m, n + 1, v.y);
            store_out(p, m, n + 2, v.z);
            store_out(p, m, n + 3, v.w);
        }
        return;
    }
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        int m = (int)(e / p.N);
        int n = (int)(e - (size_t)m * p.N);
        float v = 0.f;
        for (int z = 0; z < p.ksplit; ++z) v += p.ws[(size_t)z * total + e];
        store_out(p, m, n, v);
    }
}

int cg_fill_params(const aed_op* op, CGParams& p, int bkt) {
    p.A = (const float*)op->p[0];
    p.W = (const float*)op->p[1];
    p.bias = (const float*)op->p[2];
    p.C = (float*)op->p[3];
    p.res = (const float*)op->p[4];
    p.rowvec = (const float*)op->p[5];
    p.ws = (float*)op->p[6];
    p.dbg = (op->flags & 1) ? (long long*)op->p[7] : nullptr;
    p.late_epilogue = (op->flags & 2) ? 1 : 0;
    const int32_t* i = op->i;
    p.M = i[0]; p.N = i[1]; p.K = i[2]; p.lda = i[3]; p.ldc = i[4]; p.ldr = i[5]; p.ld_rv = i[6];
    p.IH = i[7]; p.IW = i[8]; p.OH = i[9]; p.OW = i[10]; p.Cin = i[11]; p.KH = i[12]; p.KW = i[13];
    p.stride = i[14]; p.pad_h = i[15]; p.pad_w = i[16]; p.dil_h = i[17]; p.dil_w = i[18]; p.up = i[19];
    p.a_bs = i[20]; p.o_mul = i[21]; p.o_add = i[22]; p.o_len = i[23]; p.out_bs = i[24];
    p.in_act = i[25]; p.out_act = i[26]; p.accumulate = i[27]; p.ksplit = i[28];
    p.in_slope = op->f[0]; p.out_p = op->f[1]; p.out_div = op->f[2];
    p.ln_mode = i[31]; p.ln_eps = op->f[3];
    p.C1 = i[32]; p.lda2 = i[33]; p.a_bs2 = i[34]; p.geglu = i[35];
    p.A2 = (const float*)op->p[8];
    p.gm = 0;
    p.diag = (op->flags & 0x8000) ? 1 : 0;
    p.Wq = nullptr;
    p.Wsc = nullptr;
    p.sm_group = i[36]; p.w_bs = i[37]; p.vec_ld = i[38] > 0 ? i[38] : 1; p.vec_bs = i[39];
    p.sm_scale = op->f[4];
    p.kbias = (const float*)op->p[9];
    p.rpb = p.OH * p.OW;
    p.nchunks = (p.K + bkt - 1) / bkt;
    // op flag bit 5 (32): keep the tap-major K order (A/B switch for the traffic measurement)
    {
        const int g = bkt > 32 ? bkt : 32;
        p.kgroup = (p.KH * p.KW > 1 && p.Cin % g == 0 && p.Cin > g && !(op->flags & 32)) ? g : p.Cin;
    }
    p.vIH = p.IH << p.up;
    p.vIW = p.IW << p.up;
    if (p.up) {
        // nearest resize to an explicit target (diffusers' forward_upsample_size, models.py:186-188): the target is
        // the next skip's size, 2*IH or 2*IH-1; floor(dst*IH/target) == dst>>1 for both, so only the grid bound changes
        const int th = (p.OH - 1) * p.stride - 2 * p.pad_h + p.dil_h * (p.KH - 1) + 1;
        const int tw = (p.OW - 1) * p.stride - 2 * p.pad_w + p.dil_w * (p.KW - 1) + 1;
        if (th < p.vIH) p.vIH = th;
        if (tw < p.vIW) p.vIW = tw;
        AED_REQUIRE(p.vIH >= 2 * p.IH - 1 && p.vIW >= 2 * p.IW - 1, "conv_gemm: upsample target %dx%d too small for %dx%d",
                    th, tw, p.IH, p.IW);
    }
    AED_REQUIRE(p.A && p.W && p.C, "conv_gemm: null operand");
    AED_REQUIRE(p.M > 0 && p.N > 0 && p.K > 0, "conv_gemm: bad shape M=%d N=%d K=%d", p.M, p.N, p.K);
    AED_REQUIRE(p.K == p.KH * p.KW * p.Cin, "conv_gemm: K=%d != KH*KW*Cin=%d", p.K, p.KH * p.KW * p.Cin);
    AED_REQUIRE(p.rpb > 0 && p.M % p.rpb == 0, "conv_gemm: M=%d not a multiple of OH*OW=%d", p.M, p.rpb);
    AED_REQUIRE((long long)(p.M / p.rpb) * p.a_bs + (long long)p.IH * p.IW * p.lda < (1LL << 31) &&
                    (long long)p.N * p.K < (1LL << 31) &&
                    ((long long)(p.M / p.rpb) * p.out_bs + 1) * (long long)(p.ldc > p.ldr ? p.ldc : p.ldr) < (1LL << 31),
                "conv_gemm: operand exceeds the 32-bit element-offset range");
    if (p.ksplit < 1) p.ksplit = 1;
    if (p.ksplit > p.nchunks) p.ksplit = p.nchunks;
    if (p.ksplit > 1) AED_REQUIRE(p.ws != nullptr, "conv_gemm: split-K needs a workspace");
    if (p.ln_mode)
        AED_REQUIRE(p.ksplit == 1 && p.KH * p.KW == 1 && p.rowvec && p.bias && p.in_act == 0,
                    "conv_gemm: fused LayerNorm needs a 1-tap, unsplit GEMM with folded weights");
    if (p.C1 > 0)
        AED_REQUIRE(p.A2 && p.C1 % 64 == 0 && p.C1 < p.Cin && p.lda2 % 4 == 0 && ((uintptr_t)p.A2 % 16) == 0 &&
                        (long long)(p.M / p.rpb) * p.a_bs2 + (long long)p.IH * p.IW * p.lda2 < (1LL << 31),
                    "conv_gemm: bad two-source split C1=%d of Cin=%d", p.C1, p.Cin);
    if (p.geglu)
        AED_REQUIRE(p.ksplit == 1 && p.N % 64 == 0 && !p.res && p.out_act == 0 && p.accumulate == 0 && p.o_mul == 1 &&
                        p.o_add == 0 && (p.ln_mode || !p.rowvec),
                    "conv_gemm: the GEGLU epilogue needs an unsplit GEMM with packed N %% 64 == 0 and a plain epilogue");
    return 0;
}

template <int BM, int BN, int WR, int WC, int DEPTH, int BKT>
static void launch_cfg(const CGParams& p, bool plain, hipStream_t s) {
    dim3 grid(aed_cdiv(p.N, BN), aed_cdiv(p.M, BM), p.ksplit);
    if (plain)
        hipLaunchKernelGGL((conv_gemm_kernel<BM, BN, WR, WC, false, DEPTH, BKT, true>), grid, dim3(256), 0, s, p);
    else
        hipLaunchKernelGGL((conv_gemm_kernel<BM, BN, WR, WC, false, DEPTH, BKT, false>), grid, dim3(256), 0, s, p);
}

template <int BM, int BN, int WR, int WC>
static void launch_generic(const CGParams& p, hipStream_t s) {
    dim3 grid(aed_cdiv(p.N, BN), aed_cdiv(p.M, BM), p.ksplit);
    hipLaunchKernelGGL((conv_gemm_kernel<BM, BN, WR, WC, true, 1, 32, false>), grid, dim3(256), 0, s, p);
}

// tile_cfg: 0 auto, 1 = 128x128, 2 = 128x64, 3 = 64x128, 4 = 64x64, 5 = 128x32, 6 = 32x128, 7 = wave-split-K 32x32
// i[30] = 1 forces the 32-wide K chunk (A/B testing)
int launch_conv_gemm(const aed_op* op, hipStream_t s) {
    CGParams p;
    const int32_t* i = op->i;
    const int Cin = i[11];
    const bool generic = (Cin % 32 != 0) || (i[3] % 4 != 0) || ((uintptr_t)op->p[0] % 16 != 0) ||
                         ((uintptr_t)op->p[1] % 16 != 0);
    int cfg = i[29];
    if (cfg >= 10) {                 // latency-regime kernels (lin_gemm.hip)
        int rc = cg_fill_params(op, p, 32);
        if (rc) return rc;
        return launch_lin_gemm(p, cfg, s);
    }
    if (cfg == 0) {
        const int cus = aed_num_cus();
        const int M = i[0], N = i[1], ks = i[28] > 1 ? i[28] : 1;
        auto blocks = [&](int bm, int bn) { return (long)aed_cdiv(M, bm) * aed_cdiv(N, bn) * ks; };
        if (N <= 32) cfg = 5;
        else if (M <= 32) cfg = 6;
        else if (blocks(128, 128) >= (long)cus && N >= 128) cfg = 1;
        else if (blocks(128, 64) >= 2L * cus && N >= 64) cfg = 2;
        else cfg = 4;
        if (N < 64 && cfg != 5) cfg = 5;
    }
    if (generic && (cfg == 1 || cfg == 2 || cfg == 3)) cfg = 4;     // the scalar-gather path only exists for the small tiles
    // K chunk: 32 (two LDS stages of a 128x128 tile = 72 KB -> two blocks per CU); 64 for the 64x64 tile when the
    // channel count allows (half the barriers, still two blocks per CU)
    const bool bk64 = !generic && (Cin % 64 == 0) && i[30] != 1 && cfg == 4;
    int rc = cg_fill_params(op, p, bk64 ? 64 : 32);
    if (rc) return rc;
    const bool plain = p.in_act == 0 && p.ln_mode == 0;
    AED_REQUIRE(p.sm_group == 0 && p.w_bs == 0 && p.vec_bs == 0 && p.vec_ld == 1,
                "conv_gemm: per-batch weights / grouped softmax exist only in the lin_gemm kernels (tile >= 10)");
    if (p.geglu) AED_REQUIRE(cfg == 1 || cfg == 3, "conv_gemm: the GEGLU epilogue needs 64-wide wave tiles: 128x128 or 64x128 (cfg %d)", cfg);
    if (p.C1 > 0) AED_REQUIRE(!generic, "conv_gemm: two-source A needs the vector path of a tiled kernel");
    switch (cfg) {
        case 1: launch_cfg<128, 128, 2, 2, 2, 32>(p, plain, s); break;
        case 2: launch_cfg<128, 64, 2, 2, 2, 32>(p, plain, s); break;
        case 3: launch_cfg<64, 128, 2, 2, 2, 32>(p, plain, s); break;      // 2x more workgroups than 128x128, same 64-wide wave tiles
        case 4:
            if (generic) launch_generic<64, 64, 2, 2>(p, s);
            else if (bk64) launch_cfg<64, 64, 2, 2, 2, 64>(p, plain, s);
            else launch_cfg<64, 64, 2, 2, 3, 32>(p, plain, s);
            break;
        case 5:
            if (generic) launch_generic<128, 32, 4, 1>(p, s);
            else launch_cfg<128, 32, 4, 1, 3, 32>(p, plain, s);
            break;
        case 6:
            if (generic) launch_generic<32, 128, 1, 4>(p, s);
            else launch_cfg<32, 128, 1, 4, 3, 32>(p, plain, s);
            break;
        default: AED_REQUIRE(false, "conv_gemm: bad tile cfg %d", cfg);
    }
    AED_CHECK_HIP(hipGetLastError());
    if (p.ksplit > 1) {
        size_t total = (size_t)p.M * p.N;
        int grid = (int)((((p.N & 3) == 0 ? total >> 2 : total) + 255) / 256);
        if (grid > 2048) grid = 2048;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(grid), dim3(256), 0, s, p);
        AED_CHECK_HIP(hipGetLastError());
    }
    return 0;
}

int launch_splitk_reduce_slices(const aed_op* op, int slices, hipStream_t s) {
    CGParams p;
    int rc = cg_fill_params(op, p, 32);
    if (rc) return rc;
    if (slices > 0) p.ksplit = slices;      // the producer's own clamp (a kernel with wider k chunks has fewer of them)
    size_t total = (size_t)p.M * p.N;
    int grid = (int)((((p.N & 3) == 0 ? total >> 2 : total) + 255) / 256);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(grid), dim3(256), 0, s, p);
    AED_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_splitk_reduce(const aed_op* op, hipStream_t s) { return launch_splitk_reduce_slices(op, 0, s); }
