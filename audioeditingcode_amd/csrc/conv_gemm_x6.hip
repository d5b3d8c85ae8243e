// conv_gemm_x6.hip -- the implicit-GEMM convolution / linear layer of conv_gemm.hip with SPLIT-BF16 arithmetic.
//
// EXPERIMENTAL (round 3): selected per op by flag bit 2 of an AED_OP_CONV_GEMM record (include/aed.h); nothing on the
// product path sets it yet.  Same record, same operands, same epilogue -- only the contraction differs.
//
// Why: the fp32-input MFMA (v_mfma_f32_32x32x2_f32) runs at the fp32 VECTOR rate, 1/16 of the bf16 MFMA
// (MI355X_MICROARCH.md: 157 TF vs 2.5 PF).  An fp32 value is exactly the sum of three bf16 values (24 significand bits
// = 3 x 8): a = a0 + a1 + a2, |a1| <= 2^-8 |a|, |a2| <= 2^-16 |a|.  A product a*b is then the sum of nine piece products
// a_i*b_j, each EXACT in fp32 (8 x 8 bits), of relative size 2^-8(i+j).  Keeping the six with i+j <= 2 drops terms below
// 2^-23 |a||b| -- under the rounding of the fp32 accumulation itself (tools/bf16_split_study.py: vs fp64 the six-term sum
// is as close as a plain fp32 GEMM, 1.3e-7 .. 3.5e-7 rel L2 at K = 320 .. 5760; x9 brings nothing over x6).  Six
// v_mfma_f32_32x32x16_bf16 (32 cycles each per SIMD) replace eight v_mfma_f32_32x32x2_f32 (64 cycles each) per 16 k:
// 192 vs 512 matrix-pipe cycles, a 2.67x higher roof (419 TF/s fp32-equivalent) for the same fp32 operands in HBM.
//
// Data path: operands stay fp32 in HBM (no second copy of weights or activations).  The loader threads split each
// float4 they fetched into three packed-bf16 pieces (v_cvt_pk_bf16_f32, RNE; 4.5 VALU per element) while writing the
// LDS stage.  LDS row = [hi 16 k | mid 16 k | lo 16 k | 16 B pad] = 112 B: the fragment read of one piece is one
// ds_read_b128 per lane (8 consecutive k), conflict-free with this stride (28 dwords: the 16 rows of a b128 lane group
// land on 16 distinct 4-bank windows).  A and W use the same k -> (lane half, element) map, so the result does not depend
// on the instruction's internal k order; the C/D map is the one of the fp32 32x32 MFMA (conv_gemm.hip's epilogue).
//
// Pipeline (as conv_gemm.hip): two LDS stages, chunk k+1 is split + written to the other stage while the MFMAs of
// chunk k run, one barrier per chunk, DEPTH further chunks in flight in registers.  A chunk is 16 k = one bf16 k-block.
#include "cg_params.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

int launch_conv_gemm_x6(const aed_op* op, hipStream_t s);

constexpr int X6_BK = 16;           // fp32 k per chunk
constexpr int X6_ROWQ = 7;          // uint4 (16 B) per LDS row: 3 pieces x 2 + 1 pad

// (x0, x1) -> packed bf16 pieces {hi, mid, lo}; x == hi + mid + lo exactly (element 0 in the low half)
__device__ __forceinline__ void x6_split_pair(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    const f32x2 x = {x0, x1};
    h = __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf16x2));
    const f32x2 r = {x0 - __uint_as_float(h << 16), x1 - __uint_as_float(h & 0xffff0000u)};
    m = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
    const f32x2 q = {r[0] - __uint_as_float(m << 16), r[1] - __uint_as_float(m & 0xffff0000u)};
    l = __builtin_bit_cast(unsigned, __builtin_convertvector(q, bf16x2));
}

// SCHED: 0 = leave the instruction order to the compiler; 1 = ask for MFMA / VALU / DS interleave in the main loop
// NTERMS: 6 = the product arithmetic; 3 = only the terms of relative size >= 2^-8 (DIAGNOSTIC: tells how much of the kernel time
// is matrix-pipe time; ~4e-6 rel error, never used by a product path)
template <int BM, int BN, int WROWS, int WCOLS, int DEPTH, bool PLAIN, int SCHED, int NTERMS>
__global__ __launch_bounds__(64 * WROWS * WCOLS, 2) void conv_gemm_x6_kernel(CGParams p) {
    constexpr int NT = 64 * WROWS * WCOLS;
    constexpr int WM = BM / WROWS, WN = BN / WCOLS;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int TPR = X6_BK / 4;          // loader threads per row (float4 each)
    constexpr int RPP = NT / TPR;           // rows per loader pass
    constexpr int PA = BM / RPP, PB = BN / RPP;
    constexpr int STAGE = (BM + BN) * X6_ROWQ;      // uint4 per operand stage (A rows then W rows)
    static_assert(TM >= 1 && TN >= 1 && PA >= 1 && PB >= 1, "tile");
    static_assert(BM % RPP == 0 && BN % RPP == 0, "loader passes");

    __shared__ uint4 lds[2 * STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave / WCOLS, wc = wave % WCOLS;
    // XCD-aware tile order (see conv_gemm.hip): XCD x gets a contiguous range of tile ids, n fastest
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

    int kc_begin = 0, kc_end = p.nchunks;
    if (p.ksplit > 1) {
        const int per = (p.nchunks + p.ksplit - 1) / p.ksplit;
        kc_begin = blockIdx.z * per;
        kc_end = min(p.nchunks, kc_begin + per);
    }

    const int lrow = tid / TPR;
    const int lq = tid % TPR;               // which float4 of the 16-k row
    const int lcol = lq * 4;

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
    // running (tap, channel) position of the next chunk to prefetch: prefetch() is called with consecutive kc
    int pf_c0, pf_ty, pf_tx;
    {
        const int k0 = kc_begin * X6_BK;
        const int tap = k0 / p.Cin;
        pf_c0 = k0 - tap * p.Cin;
        pf_ty = tap / p.KW;
        pf_tx = tap - pf_ty * p.KW;
    }

    float4 rbuf_a[DEPTH][PA], rbuf_b[DEPTH][PB];
    unsigned rmask[DEPTH];
    float ln_s1[PA], ln_s2[PA];
#pragma unroll
    for (int q = 0; q < PA; ++q) { ln_s1[q] = 0.f; ln_s2[q] = 0.f; }

    // Branch-free on purpose: the whole chunk body (split of chunk k+1, loads of chunk k+1+DEPTH, MFMAs of chunk k) is ONE basic
    // block, so the scheduler may put VALU / LDS / VMEM work into the shadow of the MFMAs.
    auto prefetch = [&](int kc, float4 (&ra)[PA], float4 (&rb)[PB], unsigned& mask) {
        const int k0 = min(kc, p.nchunks - 1) * X6_BK;      // dead prefetches past the end stay in bounds
        int c0 = pf_c0;
        const int dy = pf_ty * p.dil_h, dx = pf_tx * p.dil_w;
        pf_c0 += X6_BK;
        const bool wrap = pf_c0 >= p.Cin;
        pf_c0 = wrap ? 0 : pf_c0;
        pf_tx += wrap ? 1 : 0;
        const bool wrap2 = pf_tx == p.KW;
        pf_tx = wrap2 ? 0 : pf_tx;
        pf_ty += wrap2 ? 1 : 0;
        const bool second = p.C1 > 0 && c0 >= p.C1;         // two-source A: block-uniform select per chunk
        const float* src = second ? p.A2 : p.A;
        const unsigned ld = second ? (unsigned)p.lda2 : (unsigned)p.lda;
        c0 -= second ? p.C1 : 0;
        unsigned mk = 0;
#pragma unroll
        for (int q = 0; q < PA; ++q) {
            const int iy = ay0[q] + dy, ix = ax0[q] + dx;
            const bool ok = avalid[q] & ((unsigned)iy < (unsigned)vIH) & ((unsigned)ix < (unsigned)vIW);
            const unsigned okm = ok ? 0xffffffffu : 0u;
            // out-of-image taps read element lcol of the tensor (always valid); the zero padding is applied at the LDS write
            const unsigned base = second ? abase2[q] : abase[q];
            const unsigned off = base + (unsigned)((iy >> p.up) * p.IW + (ix >> p.up)) * ld + (unsigned)c0;
            ra[q] = *reinterpret_cast<const float4*>(src + ((off & okm) | ((unsigned)lcol & ~okm)));
            mk |= okm & (1u << q);
        }
        mask = mk;
#pragma unroll
        for (int q = 0; q < PB; ++q) rb[q] = *reinterpret_cast<const float4*>(p.W + wbase[q] + k0);
    };

    // one fetched float4 -> three 8-byte piece groups of its LDS row
    auto put_row = [&](uint4* st, int row, float4 v) {
        unsigned h0, m0_, l0, h1, m1, l1;
        x6_split_pair(v.x, v.y, h0, m0_, l0);
        x6_split_pair(v.z, v.w, h1, m1, l1);
        uint2* dst = reinterpret_cast<uint2*>(st + row * X6_ROWQ) + lq;      // piece pl starts at uint2 index 4 * pl
        dst[0] = make_uint2(h0, h1);
        dst[4] = make_uint2(m0_, m1);
        dst[8] = make_uint2(l0, l1);
    };

    // live == false: the chunk does not exist (past the end of this block's k range) -> the stage is written as zeros
    auto stage_write = [&](uint4* st, const float4 (&ra)[PA], const float4 (&rb)[PB], unsigned mask, bool live) {
        mask = live ? mask : 0u;
#pragma unroll
        for (int q = 0; q < PA; ++q) {
            float4 v = ra[q];
            if (!((mask >> q) & 1u)) v = make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (!PLAIN) {
                if (p.ln_mode) {
                    ln_s1[q] += (v.x + v.y) + (v.z + v.w);
                    ln_s2[q] += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
                }
                if (p.in_act) {
                    v.x = in_transform(v.x, p.in_act, p.in_slope);
                    v.y = in_transform(v.y, p.in_act, p.in_slope);
                    v.z = in_transform(v.z, p.in_act, p.in_slope);
                    v.w = in_transform(v.w, p.in_act, p.in_slope);
                }
            }
            put_row(st, lrow + RPP * q, v);
        }
#pragma unroll
        for (int q = 0; q < PB; ++q)
            put_row(st, BM + lrow + RPP * q, (wvalid[q] && live) ? rb[q] : make_float4(0.f, 0.f, 0.f, 0.f));
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int fi = lane & 31;        // fragment row (A: m, W: n)
    const int fh = lane >> 5;        // which 8 of the 16 k
    const int a_frag = (wr * WM + fi) * X6_ROWQ + fh;
    const int b_frag = (BM + wc * WN + fi) * X6_ROWQ + fh;

    bf16x8 af[TM][3], bw[TN][3];
    auto load_frags = [&](const uint4* st) {
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                af[a][pl] = __builtin_bit_cast(bf16x8, st[a_frag + a * 32 * X6_ROWQ + 2 * pl]);
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                bw[b][pl] = __builtin_bit_cast(bf16x8, st[b_frag + b * 32 * X6_ROWQ + 2 * pl]);
    };
    // the six piece products with i + j <= 2, smallest first
    auto do_mfmas = [&]() {
        constexpr int TA[6] = {0, 2, 1, 0, 1, 0};
        constexpr int TB[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
        for (int t = 6 - NTERMS; t < 6; ++t)
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][TA[t]], bw[b][TB[t]], acc[a][b], 0, 0, 0);
    };

#pragma unroll
    for (int d = 0; d < DEPTH; ++d) prefetch(kc_begin + d, rbuf_a[d], rbuf_b[d], rmask[d]);
    stage_write(lds, rbuf_a[0], rbuf_b[0], rmask[0], true);
    prefetch(kc_begin + DEPTH, rbuf_a[0], rbuf_b[0], rmask[0]);
    __syncthreads();
    int cur = 0;
    // The trip count is rounded up to a multiple of DEPTH (static register slots, no exit inside the unrolled body): a chunk
    // past kc_end is staged as zeros and its MFMAs add nothing.  Per iteration, in program order: fragment reads of chunk kc
    // (stage cur) -> split + LDS write of chunk kc+1 (other stage) -> loads of chunk kc+1+DEPTH -> MFMAs of chunk kc -> barrier.
    for (int kc0 = kc_begin; kc0 < kc_end; kc0 += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int kc = kc0 + d;
            const int sl = (d + 1) % DEPTH;     // chunk kc+1 sits in this register slot
            load_frags(lds + cur * STAGE);
            stage_write(lds + (cur ^ 1) * STAGE, rbuf_a[sl], rbuf_b[sl], rmask[sl], kc + 1 < kc_end);
            prefetch(kc + 1 + DEPTH, rbuf_a[sl], rbuf_b[sl], rmask[sl]);
            do_mfmas();
            if constexpr (SCHED == 1) {
                // one wave per SIMD hides ~5 single-issue instructions per 32-cycle MFMA: spread the split of chunk kc+1
                // (4.5 VALU per element), its LDS writes and the next loads between the MFMAs of chunk kc
                constexpr int NM = TM * TN * NTERMS;
#pragma unroll
                for (int g = 0; g < NM; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);                      // 4 VALU
                    if (g % 3 == 2) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);      // 1 DS write
                    if (g % 6 == 5) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);      // 1 VMEM read
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                      // 1 MFMA
                }
            }
            __syncthreads();
            cur ^= 1;
        }
    }

    // ---- fused LayerNorm: per-row statistics (the loader threads of a row are TPR adjacent lanes)
    float* ln_stat = reinterpret_cast<float*>(lds);      // [BM][2] (mean, rstd); the operand stages are dead
    if (p.ln_mode) {
#pragma unroll
        for (int q = 0; q < PA; ++q) {
            float s1 = ln_s1[q], s2 = ln_s2[q];
#pragma unroll
            for (int o = TPR / 2; o > 0; o >>= 1) {
                s1 += __shfl_xor(s1, o, 64);
                s2 += __shfl_xor(s2, o, 64);
            }
            if (lq == 0) {
                const float mean = s1 / (float)p.K;
                const float var = fmaxf(s2 / (float)p.K - mean * mean, 0.f);
                ln_stat[2 * (lrow + RPP * q)] = mean;
                ln_stat[2 * (lrow + RPP * q) + 1] = 1.0f / sqrtf(var + p.ln_eps);
            }
        }
        __syncthreads();
    }

    // ---- epilogue (the semantics of conv_gemm.hip's): acc[a][b][r] = C[row (r&3) + 8*(r>>2) + 4*fh][col fi] of a 32x32 tile
    if constexpr (TN % 2 == 0) {
        if (p.geglu) {          // W rows packed [32 value | 32 gate] per 32 output features
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; b += 2) {
                    const int nv = n0 + wc * WN + b * 32 + fi, ng = nv + 32;
                    const int mbase = m0 + wr * WM + a * 32 + 4 * fh;
                    if (ng >= p.N) continue;
                    const float bv = p.bias ? p.bias[nv] : 0.f, bg = p.bias ? p.bias[ng] : 0.f;
                    const float sv = p.ln_mode ? p.rowvec[nv] : 0.f, sg = p.ln_mode ? p.rowvec[ng] : 0.f;
                    const int nf = ((n0 + wc * WN + b * 32) >> 1) + fi;
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
                            const int bb = m / p.rpb;
                            const unsigned row = (unsigned)bb * (unsigned)p.out_bs + (unsigned)(m - bb * p.rpb);
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
            if (p.ln_mode) {
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
}

template <int BM, int BN, int WR, int WC, int DEPTH>
static void x6_launch(const CGParams& p, bool plain, int sched, int terms, hipStream_t s) {
    dim3 grid(aed_cdiv(p.N, BN), aed_cdiv(p.M, BM), p.ksplit);
    dim3 block(64 * WR * WC);
    if (plain) {
        if (terms == 3) hipLaunchKernelGGL((conv_gemm_x6_kernel<BM, BN, WR, WC, DEPTH, true, 0, 3>), grid, block, 0, s, p);
        else if (sched) hipLaunchKernelGGL((conv_gemm_x6_kernel<BM, BN, WR, WC, DEPTH, true, 1, 6>), grid, block, 0, s, p);
        else hipLaunchKernelGGL((conv_gemm_x6_kernel<BM, BN, WR, WC, DEPTH, true, 0, 6>), grid, block, 0, s, p);
    } else {
        hipLaunchKernelGGL((conv_gemm_x6_kernel<BM, BN, WR, WC, DEPTH, false, 0, 6>), grid, block, 0, s, p);
    }
}

// Tile codes (i[29]): 1 = 128x128, 2 = 128x64, 3 = 64x128, 4 = 64x64 (256 threads); 8 = 256x128, 9 = 128x256 (512 threads:
// one workgroup per CU, two waves per SIMD).  0 = pick.  Shapes this kernel does not take (channel counts that are not a
// multiple of 16, unaligned operands, the latency-regime tiles >= 10, per-batch weights / grouped softmax) run the fp32 path.
// Flag bit 3 (8): ask the compiler for the MFMA / VALU / DS-write interleave (A/B switch of the experiment).
// Flag bit 4 (16): DIAGNOSTIC three-term arithmetic (plain epilogue shapes only).
int launch_conv_gemm_x6(const aed_op* op, hipStream_t s) {
    const int32_t* i = op->i;
    const int Cin = i[11];
    int cfg = i[29];
    const bool fits = (Cin % X6_BK == 0) && (i[3] % 4 == 0) && ((uintptr_t)op->p[0] % 16 == 0) &&
                      ((uintptr_t)op->p[1] % 16 == 0) && cfg < 10 && cfg != 5 && cfg != 6 && i[36] == 0 && i[37] == 0 &&
                      i[39] == 0;
    if (!fits) return launch_conv_gemm(op, s);
    CGParams p;
    int rc = cg_fill_params(op, p, X6_BK);
    if (rc) return rc;
    if (p.ksplit > (p.K + 31) / 32) p.ksplit = (p.K + 31) / 32;     // launch_splitk_reduce clamps with 32-wide chunks
    if (cfg == 0) {
        const int cus = aed_num_cus();
        auto blocks = [&](int bm, int bn) { return (long)aed_cdiv(p.M, bm) * aed_cdiv(p.N, bn) * p.ksplit; };
        if (blocks(256, 128) >= 2L * cus && p.N >= 128) cfg = 8;
        else if (blocks(128, 128) >= (long)cus && p.N >= 128) cfg = 1;
        else if (blocks(128, 64) >= 2L * cus) cfg = 2;
        else cfg = 4;
    }
    const bool plain = p.in_act == 0 && p.ln_mode == 0;
    const int sched = (op->flags & 8) ? 1 : 0;
    const int terms = (op->flags & 16) ? 3 : 6;
    if (p.geglu) AED_REQUIRE(cfg == 1 || cfg == 3 || cfg == 8 || cfg == 9, "conv_gemm_x6: the GEGLU epilogue needs 64-wide wave tiles (cfg %d)", cfg);
    switch (cfg) {
        case 1: x6_launch<128, 128, 2, 2, 2>(p, plain, sched, terms, s); break;
        case 2: x6_launch<128, 64, 2, 2, 2>(p, plain, sched, terms, s); break;
        case 3: x6_launch<64, 128, 2, 2, 2>(p, plain, sched, terms, s); break;
        case 4: x6_launch<64, 64, 2, 2, 2>(p, plain, sched, terms, s); break;
        case 8: x6_launch<256, 128, 4, 2, 2>(p, plain, sched, terms, s); break;
        case 9: x6_launch<128, 256, 2, 4, 2>(p, plain, sched, terms, s); break;
        default: AED_REQUIRE(false, "conv_gemm_x6: bad tile cfg %d", cfg);
    }
    AED_CHECK_HIP(hipGetLastError());
    if (p.ksplit > 1) return launch_splitk_reduce(op, s);
    return 0;
}
