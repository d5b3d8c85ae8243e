// conv_gemm_f8.hip -- the implicit-GEMM convolution / linear layer of conv_gemm.hip on the MX-FP8 matrix cores of gfx950.
//
// EXPERIMENT (round 4; BASELINE config 5 names an "fp8 MFMA path" for the Stable Audio DiT): selected per op by flag bit 6 of an
// AED_OP_CONV_GEMM record, which only tapes built under tape.arith_mode("fp8") set.  NOT a parity path: an e4m3 element carries
// 3 significand bits, a GEMM output deviates from the fp32 result by a few 1e-2 relative (tools/fp8_tolerance_study.py,
// DESIGN.md section 8); what is tested is (a) the kernel against a CPU emulation of the SAME quantisation (agreement to fp32
// accumulation order) and (b) an acceptance bound on the edit it produces against the fp32-exact edit.
//
// Format: OCP MX (microscaling) FP8.  Along K every run of 32 elements of a row shares one power-of-two scale (e8m0 byte):
//   scale = 2^(floor(log2(amax)) - 8),  element = RNE_e4m3(clamp(x / scale, +-448))          (8 = emax of e4m3, 448 = its max)
// v_mfma_scale_f32_32x32x64_f8f6f4 consumes exactly that; products are exact in fp32, the block scales are applied in hardware,
// accumulation is fp32.  Operand layout (no ISA manual in the image: MEASURED with tools/f8_probe.cpp, profiles/r04_f8_probe.jsonl):
// lane (row, half h) holds k = 16 h .. 16 h + 15 in VGPRs 0-3 and k = 32 + 16 h .. 32 + 16 h + 15 in VGPRs 4-7 -- each 32-k block
// is spread over BOTH lane halves -- and supplies the scale byte (selected by op_sel) of block h of its row.
//
// Data path: operands stay fp32 in HBM (same record, same buffers as the other arithmetics -- no second weight copy).  The
// loader threads quantise while writing the LDS stage: a row's 64-k chunk is fetched by 16 threads (float4 each), the block
// amax is a 3-step xor reduction over 8 adjacent lanes, the scaling is an exact v_ldexp_f32, v_cvt_pk_fp8_f32 rounds to nearest
// even (5.5 VALU per element, the price of the split-bf16 loader).  One v_mfma_scale per 32x32x64 block: 16 passes for the
// work of 24 bf16 MFMAs (6 x 4 k-blocks) of the split kernel -- the kernel is bound by the loader's VALU work, not by the matrix
// pipe (pre-quantised weights would halve that; left for a follow-up).  LDS row = 64 B of fp8 + 16 B pad; scales sit beside
// the rows as one dword per (row, k block).  Two LDS stages, one barrier per 64-k chunk, one further chunk in flight in registers.
// Epilogue = conv_gemm_x6.hip's (the C/D layout of the 32x32 MFMAs does not depend on the input format).
#include "cg_params.h"

typedef int f8x32 __attribute__((ext_vector_type(8)));          // 32 fp8 = one lane's A / B operand of a 32x32x64 MFMA
typedef unsigned f8u32x4 __attribute__((ext_vector_type(4)));

int launch_conv_gemm_x6(const aed_op* op, hipStream_t s);
int launch_conv_gemm_f8(const aed_op* op, hipStream_t s);

constexpr int F8_BK = 64;           // fp32 k per chunk = one scaled MFMA
constexpr int F8_ROWQ = 5;          // uint4 per LDS row: 64 B data + 16 B pad
constexpr unsigned F8_OOB = 0x80000000u;

// four consecutive k of one row -> 4 e4m3 bytes; `sbyte` = the e8m0 scale of the row's 32-k block (8 adjacent lanes agree)
__device__ __forceinline__ unsigned f8_quant4(float4 v, int& sbyte) {
    float am = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
    am = fmaxf(am, __shfl_xor(am, 1, 64));
    am = fmaxf(am, __shfl_xor(am, 2, 64));
    am = fmaxf(am, __shfl_xor(am, 4, 64));
    int e = (int)((__float_as_uint(am) >> 23) & 0xffu) - 127 - 8;       // floor(log2(amax)) - emax(e4m3)
    e = max(e, -127);                                                    // all-zero / denormal block: smallest scale
    sbyte = e + 127;
    const float x0 = __builtin_amdgcn_fmed3f(__builtin_amdgcn_ldexpf(v.x, -e), -448.f, 448.f);
    const float x1 = __builtin_amdgcn_fmed3f(__builtin_amdgcn_ldexpf(v.y, -e), -448.f, 448.f);
    const float x2 = __builtin_amdgcn_fmed3f(__builtin_amdgcn_ldexpf(v.z, -e), -448.f, 448.f);
    const float x3 = __builtin_amdgcn_fmed3f(__builtin_amdgcn_ldexpf(v.w, -e), -448.f, 448.f);
    int w = 0;
    w = __builtin_amdgcn_cvt_pk_fp8_f32(x0, x1, w, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(x2, x3, w, true);
    return (unsigned)w;
}

// WQ: the weights arrive pre-quantised (p.Wq / p.Wsc, aed_mx_quantize_rows): the W side of the loader is a plain 16-byte copy per
// thread and chunk -- no VALU, a quarter of the HBM bytes -- and only the activations are quantised in flight
template <int BM, int BN, int WROWS, int WCOLS, bool PLAIN, bool WQ>
__global__ __launch_bounds__(64 * WROWS * WCOLS, 2) void conv_gemm_f8_kernel(CGParams p) {
    constexpr int NT = 64 * WROWS * WCOLS;
    constexpr int WM = BM / WROWS, WN = BN / WCOLS;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int TPR = F8_BK / 4;          // loader threads per row (float4 each) = 16
    constexpr int RPP = NT / TPR;           // rows per loader pass
    constexpr int PA = BM / RPP, PB = WQ ? 1 : BN / RPP;
    constexpr int RPPQ = NT / 4;            // pre-quantised W: 4 threads per 64-byte row chunk, rows per pass
    constexpr int PBQ = WQ ? BN / RPPQ : 1;
    constexpr int STAGE = (BM + BN) * F8_ROWQ;      // uint4 per operand stage (A rows then W rows)
    static_assert(TM >= 1 && TN >= 1 && PA >= 1 && (WQ || BN / RPP >= 1) && (!WQ || BN / RPPQ >= 1), "tile");
    static_assert(BM % RPP == 0 && (WQ ? BN % RPPQ == 0 : BN % RPP == 0), "loader passes");

    __shared__ uint4 lds[2 * STAGE];
    __shared__ unsigned lsc[2][(BM + BN) * 2];      // e8m0 scale (low byte) per (row, k block)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave / WCOLS, wc = wave % WCOLS;
    // XCD-aware tile order (conv_gemm.hip): XCD x gets a contiguous range of tile ids, n fastest
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
    const int lq = tid % TPR;               // which float4 of the 64-k row
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
    // pre-quantised W: thread (row tid / 4, quarter tid % 4) copies 16 bytes of its row's 64-byte chunk
    const int qrow = tid >> 2, qq = tid & 3;
    unsigned qbase[PBQ], qsbase[PBQ];
    bool qvalid[PBQ];
#pragma unroll
    for (int q = 0; q < PBQ; ++q) {
        const int n = n0 + qrow + RPPQ * q;
        qvalid[q] = n < p.N;
        qbase[q] = (unsigned)(qvalid[q] ? n : 0) * (unsigned)p.K + 16u * qq;                 // byte offset into Wq
        qsbase[q] = (unsigned)(qvalid[q] ? n : 0) * (unsigned)(p.K >> 5);                    // byte offset into Wsc
    }
    const int vIH = p.vIH, vIW = p.vIW;
    // running (tap, channel) position of the next chunk to prefetch: [group of p.kgroup channels][tap][chunk within the group]
    int pf_cg, pf_sub, pf_ty, pf_tx;
    const int gq = p.kgroup / F8_BK;
    {
        const int per_group = p.KH * p.KW * gq;
        const int g = kc_begin / per_group;
        const int rem = kc_begin - g * per_group;
        const int tap = rem / gq;
        pf_cg = g * p.kgroup;
        pf_sub = rem - tap * gq;
        pf_ty = tap / p.KW;
        pf_tx = tap - pf_ty * p.KW;
    }

    float4 rbuf_a[PA], rbuf_b[PB];
    uint4 rbuf_q[PBQ];
    unsigned rbuf_s[PBQ];
    float ln_s1[PA], ln_s2[PA];
#pragma unroll
    for (int q = 0; q < PA; ++q) { ln_s1[q] = 0.f; ln_s2[q] = 0.f; }

    const __amdgpu_buffer_rsrc_t srd_a = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, 0x7ffffff0, 0x00020000);
    const __amdgpu_buffer_rsrc_t srd_a2 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A2 ? p.A2 : p.A), 0, 0x7ffffff0, 0x00020000);
    const __amdgpu_buffer_rsrc_t srd_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, 0x7ffffff0, 0x00020000);
    const __amdgpu_buffer_rsrc_t srd_q = __builtin_amdgcn_make_buffer_rsrc((void*)(WQ ? p.Wq : (const unsigned char*)p.W), 0, 0x7ffffff0, 0x00020000);
    const __amdgpu_buffer_rsrc_t srd_s = __builtin_amdgcn_make_buffer_rsrc((void*)(WQ ? p.Wsc : (const unsigned char*)p.W), 0, 0x7ffffff0, 0x00020000);
    auto as_f4 = [](f8u32x4 v) {
        return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
    };
    auto prefetch = [&](int kc, float4 (&ra)[PA], float4 (&rb)[PB]) {
        int c0 = pf_cg + pf_sub * F8_BK;
        const int dy = pf_ty * p.dil_h, dx = pf_tx * p.dil_w;
        const unsigned k0 = (unsigned)((pf_ty * p.KW + pf_tx) * p.Cin + c0);     // W column of this chunk
        pf_sub += 1;
        const bool wrap = pf_sub == gq;
        pf_sub = wrap ? 0 : pf_sub;
        pf_tx += wrap ? 1 : 0;
        const bool wrap2 = pf_tx == p.KW;
        pf_tx = wrap2 ? 0 : pf_tx;
        pf_ty += wrap2 ? 1 : 0;
        const bool wrap3 = pf_ty == p.KH;
        pf_ty = wrap3 ? 0 : pf_ty;
        pf_cg += wrap3 ? p.kgroup : 0;
        const bool second = p.C1 > 0 && c0 >= p.C1;         // two-source A: block-uniform select per chunk
        c0 -= second ? p.C1 : 0;
        const unsigned ld = second ? (unsigned)p.lda2 : (unsigned)p.lda;
        const bool dead = kc >= kc_end;
#pragma unroll
        for (int q = 0; q < PA; ++q) {
            const int iy = ay0[q] + dy, ix = ax0[q] + dx;
            const bool ok = avalid[q] & ((unsigned)iy < (unsigned)vIH) & ((unsigned)ix < (unsigned)vIW) & !dead;
            const unsigned base = second ? abase2[q] : abase[q];
            const unsigned off = (base + (unsigned)((iy >> p.up) * p.IW + (ix >> p.up)) * ld + (unsigned)c0) * 4u;
            ra[q] = as_f4(__builtin_amdgcn_raw_buffer_load_b128(second ? srd_a2 : srd_a, ok ? off : F8_OOB, 0, 0));
        }
        if constexpr (WQ) {
#pragma unroll
            for (int q = 0; q < PBQ; ++q) {
                const bool ok = qvalid[q] & !dead;
                const f8u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(srd_q, ok ? qbase[q] + k0 : F8_OOB, 0, 0);
                rbuf_q[q] = make_uint4(v.x, v.y, v.z, v.w);
                rbuf_s[q] = (unsigned)__builtin_amdgcn_raw_buffer_load_b16(srd_s, ok ? qsbase[q] + (k0 >> 5) : F8_OOB, 0, 0);
            }
        } else {
#pragma unroll
            for (int q = 0; q < PB; ++q)
                rb[q] = as_f4(__builtin_amdgcn_raw_buffer_load_b128(
                    srd_w, (wvalid[q] & !dead) ? (wbase[q] + k0) * 4u : F8_OOB, 0, 0));
        }
    };

    // one fetched float4 -> 4 fp8 bytes of its LDS row (+ the block's scale, written by the block's first lane)
    auto put_row = [&](uint4* st, unsigned* sc, int row, float4 v) {
        int sb;
        const unsigned w = f8_quant4(v, sb);
        reinterpret_cast<unsigned*>(st + row * F8_ROWQ)[lq] = w;
        if ((lq & 7) == 0) sc[row * 2 + (lq >> 3)] = (unsigned)sb;
    };
    auto stage_write = [&](int s, const float4 (&ra)[PA], const float4 (&rb)[PB]) {
        uint4* st = lds + s * STAGE;
        unsigned* sc = lsc[s];
#pragma unroll
        for (int q = 0; q < PA; ++q) {
            float4 v = ra[q];
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
            put_row(st, sc, lrow + RPP * q, v);
        }
        if constexpr (WQ) {
#pragma unroll
            for (int q = 0; q < PBQ; ++q) {
                const int row = BM + qrow + RPPQ * q;
                st[row * F8_ROWQ + qq] = rbuf_q[q];
                if (qq == 0) {
                    sc[row * 2] = rbuf_s[q] & 0xffu;
                    sc[row * 2 + 1] = (rbuf_s[q] >> 8) & 0xffu;
                }
            }
        } else {
#pragma unroll
            for (int q = 0; q < PB; ++q) put_row(st, sc, BM + lrow + RPP * q, rb[q]);
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int fi = lane & 31;        // fragment row (A: m, W: n)
    const int fh = lane >> 5;        // which 16 of each 32-k block (operand layout above); also: whose block scale this lane supplies
    const int a_row = wr * WM + fi;
    const int b_row = BM + wc * WN + fi;

    auto compute = [&](int s) {
        const uint4* st = lds + s * STAGE;
        const unsigned* sc = lsc[s];
        f8x32 af[TM], bw[TN];
        int sa[TM], sw[TN];
#pragma unroll
        for (int a = 0; a < TM; ++a) {
            const uint4 lo = st[(a_row + a * 32) * F8_ROWQ + fh], hi = st[(a_row + a * 32) * F8_ROWQ + 2 + fh];
            af[a] = (f8x32){(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
            sa[a] = (int)sc[(a_row + a * 32) * 2 + fh];
        }
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const uint4 lo = st[(b_row + b * 32) * F8_ROWQ + fh], hi = st[(b_row + b * 32) * F8_ROWQ + 2 + fh];
            bw[b] = (f8x32){(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
            sw[b] = (int)sc[(b_row + b * 32) * 2 + fh];
        }
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
                acc[a][b] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(af[a], bw[b], acc[a][b], 0, 0, 0, sa[a], 0, sw[b]);
    };

    prefetch(kc_begin, rbuf_a, rbuf_b);
    stage_write(0, rbuf_a, rbuf_b);
    prefetch(kc_begin + 1, rbuf_a, rbuf_b);
    __syncthreads();
    int cur = 0;
    // per iteration: quantise + write chunk kc+1 into the other stage, fetch chunk kc+2, MFMAs of chunk kc, barrier
    for (int kc = kc_begin; kc < kc_end; ++kc) {
        stage_write(cur ^ 1, rbuf_a, rbuf_b);
        prefetch(kc + 2, rbuf_a, rbuf_b);
        compute(cur);
        __syncthreads();
        cur ^= 1;
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

    // ---- epilogue (conv_gemm_x6.hip's): acc[a][b][r] = C[row (r&3) + 8*(r>>2) + 4*fh][col fi] of a 32x32 tile
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

template <int BM, int BN, int WR, int WC>
static int f8_launch(const CGParams& p, bool plain, hipStream_t s) {
    dim3 grid(aed_cdiv(p.N, BN), aed_cdiv(p.M, BM), p.ksplit);
    dim3 block(64 * WR * WC);
    if (p.Wq) {
        if (plain) hipLaunchKernelGGL((conv_gemm_f8_kernel<BM, BN, WR, WC, true, true>), grid, block, 0, s, p);
        else hipLaunchKernelGGL((conv_gemm_f8_kernel<BM, BN, WR, WC, false, true>), grid, block, 0, s, p);
    } else {
        if (plain) hipLaunchKernelGGL((conv_gemm_f8_kernel<BM, BN, WR, WC, true, false>), grid, block, 0, s, p);
        else hipLaunchKernelGGL((conv_gemm_f8_kernel<BM, BN, WR, WC, false, false>), grid, block, 0, s, p);
    }
    return 0;
}

// ---- weights -> MX-FP8 once, at engine build (aed_mx_quantize_rows): the same f8_quant4 the loader applies in flight
__global__ __launch_bounds__(256) void mx_quantize_rows_kernel(const float* __restrict__ src, unsigned* __restrict__ q,
                                                               unsigned char* __restrict__ sc, long long total4, int k4) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long ld = idx < total4 ? idx : total4 - 1;          // whole 8-lane groups are valid or not (total4 % 8 == 0)
    const float4 v = reinterpret_cast<const float4*>(src)[ld];
    int sb;
    const unsigned w = f8_quant4(v, sb);
    if (idx < total4) {
        q[idx] = w;
        const long long row = idx / k4;
        const int c4 = (int)(idx - row * k4);
        if ((c4 & 7) == 0) sc[row * (k4 >> 3) + (c4 >> 3)] = (unsigned char)sb;
    }
}

extern "C" int aed_mx_quantize_rows(const float* src, void* q, void* scales, long long rows, int K, void* stream) {
    AED_REQUIRE(src && q && scales && rows > 0 && K > 0 && K % 32 == 0, "aed_mx_quantize_rows: K=%d must be a positive multiple of 32", K);
    AED_REQUIRE((uintptr_t)src % 16 == 0 && (uintptr_t)q % 4 == 0, "aed_mx_quantize_rows: unaligned operand");
    const long long total4 = rows * (K / 4);
    const long long blocks = (total4 + 255) / 256;
    AED_REQUIRE(blocks < (1LL << 31), "aed_mx_quantize_rows: too many elements for one launch");
    hipLaunchKernelGGL(mx_quantize_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, (unsigned*)q,
                       (unsigned char*)scales, total4, K / 4);
    AED_CHECK_HIP(hipGetLastError());
    return 0;
}

// Tile codes (i[29]): 1 = 128x128, 2 = 128x64, 3 = 64x128, 4 = 64x64 (256 threads); the split-bf16 kernel's 512-thread codes
// 8 / 9 map to 128x128; 0 = pick.  Shapes this kernel does not take (channel counts that are not a multiple of 64, the skinny
// and latency-regime tiles, per-batch weights / grouped softmax, unaligned or > 2 GB operands) run the split-bf16 path.
int launch_conv_gemm_f8(const aed_op* op, hipStream_t s) {
    const int32_t* i = op->i;
    const int Cin = i[11];
    int cfg = i[29];
    if (cfg == 8 || cfg == 9) cfg = 1;
    const long long batch = i[0] / (i[9] * i[10] > 0 ? i[9] * i[10] : 1);
    const bool fits = (Cin % F8_BK == 0) && (i[3] % 4 == 0) && ((uintptr_t)op->p[0] % 16 == 0) &&
                      ((uintptr_t)op->p[1] % 16 == 0) && cfg < 5 && i[36] == 0 && i[37] == 0 && i[38] <= 1 && i[39] == 0 &&
                      (i[32] == 0 || i[32] % F8_BK == 0) &&
                      batch * i[20] + (long long)i[7] * i[8] * i[3] < (1LL << 29) && (long long)i[1] * i[2] < (1LL << 29) &&
                      (i[32] == 0 || batch * i[34] + (long long)i[7] * i[8] * i[33] < (1LL << 29));
    if (!fits) return launch_conv_gemm_x6(op, s);
    CGParams p;
    int rc = cg_fill_params(op, p, F8_BK);
    if (rc) return rc;
    if (op->flags & 128) {          // pre-quantised weights: p[7] = e4m3 bytes [N][K], p[9] = e8m0 scales [N][K/32]
        p.Wq = (const unsigned char*)op->p[7];
        p.Wsc = (const unsigned char*)op->p[9];
        AED_REQUIRE(p.Wq && p.Wsc && (uintptr_t)p.Wq % 16 == 0 && (long long)p.N * p.K < (1LL << 31),
                    "conv_gemm_f8: flag bit 7 needs 16-byte aligned pre-quantised weights in p[7] / p[9]");
    }
    if (p.ksplit > p.nchunks) p.ksplit = p.nchunks;                 // this kernel walks F8_BK-wide chunks
    if (p.ksplit < 1) p.ksplit = 1;
    if (cfg == 0) {
        const int cus = aed_num_cus();
        auto blocks = [&](int bm, int bn) { return (long)aed_cdiv(p.M, bm) * aed_cdiv(p.N, bn) * p.ksplit; };
        if (blocks(128, 128) >= (long)cus && p.N >= 128) cfg = 1;
        else if (blocks(128, 64) >= 2L * cus) cfg = 2;
        else cfg = 4;
    }
    const bool plain = p.in_act == 0 && p.ln_mode == 0;
    if (p.geglu) AED_REQUIRE(cfg == 1 || cfg == 3, "conv_gemm_f8: the GEGLU epilogue needs 64-wide wave tiles (cfg %d)", cfg);
    switch (cfg) {
        case 1: rc = f8_launch<128, 128, 2, 2>(p, plain, s); break;
        case 2: rc = f8_launch<128, 64, 2, 2>(p, plain, s); break;
        case 3: rc = f8_launch<64, 128, 2, 2>(p, plain, s); break;
        case 4: rc = f8_launch<64, 64, 2, 2>(p, plain, s); break;
        default: AED_REQUIRE(false, "conv_gemm_f8: bad tile cfg %d", cfg);
    }
    if (rc) return rc;
    AED_CHECK_HIP(hipGetLastError());
    if (p.ksplit > 1) return launch_splitk_reduce_slices(op, p.ksplit, s);
    return 0;
}
