// lin_gemm.hip -- latency-regime GEMM / small convolution on the fp32 matrix cores of gfx950.
//
// The reverse (edit) loop of the reference runs the U-Net at batch 2 (inversion_utils.py:221-315: one uncond and one
// cond sample per step), one hundred dependent times per clip.  At that batch most of the ~450 contractions of a
// forward are a few hundred 32x32 output tiles with K of a few hundred: microseconds of MFMA work whose duration is
// set by how fast the operands arrive, not by the matrix pipe.  Round 1's wave-split-K kernel fed the MFMAs with
// lane=row loads (64 cache-line requests per instruction, 16 useful bytes per line per request) and spent ~10k of
// its ~22k cycles in the texture address path.  This kernel keeps its decomposition and fixes the feed:
//
//   * a workgroup of NW wavefronts owns ONE (32*TM) x (32*TN) output tile; the wavefronts split K;
//   * every wavefront stages ITS K-slice of the A and W tiles through a wave-private LDS slab with fully coalesced
//     16-byte loads (8 lanes cover 128 contiguous bytes of one row, 8 rows per instruction), two 32-wide K chunks in
//     flight in registers; no workgroup barrier inside the main loop (LDS operations of one wave execute in order);
//   * MFMA fragments come from the padded slab (conflict-free ds_read_b128), v_mfma_f32_32x32x2_f32 chains;
//   * the NW partial tiles meet once in LDS, are summed in a fixed order (deterministic) and every thread finishes
//     its outputs with coalesced stores: bias, per-batch time-embedding row, residual, activation, the folded
//     LayerNorm, and the fused GEGLU gate (value and gate columns are packed side by side, TN = 2).
//   * A(m,k) supports the same implicit-GEMM gather as conv_gemm.hip (taps, stride, padding, dilation, nearest
//     upsample) and a two-source channel split, so up-block concatenations are never materialised.
//
// Same arithmetic contract as conv_gemm.hip: exact fp32 FMA chains, fixed summation order per (tile config).
#include <mutex>
#include "cg_params.h"

// ordering of a wave's own LDS traffic: the hardware executes one wave's DS operations in issue order; this only
// pins the compiler (a memory clobber on an empty asm, no instruction)
#define LDS_ORDER() asm volatile("" ::: "memory")

// MODE 0: plain Linear / 1x1 convolution with uniformly strided rows (A row m at m*lda): no tap arithmetic, no masks,
//         no loader transform -- the lean path most launches of the edit loop take;
// MODE 1: MODE 0 + fused LayerNorm row statistics;
// MODE 2: general implicit-GEMM gather (taps, stride, padding, dilation, upsample, per-batch strides) and/or a loader
//         activation (SiLU / LeakyReLU).
// Everything that differs is compile-time: the run-time-flag version of this kernel spent more issue slots in
// scalar branches than in MFMAs (and the compiler sank the masked loads into conditionals, serialising them).
template <int NW, int TM, int TN, int DEPTH, int MODE>
__global__ __launch_bounds__(64 * NW) void lin_gemm_kernel(CGParams p) {
    constexpr bool GATHER = MODE == 2;
    constexpr bool LN = MODE == 1;
    constexpr int CH = 32;                        // K per chunk
    constexpr int LD = CH + 4;                    // padded LDS row (floats)
    constexpr int AR = 32 * TM, WR = 32 * TN;     // tile rows of A / of W
    constexpr int WAVE_LDS = (AR + WR) * LD;      // floats of one wave's slab (>= TM*TN*1024: the partial tiles fit)
    constexpr int PA = AR / 8, PW = WR / 8;       // float4 loads per lane and chunk (8 rows per wave instruction)
    static_assert(WAVE_LDS >= TM * TN * 1024, "partial tiles must fit the staging slab");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ float ln_part[NW][AR][2];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fi = lane & 31, fh = lane >> 5;
    const int lrow = lane >> 3, lseg = (lane & 7) * 4;
    float* slab = smem + wave * WAVE_LDS;
    float* As = slab;
    float* Ws = slab + AR * LD;

    // XCD-aware tile order (see conv_gemm.hip): XCD x gets a contiguous range of tile ids, n fastest, so the tiles that
    // share an A row panel share one L2.
    int tile_x, tile_y;
    {
        const unsigned nx = gridDim.x, nwg = nx * gridDim.y;
        const unsigned orig = blockIdx.y * nx + blockIdx.x;
        const unsigned xcd = orig & 7u, q = nwg >> 3, r = nwg & 7u;
        const unsigned id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
        tile_y = (int)(id / nx);
        tile_x = (int)(id - (unsigned)tile_y * nx);
    }
    const int m0 = tile_y * AR;
    const int n0 = tile_x * WR;                   // packed W row of the tile (GEGLU: 64 packed rows = 32 features)
    // optional in-kernel timeline (s_memtime, shader cycles): slots 0..15 = first workgroup, 16..31 = last workgroup
    const bool last_wg = blockIdx.x == gridDim.x - 1 && blockIdx.y == gridDim.y - 1;
    const bool dbg_on = p.dbg != nullptr && tid == 0 && ((blockIdx.x == 0 && blockIdx.y == 0) || last_wg);
    int dbg_n = last_wg ? 16 : 0;
    const int dbg_end = dbg_n + 15;
#define STAMP() do { if (dbg_on && dbg_n < dbg_end) p.dbg[dbg_n++] = __builtin_amdgcn_s_memtime(); } while (0)
    STAMP();

    // this wave's range of 32-wide K chunks
    const int nch = p.K / CH;
    const int cbase = nch / NW, crem = nch % NW;
    const int c_begin = wave * cbase + min(wave, crem);
    const int c_end = c_begin + cbase + (wave < crem ? 1 : 0);

    // loader rows of this lane
    int ay0[PA], ax0[PA], ab[PA];
    unsigned aoff[PA], aoff2[PA];    // MODE 0/1: element offset of the row in A / in the second source
#pragma unroll
    for (int j = 0; j < PA; ++j) {
        const int m = m0 + lrow + 8 * j;
        const int mm = m < p.M ? m : p.M - 1;      // rows past M compute garbage that is never stored
        if constexpr (GATHER) {
            const int b = mm / p.rpb;
            const int r = mm - b * p.rpb;
            const int oy = r / p.OW, ox = r - oy * p.OW;
            ay0[j] = oy * p.stride - p.pad_h;
            ax0[j] = ox * p.stride - p.pad_w;
            ab[j] = b;
        } else {
            aoff[j] = (unsigned)mm * (unsigned)p.lda + lseg;
            aoff2[j] = (unsigned)mm * (unsigned)p.lda2 + lseg;
        }
    }
    // batch item of this tile's rows (per-batch weights / vectors; a tile never straddles batch items there)
    const int bt = (p.w_bs | p.vec_bs) ? min(m0, p.M - 1) / p.rpb : 0;
    unsigned woff[PW];
#pragma unroll
    for (int j = 0; j < PW; ++j) {
        const int n = min(n0 + lrow + 8 * j, p.N - 1);      // rows past N compute garbage columns that are never stored
        woff[j] = (unsigned)bt * (unsigned)p.w_bs + (unsigned)n * (unsigned)p.K + lseg;
    }
    const int vIH = p.vIH, vIW = p.vIW;
    // running (tap, channel) position of the next chunk to prefetch
    int pf_c0 = c_begin * CH, pf_ty = 0, pf_tx = 0;
    if constexpr (GATHER) {
        const int k0 = c_begin * CH;
        const int tap = k0 / p.Cin;
        pf_c0 = k0 - tap * p.Cin;
        pf_ty = tap / p.KW;
        pf_tx = tap - pf_ty * p.KW;
    }

    // epilogue operands of this thread's outputs are fetched NOW (column t & 31 of rows (t >> 5) + 2*NW*i of every 32x32
    // sub-tile): after the last barrier the finish is LDS reads, arithmetic and stores -- no dependent global load
    const int col = tid & 31;
    constexpr int RSTEP = 2 * NW;
    constexpr int RI = (32 + RSTEP - 1) / RSTEP;
    // simple rows (every Linear of the U-Net): output row = m, no per-batch row vector, no accumulate modes
    const bool simple = p.o_mul == 1 && p.o_add == 0 && p.out_bs == p.rpb && p.o_len == p.rpb && p.accumulate == 0 &&
                        (p.rowvec == nullptr || p.ln_mode);
    float e_bias[TN], e_sum[TN], e_res[TM][RI][TN];
#pragma unroll
    for (int b = 0; b < TN; ++b) {
        const int no = min(n0 + b * 32 + col, p.N - 1);
        const size_t vo = (size_t)bt * p.vec_bs + (size_t)no * p.vec_ld;
        e_bias[b] = p.bias ? p.bias[vo] : 0.f;
        e_sum[b] = LN ? p.rowvec[vo] : 0.f;
    }
    if (simple && p.res != nullptr && !p.late_epilogue) {
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int i = 0; i < RI; ++i)
#pragma unroll
                for (int b = 0; b < TN; ++b) {
                    const int mo = min(m0 + a * 32 + min((tid >> 5) + RSTEP * i, 31), p.M - 1);
                    const int no = min(n0 + b * 32 + col, p.N - 1);
                    e_res[a][i][b] = p.res[(size_t)mo * p.ldr + no];
                }
    } else {
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int i = 0; i < RI; ++i)
#pragma unroll
                for (int b = 0; b < TN; ++b) e_res[a][i][b] = 0.f;
    }

    float4 ra[DEPTH][PA], rw[DEPTH][PW];
    int rmask[DEPTH][GATHER ? PA : 1];   // GATHER: per staged row, all ones inside the map / 0 in the zero padding

    auto prefetch = [&](int kc, float4 (&a)[PA], float4 (&w)[PW], int (&mask)[GATHER ? PA : 1]) {
        const int k0 = min(kc, nch - 1) * CH;               // dead prefetches past the end stay in bounds
        if constexpr (GATHER) {
            const int dy = pf_ty * p.dil_h, dx = pf_tx * p.dil_w;
            int c0 = pf_c0;
            pf_c0 += CH;
            if (pf_c0 >= p.Cin) {
                pf_c0 = 0;
                if (++pf_tx == p.KW) { pf_tx = 0; ++pf_ty; }
            }
            // wave-uniform source select (C1 is a multiple of the chunk width)
            const float* src = p.A;
            int ld = p.lda, bs = p.a_bs;
            if (p.C1 > 0 && c0 >= p.C1) { src = p.A2; ld = p.lda2; bs = p.a_bs2; c0 -= p.C1; }
#pragma unroll
            for (int j = 0; j < PA; ++j) {
                const int iy = ay0[j] + dy, ix = ax0[j] + dx;
                // clamped, always-valid address; the zero padding is applied at the LDS write, so the load is unconditional
                // and stays in flight (a select lets the compiler sink the load into a branch)
                const int cy = min(max(iy, 0), vIH - 1) >> p.up, cx = min(max(ix, 0), vIW - 1) >> p.up;
                const unsigned off = (unsigned)ab[j] * (unsigned)bs + (unsigned)(cy * p.IW + cx) * (unsigned)ld + c0 + lseg;
                a[j] = *reinterpret_cast<const float4*>(src + off);
#ifdef LIN_GATHER_R5_FORM
                mask[j] = (((unsigned)iy < (unsigned)vIH) & ((unsigned)ix < (unsigned)vIW)) ? 0x3f800000 : 0;
#else
                // keep bits: the sign of iy | (vIH-1-iy) | ix | (vIW-1-ix) is clear exactly when 0 <= iy < vIH and
                // 0 <= ix < vIW.  (The shift is opaque to the optimiser, which would turn `x >> 31` back into v_cmp + select.)
                int outside = iy | (vIH - 1 - iy) | ix | (vIW - 1 - ix);
                asm volatile("v_ashrrev_i32 %0, 31, %0" : "+v"(outside));       // 0 inside, -1 in the padding
                mask[j] = ~outside;
#endif
            }
        } else {
            int c0 = k0;
            if (p.C1 > 0 && c0 >= p.C1) {                   // wave-uniform
                c0 -= p.C1;
#pragma unroll
                for (int j = 0; j < PA; ++j) a[j] = *reinterpret_cast<const float4*>(p.A2 + aoff2[j] + c0);
            } else {
#pragma unroll
                for (int j = 0; j < PA; ++j) a[j] = *reinterpret_cast<const float4*>(p.A + aoff[j] + c0);
            }
        }
#pragma unroll
        for (int j = 0; j < PW; ++j) w[j] = *reinterpret_cast<const float4*>(p.W + woff[j] + k0);
    };

    auto stage_write = [&](const float4 (&a)[PA], const float4 (&w)[PW], const int (&mask)[GATHER ? PA : 1]) {
#pragma unroll
        for (int j = 0; j < PA; ++j) {
            float4 v = a[j];
            if constexpr (GATHER) {
#ifdef LIN_GATHER_R5_FORM
                {   // rounds 1-5: data * (1.0 | 0.0), scheduled by the compiler as v_pk_mul_f32 ... op_sel_hi:[1,0].  Reproducibly
                    // wrong in lanes 48-63 of single loader rows when split-bf16 GEMM workgroups of another queue share the CU
                    // (tools/diag/lin_gather_stress.py builds this form to show that the harness still bites)
                    const float m = __int_as_float(mask[j]);
                    v.x *= m; v.y *= m; v.z *= m; v.w *= m;
                }
#else
                // Zero padding by bitwise AND with the keep bits, FENCED: in rounds 1-5 this was a multiply by a 1.0 / 0.0 float
                // that the compiler scheduled freely (v_pk_mul_f32 with a broadcast mask); on the MI355X that form produced wrong
                // values in lanes 48-63 of single loader rows whenever split-bf16 GEMM workgroups of ANOTHER queue were
                // co-resident on the CU -- the first perturbed node of round 5's CFG-row-sharing issue (tiles 11 / 15, the 72 KB
                // configurations that fit beside them; profiles/r06_lin_gather_hazard.md: reproducer, 20 kernel variants, what
                // cured it and what did not).  Every variant with the masking inside one asm volatile block was clean (7 forms x
                // 400 launches x 2 tiles); the AND is also exact whatever the clamped load returned (0.0 * Inf would be NaN).
                asm volatile("v_and_b32 %0, %0, %4\n\tv_and_b32 %1, %1, %4\n\tv_and_b32 %2, %2, %4\n\tv_and_b32 %3, %3, %4"
                             : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w) : "v"(mask[j]));
#endif
                if (p.in_act == AED_ACT_SILU) {             // f(0) = 0 keeps the zero padding
                    v.x = v.x / (1.0f + __expf(-v.x)); v.y = v.y / (1.0f + __expf(-v.y));
                    v.z = v.z / (1.0f + __expf(-v.z)); v.w = v.w / (1.0f + __expf(-v.w));
                } else if (p.in_act == AED_ACT_LEAKY) {
                    v.x = v.x > 0.f ? v.x : v.x * p.in_slope; v.y = v.y > 0.f ? v.y : v.y * p.in_slope;
                    v.z = v.z > 0.f ? v.z : v.z * p.in_slope; v.w = v.w > 0.f ? v.w : v.w * p.in_slope;
                }
            }
            *reinterpret_cast<float4*>(As + (lrow + 8 * j) * LD + lseg) = make_float4(v.x, v.y, v.z, v.w);
        }
#pragma unroll
        for (int j = 0; j < PW; ++j) {
            const float4 v = w[j];
            *reinterpret_cast<float4*>(Ws + (lrow + 8 * j) * LD + lseg) = make_float4(v.x, v.y, v.z, v.w);
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    float ln_s1[TM], ln_s2[TM];
#pragma unroll
    for (int a = 0; a < TM; ++a) { ln_s1[a] = 0.f; ln_s2[a] = 0.f; }

    const float* a_frag = As + fi * LD + 4 * fh;
    const float* b_frag = Ws + fi * LD + 4 * fh;

    // the MFMAs of one staged chunk; the fragments of k-block kb+1 are fetched before the MFMAs of kb issue
    auto mfma_chunk = [&]() {
        float4 af[2][TM], bf[2][TN];
#pragma unroll
        for (int a = 0; a < TM; ++a) af[0][a] = *reinterpret_cast<const float4*>(a_frag + a * 32 * LD);
#pragma unroll
        for (int b = 0; b < TN; ++b) bf[0][b] = *reinterpret_cast<const float4*>(b_frag + b * 32 * LD);
#pragma unroll
        for (int kb = 0; kb < CH / 8; ++kb) {
            const int c = kb & 1, nx = c ^ 1;
            if (kb + 1 < CH / 8) {
#pragma unroll
                for (int a = 0; a < TM; ++a)
                    af[nx][a] = *reinterpret_cast<const float4*>(a_frag + a * 32 * LD + 8 * (kb + 1));
#pragma unroll
                for (int b = 0; b < TN; ++b)
                    bf[nx][b] = *reinterpret_cast<const float4*>(b_frag + b * 32 * LD + 8 * (kb + 1));
            }
            if constexpr (LN) {
#pragma unroll
                for (int a = 0; a < TM; ++a) {
                    const float4 v = af[c][a];
                    ln_s1[a] += (v.x + v.y) + (v.z + v.w);
                    ln_s2[a] += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
                }
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

    // every prefetch is unconditional (clamped addresses): the number of loads in flight is static
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) prefetch(c_begin + d, ra[d], rw[d], rmask[d]);
    STAMP();
    for (int kc0 = c_begin; kc0 < c_end; kc0 += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int kc = kc0 + d;
            if (kc >= c_end) break;
            stage_write(ra[d], rw[d], rmask[d]);
            prefetch(kc + DEPTH, ra[d], rw[d], rmask[d]);
            // the slab is wave-private: LDS operations of one wave complete in issue order; only keep the compiler from
            // moving the fragment reads above the staging writes (and the next writes above these reads)
            LDS_ORDER();
            __builtin_amdgcn_wave_barrier();
            mfma_chunk();
            LDS_ORDER();
            __builtin_amdgcn_wave_barrier();
            STAMP();
        }
    }

    // ---- partial tiles -> this wave's own slab.  acc[a][b][r] = C[row (r&3)+8*(r>>2)+4*fh][col fi] of tile (a,b)
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) slab[((a * TN + b) * 16 + r) * 64 + lane] = acc[a][b][r];
    if constexpr (LN) {
#pragma unroll
        for (int a = 0; a < TM; ++a) {
            const float s1 = ln_s1[a] + __shfl_xor(ln_s1[a], 32, 64);
            const float s2 = ln_s2[a] + __shfl_xor(ln_s2[a], 32, 64);
            if (fh == 0) { ln_part[wave][a * 32 + fi][0] = s1; ln_part[wave][a * 32 + fi][1] = s2; }
        }
    }
    STAMP();
    __syncthreads();
    STAMP();

    // ---- finish: thread t owns column t & 31 of rows (t >> 5) + 2*NW*i of every 32x32 sub-tile
#pragma unroll
    for (int a = 0; a < TM; ++a) {
#pragma unroll
        for (int i = 0; i < RI; ++i) {
            const int row = (tid >> 5) + RSTEP * i;
            if (row >= 32) break;
            const int h = (row >> 2) & 1;
            const int r = (row & 3) + 4 * (row >> 3);
            const int src_lane = col + 32 * h;
            const int mo = m0 + a * 32 + row;
            float mean = 0.f, rstd = 1.f;
            if constexpr (LN) {
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int w = 0; w < NW; ++w) { s1 += ln_part[w][a * 32 + row][0]; s2 += ln_part[w][a * 32 + row][1]; }
                mean = s1 / (float)p.K;
                const float var = fmaxf(s2 / (float)p.K - mean * mean, 0.f);
                rstd = 1.0f / sqrtf(var + p.ln_eps);
            }
            float v[TN];
#pragma unroll
            for (int b = 0; b < TN; ++b) {
                float sacc = 0.f;
#pragma unroll
                for (int w = 0; w < NW; ++w) sacc += smem[w * WAVE_LDS + ((a * TN + b) * 16 + r) * 64 + src_lane];
                v[b] = sacc;
            }
            if (mo >= p.M) continue;
            if (TN % 2 == 0 && p.geglu) {
                // packed columns: sub-tile 2q holds 32 value columns, sub-tile 2q+1 the matching 32 gate columns
#pragma unroll
                for (int b = 0; b + 1 < TN; b += 2) {
                    const int nv = n0 + b * 32 + col, ng = nv + 32;
                    if (ng >= p.N) continue;
                    float val = v[b], gate = v[b + 1];
                    if constexpr (LN) {
                        val = rstd * (val - mean * e_sum[b]);
                        gate = rstd * (gate - mean * e_sum[b + 1]);
                    }
                    val += e_bias[b];
                    gate += e_bias[b + 1];
                    const int b0 = mo / p.rpb;
                    const size_t orow = (size_t)b0 * p.out_bs + (mo - b0 * p.rpb);
                    p.C[orow * p.ldc + (n0 >> 1) + b * 16 + col] = val * glu_gate(gate, p.geglu);
                }
            } else if (p.sm_group > 0) {
                // grouped softmax (cross-attention scores: one group = the keys of one head).  All lanes of a group share
                // the row (col = lane & 31, groups are aligned runs of 8/16/32 lanes), N is a multiple of 32 here.
#pragma unroll
                for (int b = 0; b < TN; ++b) {
                    const int no = n0 + b * 32 + col;
                    float val = v[b];
                    if constexpr (LN) val = rstd * (val - mean * e_sum[b]);
                    val = (val + e_bias[b]) * p.sm_scale;
                    if (p.kbias) val += p.kbias[(size_t)bt * p.sm_group + (no & (p.sm_group - 1))];
                    float mx = val;
                    for (int o = p.sm_group >> 1; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
                    const float e = __expf(val - mx);
                    float sum = e;
                    for (int o = p.sm_group >> 1; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
                    if (no < p.N) p.C[(size_t)mo * p.ldc + no] = e / sum;      // (a 64-wide tile may overhang N = 32)
                }
            } else if (simple) {
#pragma unroll
                for (int b = 0; b < TN; ++b) {
                    const int no = n0 + b * 32 + col;
                    if (no >= p.N) continue;
                    float val = v[b];
                    if constexpr (LN) val = rstd * (val - mean * e_sum[b]);
                    val += e_bias[b] + (p.late_epilogue && p.res ? p.res[(size_t)mo * p.ldr + no] : e_res[a][i][b]);
                    if (p.out_act != AED_ACT_NONE) val = aed_apply_act(val, p.out_act, p.out_p);
                    p.C[(size_t)mo * p.ldc + no] = val;
                }
            } else {
#pragma unroll
                for (int b = 0; b < TN; ++b) {
                    const int no = n0 + b * 32 + col;
                    if (no >= p.N) continue;
                    float val = v[b];
                    if constexpr (LN) val = rstd * (val - mean * e_sum[b]);
                    store_out(p, mo, no, val);
                }
            }
        }
    }
    STAMP();
    if (dbg_on) p.dbg[dbg_end] = dbg_n - (last_wg ? 16 : 0);
#undef STAMP
}

template <int NW, int TM, int TN, int DEPTH, int MODE>
static int launch_lin_mode(const CGParams& p, hipStream_t s) {
    constexpr int WAVE_LDS = (32 * TM + 32 * TN) * 36;
    const size_t bytes = sizeof(float) * NW * WAVE_LDS;
    // once per instantiation, whichever host thread gets here first (the clip pipeline launches from ~5 threads)
    static std::once_flag attr_once;
    static hipError_t attr_rc = hipSuccess;
    std::call_once(attr_once, [&] {
        attr_rc = hipFuncSetAttribute((const void*)lin_gemm_kernel<NW, TM, TN, DEPTH, MODE>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    });
    AED_CHECK_HIP(attr_rc);
    dim3 grid(aed_cdiv(p.N, 32 * TN), aed_cdiv(p.M, 32 * TM), 1);
    hipLaunchKernelGGL((lin_gemm_kernel<NW, TM, TN, DEPTH, MODE>), grid, dim3(64 * NW), bytes, s, p);
    return 0;
}

template <int NW, int TM, int TN, int DEPTH>
static int launch_lin(const CGParams& p, hipStream_t s) {
    // uniformly strided rows: a Linear, or a 1x1 stride-1 convolution whose batch items are contiguous
    const bool uniform = p.KH * p.KW == 1 && p.stride == 1 && p.pad_h == 0 && p.pad_w == 0 && p.up == 0 &&
                         (p.M == p.rpb || p.a_bs == p.rpb * p.lda) && p.IH * p.IW == p.rpb &&
                         (p.C1 == 0 || p.M == p.rpb || p.a_bs2 == p.rpb * p.lda2) && p.in_act == 0;
    if (p.ln_mode) {
        AED_REQUIRE(uniform, "lin_gemm: the fused LayerNorm needs uniformly strided rows");
        return launch_lin_mode<NW, TM, TN, DEPTH, 1>(p, s);
    }
    if (uniform) return launch_lin_mode<NW, TM, TN, DEPTH, 0>(p, s);
    return launch_lin_mode<NW, TM, TN, DEPTH, 2>(p, s);
}

// cfg: 10 = 4 waves 32x32 | 11 = 8 waves 32x32 | 12 = 16 waves 32x32 | 13 = 4 waves 32x64 | 14 = 8 waves 32x64 |
//      15 = 4 waves 64x64 | 16 = 4 waves 64x32 | 17 = 8 waves 64x64 | 18 = 10 waves 32x32 | 19 = 12 waves 32x32
//      (10 / 12 waves split K = 640 / 384 -- the channel counts of U-Net levels 3 / 2 -- into equal chunk counts)
int launch_lin_gemm(const CGParams& p, int cfg, hipStream_t s) {
    AED_REQUIRE(p.Cin % 32 == 0 && p.lda % 4 == 0 && ((uintptr_t)p.A % 16) == 0 && ((uintptr_t)p.W % 16) == 0,
                "lin_gemm: needs Cin %% 32 == 0 and 16-byte aligned operands (Cin=%d lda=%d)", p.Cin, p.lda);
    AED_REQUIRE(p.ksplit <= 1, "lin_gemm: K is split across the wavefronts of a workgroup, not across workgroups");
    if (p.C1 > 0)
        AED_REQUIRE(p.A2 && p.C1 % 32 == 0 && p.C1 < p.Cin && p.lda2 % 4 == 0 && ((uintptr_t)p.A2 % 16) == 0,
                    "lin_gemm: bad two-source split C1=%d of Cin=%d", p.C1, p.Cin);
    if (p.w_bs || p.vec_bs || p.sm_group)
        AED_REQUIRE(p.rpb % 64 == 0 && p.M % p.rpb == 0, "lin_gemm: per-batch operands need 64-row aligned batch items");
    if (p.sm_group)
        AED_REQUIRE((p.sm_group == 8 || p.sm_group == 16 || p.sm_group == 32) && p.N % 32 == 0 && !p.res && !p.geglu &&
                        p.o_mul == 1 && p.o_add == 0 && p.out_bs == p.rpb && p.accumulate == 0,
                    "lin_gemm: grouped softmax needs groups of 8/16/32 columns, N %% 32 == 0 and a plain row layout");
    if (p.geglu)
        AED_REQUIRE(p.N % 64 == 0 && (cfg == 13 || cfg == 14 || cfg == 15 || cfg == 17) && !p.res && p.out_act == 0 &&
                        p.o_mul == 1 && p.o_add == 0 && (p.ln_mode || !p.rowvec),
                    "lin_gemm: GEGLU needs a 64-wide tile (cfg %d), packed N %% 64 == 0 and a plain epilogue", cfg);
    int rc = 0;
    switch (cfg) {
        case 10: rc = launch_lin<4, 1, 1, 2>(p, s); break;
        case 11: rc = launch_lin<8, 1, 1, 2>(p, s); break;
        case 12: rc = launch_lin<16, 1, 1, 1>(p, s); break;
        case 13: rc = launch_lin<4, 1, 2, 2>(p, s); break;
        case 14: rc = launch_lin<8, 1, 2, 2>(p, s); break;
        case 15: rc = launch_lin<4, 2, 2, 1>(p, s); break;
        case 16: rc = launch_lin<4, 2, 1, 2>(p, s); break;
        case 17: rc = launch_lin<8, 2, 2, 1>(p, s); break;
        case 18: rc = launch_lin<10, 1, 1, 2>(p, s); break;
        case 19: rc = launch_lin<12, 1, 1, 2>(p, s); break;
        default: AED_REQUIRE(false, "lin_gemm: bad tile cfg %d", cfg);
    }
    if (rc) return rc;
    AED_CHECK_HIP(hipGetLastError());
    return 0;
}
