// conv_gemm_x6.hip -- the implicit-GEMM convolution / linear layer of conv_gemm.hip with SPLIT-BF16 arithmetic.
//
// The product's arithmetic for the LDS-staged GEMMs since round 4 (built in round 3): selected per op by flag bit 2 of an
// AED_OP_CONV_GEMM record (include/aed.h); tapes built under tape.arith_mode("bf16x6") -- every U-Net / DiT / codec engine by
// default -- set it.  Same record, same operands, same epilogue as conv_gemm.hip -- only the contraction differs.
//
// Why: the fp32-input MFMA (v_mfma_f32_32x32x2_f32) runs at the fp32 VECTOR rate, 1/16 of the bf16 MFMA
// (MI355X_MICROARCH.md: 157 TF vs 2.5 PF).  An fp32 value is exactly the sum of three bf16 values (24 significand bits
// = 3 x 8): a = a0 + a1 + a2, |a1| <= 2^-8 |a|, |a2| <= 2^-16 |a|.  A product a*b is then the sum of nine piece products
// a_i*b_j, each EXACT in fp32 (8 x 8 bits), of relative size 2^-8(i+j).  Keeping the six with i+j <= 2 drops terms below
// 2^-23 |a||b| -- under the rounding of the fp32 accumulation itself (tools/bf16_split_study.py on the CPU and
// profiles/r03_x6_gemm.md on the GPU: vs fp64 the six-term result is as close as -- in fact slightly closer than -- the
// fp32-MFMA kernel's).  Six v_mfma_f32_32x32x16_bf16 (32 cycles each per SIMD) replace eight v_mfma_f32_32x32x2_f32 (64
// cycles each) per 16 k: 192 vs 512 matrix-pipe cycles, a 2.67x higher roof (419 TF/s fp32-equivalent) for the same fp32
// operands in HBM.  Measured at the U-Net's batch-200 shapes: 1.45-1.8x over conv_gemm.hip (180-210 TF/s, above the fp32
// MFMA peak); the kernel is then at ~44 % of the bf16 MFMA rate, the range of HIP-level bf16 GEMMs on this chip.
//
// Data path: operands stay fp32 in HBM (no second copy of weights or activations).  Loads are buffer loads: a lane whose
// element does not exist (conv zero padding, rows past M / N, a chunk past the block's k range) gets offset X6_OOB and
// the hardware returns 0 -- no masks or selects on the data.  The loader threads split each float4 into three packed-bf16
// pieces (v_cvt_pk_bf16_f32 RNE + scalar subtractions: 5.5 VALU per element; v_pk_add_f32 issues badly beside MFMAs) while
// writing the LDS stage.  LDS row = [hi 16 k | mid 16 k | lo 16 k | 16 B pad] = 112 B: the fragment read of one piece is
// one ds_read_b128 per lane (8 consecutive k), conflict-free with this stride (28 dwords: the 16 rows of a b128 lane
// group land on 16 distinct 4-bank windows).  A and W use the same k -> (lane half, element) map, so the result does not
// depend on the instruction's internal k order; the C/D map is the one of the fp32 32x32 MFMA (conv_gemm.hip's epilogue).
//
// Pipeline (as conv_gemm.hip): two LDS stages, chunk k+1 is split + written to the other stage while the MFMAs of
// chunk k run, one barrier per chunk, DEPTH further chunks in flight in registers.  A chunk is 16 k = one bf16 k-block.
// The chunk body is ONE basic block (branch-free address arithmetic, trip count rounded up to DEPTH with zero chunks), and
// VALU-only sched_group_barrier hints spread the split between the MFMAs (+3-4 % measured).
//
// Tried and dropped (profiles/r03_x6_gemm.md): reading the fragments of chunk k+1 while the MFMAs of chunk k run from a
// second register set (-2 %: LDS latency behind the barrier is not what is exposed), 256x256 / 256x128 tiles with ONE wave per SIMD and the accumulators in
// AGPRs (4x slower as compiled), a row-permuted loader for conflict-free ds_write_b64 (-4 %), LDS-write / VMEM groups in
// the hints (the solver gives up), packed subtractions (-10 %), flat loads with LDS-side zero masking (-3 %).
#include "cg_params.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

int launch_conv_gemm_x6(const aed_op* op, hipStream_t s);

constexpr int X6_BK = 16;           // fp32 k per chunk of the original form (one bf16 MFMA k-block per barrier)
// Round 5: BK = 32 ("wide chunks", op flag bit 8) -- two k-blocks per LDS stage and barrier.  After a barrier a wave must read its
// fragments from LDS before its first MFMA can issue (~150-250 cycles in which BOTH waves of a SIMD of a 512-thread workgroup
// idle the matrix pipe, visible in the ISA as ds_read x 12 -> s_waitcnt -> v_mfma); 48 instead of 24 MFMAs per wave between
// barriers halve that bubble's share.  LDS row = [hi BK k | mid BK k | lo BK k | 16 B pad]: 112 B at BK 16, 208 B at BK 32 (52
// dwords: the 16 rows of a ds_read_b128 lane group still land on 16 distinct 4-bank windows); two stages of the 256x128 /
// 128x256 tiles are 159 744 B of the CU's 163 840.

// (x0, x1) -> packed bf16 pieces {hi, mid, lo}; x == hi + mid + lo exactly (element 0 in the low half).  Scalar subtractions:
// v_pk_add_f32 issues badly next to MFMAs (MI355X_MICROARCH.md, filler prices; measured -10 %)
__device__ __forceinline__ void x6_split_pair(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    h = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){x0, x1}, bf16x2));
    float r0 = x0 - __uint_as_float(h << 16);
    float r1 = x1 - __uint_as_float(h & 0xffff0000u);
    asm volatile("" : "+v"(r0), "+v"(r1));              // keep the two subtractions scalar (no SLP packing)
    m = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){r0, r1}, bf16x2));
    float q0 = r0 - __uint_as_float(m << 16);
    float q1 = r1 - __uint_as_float(m & 0xffff0000u);
    asm volatile("" : "+v"(q0), "+v"(q1));
    l = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){q0, q1}, bf16x2));
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned X6_OOB = 0x80000000u;        // byte offset past every buffer (num_records 0x7ffffff0): the load returns 0

// Compile-time interleave request for one chunk body: NM MFMAs with NV VALU instructions spread evenly between them
// (sched_group_barrier wants immediates, hence the recursion).
template <int G, int NM, int NV>
__device__ __forceinline__ void x6_hints() {
    if constexpr (G < NM) {
        constexpr int v = (NV * (G + 1)) / NM - (NV * G) / NM;
        if constexpr (v > 0) __builtin_amdgcn_sched_group_barrier(0x002, v, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        x6_hints<G + 1, NM, NV>();
    }
}

// SCHED: 0 = leave the instruction order to the compiler; 1 = VALU-between-MFMA interleave hints in the main loop (x6_hints)
// NTERMS: 6 = the product arithmetic; 3 = only the terms of relative size >= 2^-8 (DIAGNOSTIC: tells how much of the kernel time
// is matrix-pipe time; ~4e-6 rel error, never used by a product path)
template <int BM, int BN, int WROWS, int WCOLS, int DEPTH, bool PLAIN, int SCHED, int NTERMS, int BK = X6_BK>
__global__ __launch_bounds__(64 * WROWS * WCOLS, 2) void conv_gemm_x6_kernel(CGParams p) {
    constexpr int NT = 64 * WROWS * WCOLS;
    constexpr int WM = BM / WROWS, WN = BN / WCOLS;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int TPR = BK / 4;             // loader threads per row (float4 each)
    constexpr int RPP = NT / TPR;           // rows per loader pass
    constexpr int PA = BM / RPP, PB = BN / RPP;
    constexpr int PQ = BK / 8;              // uint4 per piece of a row (8 bf16 each)
    constexpr int X6_ROWQ = 3 * PQ + 1;     // uint4 (16 B) per LDS row: 3 pieces + 1 pad
    constexpr int KB = BK / 16;             // bf16 MFMA k-blocks per chunk
    constexpr int STAGE = (BM + BN) * X6_ROWQ;      // uint4 per operand stage (A rows then W rows)
    static_assert(BK == 16 || BK == 32, "chunk width");
    static_assert(TM >= 1 && TN >= 1 && PA >= 1 && PB >= 1, "tile");
    static_assert(BM % RPP == 0 && BN % RPP == 0, "loader passes");

    __shared__ uint4 lds[2 * STAGE];
    const unsigned nx = gridDim.x, ny = gridDim.y, bz = blockIdx.z, gz = gridDim.z;
    const unsigned orig = blockIdx.y * nx + blockIdx.x;     // linear workgroup id in launch order (consecutive ids -> consecutive XCDs)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave / WCOLS, wc = wave % WCOLS;
    // XCD-aware tile order (see conv_gemm.hip): XCD x gets a contiguous range of tile ids, n fastest
    int tile_y = (int)(orig / nx), tile_x = (int)(orig - (unsigned)tile_y * nx);
    if (gz == 1) {
        const unsigned nwg = nx * ny;
        const unsigned xcd = orig & 7u, q = nwg >> 3, r = nwg & 7u;
        const unsigned id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
        if (p.gm > 1) {     // grouped order: p.gm row panels per group, m fastest inside a group (bijective for any grid)
            const unsigned per_group = (unsigned)p.gm * nx;
            const unsigned g = id / per_group, first = g * (unsigned)p.gm;
            const unsigned gsz = min((unsigned)p.gm, ny - first);
            const unsigned within = id - g * per_group;
            tile_x = (int)(within / gsz);
            tile_y = (int)(first + (within - (unsigned)tile_x * gsz));
        } else {
            tile_y = (int)(id / nx);
            tile_x = (int)(id - (unsigned)tile_y * nx);
        }
    }
    const int m0 = tile_y * BM;
    const int n0 = tile_x * BN;

    int kc_begin = 0, kc_end = p.nchunks;
    if (p.ksplit > 1) {
        const int per = (p.nchunks + p.ksplit - 1) / p.ksplit;
        kc_begin = (int)bz * per;
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
        int b = 0, r = mm, oy = mm, ox = 0;
        if (p.rpb != p.M || p.OW != 1) {        // (a Linear is one batch item of M x 1 pixels: no run-time divisions in its prologue)
            b = mm / p.rpb;
            r = mm - b * p.rpb;
            oy = r / p.OW;
            ox = r - oy * p.OW;
        }
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
    // [group of p.kgroup channels][tap][chunk within the group] (cg_params.h): pf_cg = first channel of the group
    int pf_cg, pf_sub, pf_ty, pf_tx;
    const int gq = p.kgroup / BK;             // chunks per (group, tap)
    if (kc_begin == 0) {
        pf_cg = pf_sub = pf_ty = pf_tx = 0;
    } else {
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
    float ln_s1[PA], ln_s2[PA];
#pragma unroll
    for (int q = 0; q < PA; ++q) { ln_s1[q] = 0.f; ln_s2[q] = 0.f; }

    // Buffer descriptors over "everything below 2 GB" (the launcher checks the operands fit): an offset of X6_OOB is out of
    // range and reads 0.  Branch-free on purpose: the whole chunk body (split of chunk k+1, loads of chunk k+1+DEPTH, MFMAs
    // of chunk k) is ONE basic block, so VALU / LDS / VMEM work can sit in the shadow of the MFMAs.
    const __amdgpu_buffer_rsrc_t srd_a = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, 0x7ffffff0, 0x00020000);
    const __amdgpu_buffer_rsrc_t srd_a2 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A2 ? p.A2 : p.A), 0, 0x7ffffff0, 0x00020000);
    const __amdgpu_buffer_rsrc_t srd_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, 0x7ffffff0, 0x00020000);
    auto as_f4 = [](u32x4 v) {
        return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
    };
    auto prefetch = [&](int kc, float4 (&ra)[PA], float4 (&rb)[PB]) {
        int c0 = pf_cg + pf_sub * BK;
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
        // a chunk past this block's k range, an out-of-image tap, a row past M or N: the lane's offset is X6_OOB
        const bool dead = kc >= kc_end;
#pragma unroll
        for (int q = 0; q < PA; ++q) {
            const int iy = ay0[q] + dy, ix = ax0[q] + dx;
            const bool ok = avalid[q] & ((unsigned)iy < (unsigned)vIH) & ((unsigned)ix < (unsigned)vIW) & !dead;
            const unsigned base = second ? abase2[q] : abase[q];
            const unsigned off = (base + (unsigned)((iy >> p.up) * p.IW + (ix >> p.up)) * ld + (unsigned)c0) * 4u;
            ra[q] = as_f4(__builtin_amdgcn_raw_buffer_load_b128(second ? srd_a2 : srd_a, ok ? off : X6_OOB, 0, 0));
        }
#pragma unroll
        for (int q = 0; q < PB; ++q)
            rb[q] = as_f4(__builtin_amdgcn_raw_buffer_load_b128(
                srd_w, (wvalid[q] & !dead) ? (wbase[q] + k0) * 4u : X6_OOB, 0, 0));
    };

    // one fetched float4 -> three 8-byte piece groups of its LDS row
    auto put_row = [&](uint4* st, int row, float4 v, bool is_w = false) {
        unsigned h0, m0_, l0, h1, m1, l1;
#ifdef X6_DIAG_NOSPLIT
        // DIAGNOSTIC BUILD ONLY (tools/leases/r06_l17.sh; never in libaed.so): what would a loader WITHOUT the split cost?  The hi
        // piece is computed (one v_cvt_pk per pair) and stored three times -- wrong values, representative time: an upper bound
        // on what pre-split operands in HBM could give (1 = the W rows only, 2 = A and W), before their 1.5x load bytes.
        if (X6_DIAG_NOSPLIT == 2 || is_w) {
            h0 = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){v.x, v.y}, bf16x2));
            h1 = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){v.z, v.w}, bf16x2));
            m0_ = l0 = h0;
            m1 = l1 = h1;
        } else
#endif
        {
            x6_split_pair(v.x, v.y, h0, m0_, l0);
            x6_split_pair(v.z, v.w, h1, m1, l1);
        }
        uint2* dst = reinterpret_cast<uint2*>(st + row * X6_ROWQ) + lq;      // piece pl starts at uint2 index 2 * PQ * pl
        dst[0] = make_uint2(h0, h1);
        dst[2 * PQ] = make_uint2(m0_, m1);
        dst[4 * PQ] = make_uint2(l0, l1);
    };

    auto stage_write = [&](uint4* st, const float4 (&ra)[PA], const float4 (&rb)[PB]) {
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
            put_row(st, lrow + RPP * q, v);
        }
#pragma unroll
        for (int q = 0; q < PB; ++q) put_row(st, BM + lrow + RPP * q, rb[q], true);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int fi = lane & 31;        // fragment row (A: m, W: n)
    const int fh = lane >> 5;        // which 8 of a k-block's 16 k
    const int a_frag = (wr * WM + fi) * X6_ROWQ + fh;
    const int b_frag = (BM + wc * WN + fi) * X6_ROWQ + fh;

    bf16x8 af[TM][3], bw[TN][3];
    // fragments of k-block kb of the staged chunk: piece pl of a row starts at uint4 PQ * pl, its k-block kb at + 2 * kb
    auto load_frags = [&](const uint4* st, int kb, bf16x8 (&fa)[TM][3], bf16x8 (&fb)[TN][3]) {
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                fa[a][pl] = __builtin_bit_cast(bf16x8, st[a_frag + a * 32 * X6_ROWQ + PQ * pl + 2 * kb]);
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                fb[b][pl] = __builtin_bit_cast(bf16x8, st[b_frag + b * 32 * X6_ROWQ + PQ * pl + 2 * kb]);
    };
    // the six piece products with i + j <= 2, smallest first
    auto do_mfmas = [&](const bf16x8 (&fa)[TM][3], const bf16x8 (&fb)[TN][3]) {
        constexpr int TA[6] = {0, 2, 1, 0, 1, 0};
        constexpr int TB[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
        for (int t = 6 - NTERMS; t < 6; ++t)
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][TA[t]], fb[b][TB[t]], acc[a][b], 0, 0, 0);
    };
    auto hints = [&]() {
        if constexpr (SCHED == 1) {
            // a wave hides ~5 single-issue instructions per 32-cycle MFMA: spread the split of the chunk being staged (5.5 VALU
            // per element + addresses) between the MFMAs.  VALU groups only: asking for LDS-write / load groups as well makes
            // the solver give up and emit all VALU first.
            constexpr int NM = TM * TN * NTERMS * KB;
            x6_hints<0, NM, NM * (((PA + PB) * 22 + 40 + NM - 1) / NM)>();
        }
    };

#pragma unroll
    for (int d = 0; d < DEPTH; ++d) prefetch(kc_begin + d, rbuf_a[d], rbuf_b[d]);
    stage_write(lds, rbuf_a[0], rbuf_b[0]);
    prefetch(kc_begin + DEPTH, rbuf_a[0], rbuf_b[0]);
    __syncthreads();
    int cur = 0;
    // The trip count is rounded up to a multiple of DEPTH (static register slots, no exit inside the unrolled body): a chunk
    // past kc_end was loaded as zeros and its MFMAs add nothing.  Per iteration, in program order: fragment reads of chunk kc
    // (stage cur) -> split + LDS write of chunk kc+1 (other stage) -> loads of chunk kc+1+DEPTH -> MFMAs of chunk kc -> barrier.
    for (int kc0 = kc_begin; kc0 < kc_end; kc0 += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int kc = kc0 + d;
            const int sl = (d + 1) % DEPTH;     // chunk kc+1 sits in this register slot
            load_frags(lds + cur * STAGE, 0, af, bw);
            stage_write(lds + (cur ^ 1) * STAGE, rbuf_a[sl], rbuf_b[sl]);
            prefetch(kc + 1 + DEPTH, rbuf_a[sl], rbuf_b[sl]);
            do_mfmas(af, bw);
            if constexpr (KB == 2) {            // second k-block of the wide chunk: fragments read under the first block's MFMAs
                bf16x8 af2[TM][3], bw2[TN][3];
                load_frags(lds + cur * STAGE, 1, af2, bw2);
                do_mfmas(af2, bw2);
            }
            hints();
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
            const bool rows_are_m = p.out_bs == p.rpb;      // (every FF1 of the engines: output row = m, as in the simple-rows epilogue)
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
                    float out[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float val = acc[a][b][r], gate = acc[a][b + 1][r];
                        if (p.ln_mode) {
                            const int lr = wr * WM + a * 32 + 4 * fh + (r & 3) + 8 * (r >> 2);
                            const float mean = ln_stat[2 * lr], rstd = ln_stat[2 * lr + 1];
                            val = rstd * (val - mean * sv);
                            gate = rstd * (gate - mean * sg);
                        }
                        val += bv;
                        gate += bg;
                        out[r] = val * glu_gate(gate, p.geglu);
                    }
                    if (rows_are_m) {
                        float* cp = p.C + nf;
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int m = mbase + (r & 3) + 8 * (r >> 2);
                            if (m < p.M) cp[(unsigned)m * (unsigned)p.ldc] = out[r];
                        }
                    } else {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int m = mbase + (r & 3) + 8 * (r >> 2);
                            if (m < p.M) {
                                const int bb = m / p.rpb;
                                const unsigned row = (unsigned)bb * (unsigned)p.out_bs + (unsigned)(m - bb * p.rpb);
                                p.C[row * (unsigned)p.ldc + nf] = out[r];
                            }
                        }
                    }
                }
            return;
        }
    }
    // ---- simple rows (round 6): output row = m (every Linear and stride-1 convolution of the U-Net / DiT engines: no row scatter, no
    // accumulate mode).  The general epilogue below spends ~30 instructions per output on row arithmetic that is the identity here;
    // at the batch-200 forward's short-K Linears (K = 256 / 384: 16-24 chunks) that was a third of a tile's time
    // (profiles/r06_short_k.md).  Same operations on the values, in the same order: bit-identical.
    if (p.ksplit <= 1 && p.o_mul == 1 && p.o_add == 0 && p.out_bs == p.rpb && p.o_len == p.rpb && p.accumulate == 0 && !(p.diag & 1)) {
        const bool has_rv = p.rowvec != nullptr && !p.ln_mode;
        const int bmax = (p.M - 1) / p.rpb;
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b) {
                const int n = n0 + wc * WN + b * 32 + fi;
                const int mbase = m0 + wr * WM + a * 32 + 4 * fh;
                if (n >= p.N) continue;
                const float bias_v = p.bias ? p.bias[n] : 0.f;
                float val[16], rv[16];
                if (p.res) {                // requested first: in flight while the values are finished
                    const float* rp = p.res + n;
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        rv[r] = rp[(unsigned)min(mbase + (r & 3) + 8 * (r >> 2), p.M - 1) * (unsigned)p.ldr];
                }
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
                if (has_rv) {               // per-batch-item row vector (the resnets' time-embedding row)
                    const int mb = min(mbase, p.M - 1), b0 = mb / p.rpb, q0 = mb - b0 * p.rpb;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int dm = (r & 3) + 8 * (r >> 2);
                        int bb;
                        if (p.rpb >= 32) bb = (q0 + dm >= p.rpb) ? b0 + 1 : b0;
                        else bb = min(mbase + dm, p.M - 1) / p.rpb;
                        val[r] += p.rowvec[(unsigned)min(bb, bmax) * (unsigned)p.ld_rv + n];
                    }
                }
                if (p.res) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) val[r] += rv[r];
                }
                if (p.out_act != AED_ACT_NONE) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) val[r] = aed_apply_act(val[r], p.out_act, p.out_p);
                }
                float* cp = p.C + n;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = mbase + (r & 3) + 8 * (r >> 2);
                    if (m < p.M) cp[(unsigned)m * (unsigned)p.ldc] = val[r];
                }
            }
        return;
    }
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const int n = n0 + wc * WN + b * 32 + fi;
            const int mbase = m0 + wr * WM + a * 32 + 4 * fh;
            if (n >= p.N) continue;
            if (p.ksplit > 1) {
                float* wsp = p.ws + ((size_t)bz * p.M + mbase) * p.N + n;
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

template <int BM, int BN, int WR, int WC, int DEPTH, bool DIAG, int BK = X6_BK>
static int x6_launch(const CGParams& p, bool plain, int sched, int terms, hipStream_t s) {
    dim3 grid(aed_cdiv(p.N, BN), aed_cdiv(p.M, BM), p.ksplit);
    dim3 block(64 * WR * WC);
#define X6_GO(PL, SC, NT_) hipLaunchKernelGGL((conv_gemm_x6_kernel<BM, BN, WR, WC, DEPTH, PL, SC, NT_, BK>), grid, block, 0, s, p)
    if (terms == 3) {
        if constexpr (DIAG) {
            if (plain) { X6_GO(true, 1, 3); return 0; }
        }
        AED_REQUIRE(false, "conv_gemm_x6: the three-term diagnostic exists for tiles 1 and 8 with a plain A operand");
    }
    if (!plain) X6_GO(false, 1, 6);
    else if (sched) X6_GO(true, 1, 6);
    else X6_GO(true, 0, 6);
#undef X6_GO
    return 0;
}

// Tile codes (i[29]): 1 = 128x128, 2 = 128x64, 3 = 64x128, 4 = 64x64 (256 threads, two workgroups per CU); 8 = 256x128,
// 9 = 128x256 (512 threads: one workgroup per CU, two waves per SIMD).  0 = pick.  Shapes this kernel does not take (channel
// counts that are not a multiple of 16, unaligned or > 2 GB operands, the skinny tiles 5 / 6, the latency-regime tiles >= 10,
// per-batch weights / grouped softmax) run the fp32 path.
// Flag bit 3 (8): WITHOUT it the compiler orders the chunk body itself (A/B switch; tapes set it).
// Flag bit 4 (16): DIAGNOSTIC three-term arithmetic (~4e-6 rel error; tells how much of the kernel time is matrix-pipe time).
int launch_conv_gemm_x6(const aed_op* op, hipStream_t s) {
    const int32_t* i = op->i;
    const int Cin = i[11];
    int cfg = i[29];
    const long long batch = i[0] / (i[9] * i[10] > 0 ? i[9] * i[10] : 1);
    const bool fits = (Cin % X6_BK == 0) && (i[3] % 4 == 0) && ((uintptr_t)op->p[0] % 16 == 0) &&
                      ((uintptr_t)op->p[1] % 16 == 0) && cfg < 10 && cfg != 5 && cfg != 6 && cfg != 7 && i[36] == 0 &&
                      i[37] == 0 && i[38] <= 1 && i[39] == 0 &&
                      // buffer loads address bytes below 2 GB of every operand
                      batch * i[20] + (long long)i[7] * i[8] * i[3] < (1LL << 29) && (long long)i[1] * i[2] < (1LL << 29) &&
                      (i[32] == 0 || batch * i[34] + (long long)i[7] * i[8] * i[33] < (1LL << 29));
    if (!fits) {
        // the fp32 launcher does not know the x6-only tile codes 8 / 9 (256x128, 128x256): hand it the 128x128 tile
        if (cfg == 8 || cfg == 9) {
            aed_op fp32_op = *op;
            fp32_op.i[29] = 1;
            return launch_conv_gemm(&fp32_op, s);
        }
        return launch_conv_gemm(op, s);
    }
    // flag bit 8 (256): wide chunks (BK = 32) on the 512-thread tiles 8 / 9, where two stages fill the CU's LDS; needs 32 | Cin
    const bool wide = (op->flags & 256) && (cfg == 8 || cfg == 9) && Cin % 32 == 0 && !(op->flags & 16);   // (the three-term
    //                  diagnostic, flag bit 4, only has the BK = 16 kernel: never count its K in 32-wide chunks)
    // (flag bit 9, the deep-prefetch variant of round 5 -- DEPTH 4 x BK 32 on tiles 2 / 3 / 4, 0.8-7 % on ten shapes, never adopted --
    // left the library in round 6: profiles/r05_small_m.md section 5, git history)
    CGParams p;
    int rc = cg_fill_params(op, p, wide ? 32 : X6_BK);
    if (rc) return rc;
    if (p.ksplit > (p.K + 31) / 32) p.ksplit = (p.K + 31) / 32;     // launch_splitk_reduce clamps with 32-wide chunks
    if (cfg == 0) {
        const int cus = aed_num_cus();
        auto blocks = [&](int bm, int bn) { return (long)aed_cdiv(p.M, bm) * aed_cdiv(p.N, bn) * p.ksplit; };
        if (p.N >= 512 && p.N % 256 == 0 && blocks(128, 256) >= 2L * cus) cfg = 9;      // wide Linears (qkv, FF1)
        else if (blocks(256, 128) >= 2L * cus && p.N >= 128) cfg = 8;
        else if (blocks(128, 128) >= (long)cus && p.N >= 128) cfg = 1;
        else if (blocks(128, 64) >= 2L * cus) cfg = 2;
        else cfg = 4;
    }
    // Tile order (round 6).  The workgroups that run together on one XCD (its CUs x workgroups per CU) should touch as few
    // distinct operand bytes as possible: g row panels x (resident / g) column tiles with g ~ sqrt(resident * BN / BM).  With n
    // fastest (rounds 1-5) they covered ~1 panel x every column tile, and a wide Linear's W (FF1 at level 3: 13 MB against 4 MB of
    // L2) was streamed again for every row panel.  Flag bit 10 (1024) keeps the old order (A/B).
    if (!(op->flags & 1024) && p.ksplit <= 1) {
        static const int BMs[10] = {0, 128, 128, 64, 64, 0, 0, 0, 256, 128}, BNs[10] = {0, 128, 64, 128, 64, 0, 0, 0, 128, 256};
        const int nx = aed_cdiv(p.N, BNs[cfg]), ny = aed_cdiv(p.M, BMs[cfg]);
        const int resident = ((aed_num_cus() >> ((op->flags >> 16) & 3)) / 8) * ((cfg == 8 || cfg == 9) ? 1 : 2);
        if (nx > 1 && ny > 1) {
            int g = 1;
            while (g * 2 <= ny && (long)(g * 2) * (g * 2) * BMs[cfg] <= (long)resident * BNs[cfg]) g *= 2;
            if (resident / g < nx) p.gm = g;        // (n fastest already keeps every column tile of few panels together otherwise)
            if (op->flags & 0x3800) p.gm = 1 << ((op->flags >> 11) & 7);       // flag bits 11-13: forced group height (sweeps)
        }
    }
    const bool plain = p.in_act == 0 && p.ln_mode == 0;
    const int sched = (op->flags & 8) ? 1 : 0;
    const int terms = (op->flags & 16) ? 3 : 6;
    if (p.geglu) AED_REQUIRE(cfg == 1 || cfg == 3 || cfg == 8 || cfg == 9, "conv_gemm_x6: the GEGLU epilogue needs 64-wide wave tiles (cfg %d)", cfg);
    switch (cfg) {
        case 1: rc = x6_launch<128, 128, 2, 2, 2, true>(p, plain, sched, terms, s); break;
        case 2: rc = x6_launch<128, 64, 2, 2, 2, false>(p, plain, sched, terms, s); break;
        case 3: rc = x6_launch<64, 128, 2, 2, 2, false>(p, plain, sched, terms, s); break;
        case 4: rc = x6_launch<64, 64, 2, 2, 2, false>(p, plain, sched, terms, s); break;
        case 8: rc = wide && terms == 6 ? x6_launch<256, 128, 4, 2, 2, false, 32>(p, plain, sched, terms, s)
                                        : x6_launch<256, 128, 4, 2, 2, true>(p, plain, sched, terms, s); break;
        case 9: rc = wide && terms == 6 ? x6_launch<128, 256, 2, 4, 2, false, 32>(p, plain, sched, terms, s)
                                        : x6_launch<128, 256, 2, 4, 2, false>(p, plain, sched, terms, s); break;
        default: AED_REQUIRE(false, "conv_gemm_x6: bad tile cfg %d", cfg);
    }
    if (rc) return rc;
    AED_CHECK_HIP(hipGetLastError());
    if (p.ksplit > 1) return launch_splitk_reduce(op, s);
    return 0;
}
