// attention_x6.hip -- the transposed-score flash attention of attention.hip (attention_t_kernel, KSPLIT = 1) with SPLIT-BF16
// arithmetic: softmax(Q K^T * scale + key_bias) V for long key sequences in the throughput regime (the timestep-batched
// inversion's self-attention at U-Net batch 2G: 19 % of that forward on the fp32 MFMAs; the Stable Audio DiT's batches).
//
// Arithmetic (as conv_gemm_x6.hip): an fp32 value is EXACTLY the sum of three bf16 values; a product is nine exact piece
// products, the six of relative size >= 2^-16 are accumulated in fp32 -- as close to fp64 as the fp32 MFMA chain.  Both
// contractions take it: S^T = K Q^T (K, Q split) and O^T += V^T P^T (V and the probabilities split; p in [0, 1] is an fp32
// number like any other).  Per 32-key tile and 32 queries with d_head 32 that is 24 v_mfma_f32_32x32x16_bf16 (768 matrix-pipe
// cycles per wave) against 32 v_mfma_f32_32x32x2_f32 (2048); max / exp / sum stay fp32 VALU as before.
//
// What makes it pay (NOTES.md, round 3: splitting K and V per wave makes the kernel VALU-bound): the 4 waves of a workgroup
// take 4 consecutive query tiles of ONE (batch, head), so the K / V tile is fetched from HBM / L2 ONCE per workgroup (the
// fp32 kernel fetches it once per wave), split ONCE (22 + 22 VALU per thread and tile instead of 88 + 88 per wave), and
// left in LDS in MFMA OPERAND ORDER: a fragment is one conflict-free ds_read_b128 per lane.
//   K stage:  [d block db][piece][key 32][lane half h][e 8] bf16      element = K[key][16 db + 8 h + e]
//   V stage:  [key block kb][piece][d tile dt][d 32][h][e 8] bf16      element = V[16 kb + (e&3) + 8 (e>>2) + 4 h][32 dt + d]
// The V order is dictated by the accumulator layout of S^T: register r of lane (query, h) holds key (r&3) + 8 (r>>2) + 4 h,
// so registers 8 kb .. 8 kb + 7 ARE the B operand of key block kb (the probabilities still go from the accumulator straight
// into the second MFMA, through three packed-bf16 pieces instead of as fp32).  A loader thread fetches 4 CONSECUTIVE KEYS of
// one channel (four coalesced 4-byte loads: the 32 lanes of a half-wave cover one 128-byte row segment each) so that its 4
// values are 4 consecutive e of the operand order: one 8-byte LDS write per piece, no transposing scatter.
// Pipeline: two LDS stages, one workgroup barrier per key tile; tile t+1 is split + written while the MFMAs of tile t run, tiles
// t+2 / t+3 are in flight in registers.
#include "attn_params.h"

typedef __bf16 abf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 abf16x2 __attribute__((ext_vector_type(2)));
typedef float af32x2 __attribute__((ext_vector_type(2)));
typedef float af32x16 __attribute__((ext_vector_type(16)));

// (x0, x1) -> packed bf16 pieces {hi, mid, lo}; x == hi + mid + lo exactly (element 0 in the low half)
__device__ __forceinline__ void ax6_split_pair(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    h = __builtin_bit_cast(unsigned, __builtin_convertvector((af32x2){x0, x1}, abf16x2));
    float r0 = x0 - __uint_as_float(h << 16);
    float r1 = x1 - __uint_as_float(h & 0xffff0000u);
    asm volatile("" : "+v"(r0), "+v"(r1));              // scalar subtractions (v_pk_add_f32 issues badly beside MFMAs)
    m = __builtin_bit_cast(unsigned, __builtin_convertvector((af32x2){r0, r1}, abf16x2));
    float q0 = r0 - __uint_as_float(m << 16);
    float q1 = r1 - __uint_as_float(m & 0xffff0000u);
    asm volatile("" : "+v"(q0), "+v"(q1));
    l = __builtin_bit_cast(unsigned, __builtin_convertvector((af32x2){q0, q1}, abf16x2));
}

// eight floats -> the three 8-element bf16 fragments
__device__ __forceinline__ void ax6_split8(const float (&x)[8], abf16x8 (&f)[3]) {
    uint4 h, m, l;
    ax6_split_pair(x[0], x[1], h.x, m.x, l.x);
    ax6_split_pair(x[2], x[3], h.y, m.y, l.y);
    ax6_split_pair(x[4], x[5], h.z, m.z, l.z);
    ax6_split_pair(x[6], x[7], h.w, m.w, l.w);
    f[0] = __builtin_bit_cast(abf16x8, h);
    f[1] = __builtin_bit_cast(abf16x8, m);
    f[2] = __builtin_bit_cast(abf16x8, l);
}

// acc += sum over the six piece products with i + j <= 2, smallest first (a: A-operand pieces, b: B-operand pieces)
__device__ __forceinline__ void ax6_mfma6(af32x16& acc, const abf16x8 (&a)[3], const abf16x8 (&b)[3]) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
}

template <int D>
__global__ __launch_bounds__(256, 2) void attention_x6_kernel(AttnParams p) {
    constexpr int NDB = D / 16;                    // 16-wide d blocks of the S^T contraction
    constexpr int DT = (D + 31) / 32;              // 32-row tiles of O^T
    constexpr int KQ = NDB * 3 * 64;               // uint4 per K stage ([db][piece][key 32][h 2] x 16 B)
    constexpr int VQ = 2 * 3 * DT * 64;            // uint4 per V stage ([kb][piece][dt][d 32][h 2] x 16 B)
    constexpr int STAGE = KQ + VQ;
    constexpr int C4 = D / 4;                      // float4 per K row
    constexpr int NK = (8 * D + 255) / 256;        // K loader items (one float4) per thread
    constexpr int NV = (8 * D + 255) / 256;        // V loader items (4 keys x 1 channel) per thread
    static_assert(D % 16 == 0 && 2 * STAGE * 16 <= 65536, "head dim");
    __shared__ uint4 lds[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fi = lane & 31, fh = lane >> 5;
    // XCD-aware order (attention.hip): the query tiles of one (batch, head) run on one XCD
    int qt, head, b;
    {
        const unsigned nx = gridDim.x, ny = gridDim.y, nwg = nx * ny * gridDim.z;
        const unsigned orig = blockIdx.x + nx * (blockIdx.y + ny * blockIdx.z);
        const unsigned xcd = orig & 7u, q = nwg >> 3, r = nwg & 7u;
        const unsigned id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
        qt = (int)(id % nx);
        head = (int)((id / nx) % ny);
        b = (int)(id / (nx * ny));
    }
    const int q0 = qt * 128 + wave * 32;
    const int hoff = head * D;
    const float* Q = p.q + (size_t)b * p.bsq + hoff;
    const float* K = p.k + (size_t)b * p.bsk + hoff;
    const float* V = p.v + (size_t)b * p.bsv + hoff;
    const float* bias = p.bias ? p.bias + (size_t)b * p.ld_bias : nullptr;

    // Q^T pieces (B operand of S^T): lane (query fi, half fh) holds Q[q0 + fi][16 db + 8 fh + e], pre-scaled
    abf16x8 qp[NDB][3];
    {
        const float* qrow = Q + (size_t)min(q0 + fi, p.Nq - 1) * p.ldq + 8 * fh;
#pragma unroll
        for (int db = 0; db < NDB; ++db) {
            const float4 a = *reinterpret_cast<const float4*>(qrow + 16 * db);
            const float4 c = *reinterpret_cast<const float4*>(qrow + 16 * db + 4);
            const float x[8] = {a.x * p.scale, a.y * p.scale, a.z * p.scale, a.w * p.scale,
                                c.x * p.scale, c.y * p.scale, c.z * p.scale, c.w * p.scale};
            ax6_split8(x, qp[db]);
        }
    }
    af32x16 oacc[DT];
#pragma unroll
    for (int c = 0; c < DT; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[c][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;          // l_run: this lane's 16-key share; the halves are added at the end

    const int ntiles = (p.Nk + 31) / 32;
    float4 kreg[2][NK];
    float vreg[2][NV][4];
    auto prefetch = [&](int t, float4 (&kr)[NK], float (&vr)[NV][4]) {
        const int k0 = min(t, ntiles - 1) * 32;    // dead prefetches stay in bounds
#pragma unroll
        for (int i = 0; i < NK; ++i) {
            const int item = min(tid + 256 * i, 8 * D - 1);
            const int key = min(k0 + item / C4, p.Nk - 1);
            kr[i] = *reinterpret_cast<const float4*>(K + (size_t)key * p.ldk + 4 * (item % C4));
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int item = min(tid + 256 * i, 8 * D - 1);
            const int kg = item / D, d = item - kg * D;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                vr[i][j] = V[(size_t)min(k0 + 4 * kg + j, p.Nk - 1) * p.ldv + d];
        }
    };
    auto stage_write = [&](uint4* st, const float4 (&kr)[NK], const float (&vr)[NV][4]) {
        uint2* k2 = reinterpret_cast<uint2*>(st);
        uint2* v2 = reinterpret_cast<uint2*>(st + KQ);
#pragma unroll
        for (int i = 0; i < NK; ++i) {
            const int item = tid + 256 * i;
            if (8 * D % 256 != 0 && item >= 8 * D) break;
            const int key = item / C4, c4 = item % C4;
            unsigned h0, m0, l0, h1, m1, l1;
            ax6_split_pair(kr[i].x, kr[i].y, h0, m0, l0);
            ax6_split_pair(kr[i].z, kr[i].w, h1, m1, l1);
            // K[key][4 c4 + j]: d block c4 >> 2, lane half (c4 >> 1) & 1, e = 4 (c4 & 1) + j
            uint2* dst = k2 + ((c4 >> 2) * 3 * 32 + key) * 4 + ((c4 >> 1) & 1) * 2 + (c4 & 1);
            dst[0] = make_uint2(h0, h1);
            dst[32 * 4] = make_uint2(m0, m1);
            dst[2 * 32 * 4] = make_uint2(l0, l1);
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int item = tid + 256 * i;
            if (8 * D % 256 != 0 && item >= 8 * D) break;
            const int kg = item / D, d = item - kg * D;
            unsigned h0, m0, l0, h1, m1, l1;
            ax6_split_pair(vr[i][0], vr[i][1], h0, m0, l0);
            ax6_split_pair(vr[i][2], vr[i][3], h1, m1, l1);
            // keys 4 kg + j of channel d: key block kg >> 2, g = kg & 3 -> lane half g & 1, e = 4 (g >> 1) + j
            const int kb = kg >> 2, g = kg & 3;
            uint2* dst = v2 + (((kb * 3) * DT + (d >> 5)) * 32 + (d & 31)) * 4 + (g & 1) * 2 + (g >> 1);
            dst[0] = make_uint2(h0, h1);
            dst[DT * 32 * 4] = make_uint2(m0, m1);
            dst[2 * DT * 32 * 4] = make_uint2(l0, l1);
        }
    };
    if constexpr (DT * 32 > D) {                   // rows d >= D of V^T are never staged: keep them zero (both stages)
        for (int e = tid; e < 2 * STAGE; e += 256) lds[e] = make_uint4(0u, 0u, 0u, 0u);
        __syncthreads();
    }

    prefetch(0, kreg[0], vreg[0]);
    prefetch(1, kreg[1], vreg[1]);
    stage_write(lds, kreg[0], vreg[0]);
    prefetch(2, kreg[0], vreg[0]);
    __syncthreads();
    // two tiles per trip (static register slots); a tile past the end is fully masked and adds nothing
    for (int t0 = 0; t0 < ntiles; t0 += 2) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int t = t0 + u;
            const uint4* st = lds + u * STAGE;
            // ---- S^T tile: 32 keys x 32 queries
            af32x16 sacc;
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
#pragma unroll
            for (int db = 0; db < NDB; ++db) {
                abf16x8 kf[3];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) kf[pl] = __builtin_bit_cast(abf16x8, st[((db * 3 + pl) * 32 + fi) * 2 + fh]);
                ax6_mfma6(sacc, kf, qp[db]);
            }
            // tile t+1 -> the other stage (its readers finished before the last barrier), then refill the slot with tile t+3
            stage_write(lds + (u ^ 1) * STAGE, kreg[u ^ 1], vreg[u ^ 1]);
            prefetch(t + 3, kreg[u ^ 1], vreg[u ^ 1]);
            // sacc[r] = S[key k0 + (r&3) + 8*(r>>2) + 4*fh][query fi]
            const int k0 = t * 32;
            if (bias) {
#pragma unroll
                for (int rq = 0; rq < 4; ++rq)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        sacc[4 * rq + e] += bias[min(k0 + 8 * rq + 4 * fh + e, p.Nk - 1)];
            }
            if (k0 + 32 > p.Nk) {                  // ragged last tile / dead tile (wave-uniform)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (k0 + (r & 3) + 8 * (r >> 2) + 4 * fh >= p.Nk) sacc[r] = -INFINITY;
            }
            float mx = sacc[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sacc[r]);
            {
                auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
                mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));       // the other 16 keys of this query
            }
            const float mn = fmaxf(m_run, mx);
            const float alpha = __expf(m_run - mn);
            m_run = mn;
            float rsum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                sacc[r] = __expf(sacc[r] - mn);
                rsum += sacc[r];
            }
            l_run = l_run * alpha + rsum;
#pragma unroll
            for (int c = 0; c < DT; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[c][r] *= alpha;
            // ---- O^T += V^T P^T: registers 8 kb .. 8 kb + 7 are the B operand of key block kb
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                abf16x8 pf[3];
                const float x[8] = {sacc[8 * kb], sacc[8 * kb + 1], sacc[8 * kb + 2], sacc[8 * kb + 3],
                                    sacc[8 * kb + 4], sacc[8 * kb + 5], sacc[8 * kb + 6], sacc[8 * kb + 7]};
                ax6_split8(x, pf);
#pragma unroll
                for (int c = 0; c < DT; ++c) {
                    abf16x8 vf[3];
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl)
                        vf[pl] = __builtin_bit_cast(abf16x8, st[KQ + (((kb * 3 + pl) * DT + c) * 32 + fi) * 2 + fh]);
                    ax6_mfma6(oacc[c], vf, pf);
                }
            }
            __syncthreads();
        }
    }

    // ---- finish: add the two lane halves' shares of l, normalise, store O[q][d] (4 consecutive d per register quad)
    {
        auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(l_run), __float_as_uint(l_run), false, false);
        l_run = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
    }
    const float inv = 1.0f / l_run;
    const int qr = q0 + fi;
    if (qr < p.Nq) {
        float* O = p.o + (size_t)b * p.bso + hoff + (size_t)qr * p.ldo;
#pragma unroll
        for (int c = 0; c < DT; ++c)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int d0 = c * 32 + 8 * rq + 4 * fh;
                if (d0 < D)
                    *reinterpret_cast<float4*>(O + d0) = make_float4(oacc[c][4 * rq] * inv, oacc[c][4 * rq + 1] * inv,
                                                                     oacc[c][4 * rq + 2] * inv, oacc[c][4 * rq + 3] * inv);
            }
    }
}

template <int D>
static int launch_ax6(const AttnParams& p, int B, hipStream_t s) {
    dim3 grid(aed_cdiv(p.Nq, 128), p.H, B);
    hipLaunchKernelGGL((attention_x6_kernel<D>), grid, dim3(256), 0, s, p);
    AED_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_attention_x6(const AttnParams& p, int B, int D, hipStream_t s) {
    if (((uintptr_t)p.q | (uintptr_t)p.k) % 16 != 0) return -1;          // float4 fragment loads
    switch (D) {
        case 32: return launch_ax6<32>(p, B, s);
        case 48: return launch_ax6<48>(p, B, s);
        case 64: return launch_ax6<64>(p, B, s);
        default: return -1;
    }
}
