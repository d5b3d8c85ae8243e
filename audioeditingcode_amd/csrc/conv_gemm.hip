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
// Tiling: 256 threads = 4 waves; block tile BM x BN, K-chunk 32; operands staged through LDS with
// a 4-float row pad (conflict-free ds_read_b128 fragment reads); the next K-chunk is prefetched
// into registers while the current one feeds the MFMAs.  Each lane reads 4 consecutive k per
// ds_read_b128; MFMA step s of a k-block uses k = {s, 4+s} (A and B agree, the sum over k is
// order-free).
#include "aed_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct CGParams {
    const float* A;
    const float* W;
    const float* bias;
    float* C;
    const float* res;
    const float* rowvec;
    float* ws;
    int M, N, K;
    int lda, ldc, ldr, ld_rv;
    int IH, IW, OH, OW, Cin;
    int KH, KW, stride, pad_h, pad_w, dil_h, dil_w, up;
    int a_bs;
    int o_mul, o_add, o_len, out_bs;
    int in_act, out_act, accumulate, ksplit;
    int rpb;                 // output rows per batch item = OH*OW
    int nchunks;             // ceil(K/32)
    float in_slope, out_p, out_div;
};

#define BK 32
#define LDS_LD (BK + 4)

__device__ __forceinline__ float in_transform(float v, int act, float slope) {
    if (act == AED_ACT_SILU) return v / (1.0f + __expf(-v));
    if (act == AED_ACT_LEAKY) return v > 0.0f ? v : v * slope;
    return v;
}

__device__ __forceinline__ void store_out(const CGParams& p, int m, int n, float v) {
    // m < M, n < N guaranteed by caller
    int b = m / p.rpb;
    int q = m - b * p.rpb;
    int o = q * p.o_mul + p.o_add;
    if ((unsigned)o >= (unsigned)p.o_len) return;
    size_t row = (size_t)b * p.out_bs + o;
    if (p.bias) v += p.bias[n];
    if (p.rowvec) v += p.rowvec[(size_t)b * p.ld_rv + n];
    if (p.res) v += p.res[row * p.ldr + n];
    v = aed_apply_act(v, p.out_act, p.out_p);
    float* dst = p.C + row * p.ldc + n;
    if (p.accumulate == 1) v += *dst;
    else if (p.accumulate == 2) v = (*dst + v) / p.out_div;
    *dst = v;
}

template <int BM, int BN, int WROWS, int WCOLS, bool GENERIC>
__global__ __launch_bounds__(256) void conv_gemm_kernel(CGParams p) {
    constexpr int WM = BM / WROWS, WN = BN / WCOLS;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int PA = BM / 32, PB = BN / 32;
    static_assert(WROWS * WCOLS == 4, "4 waves");
    static_assert(TM >= 1 && TN >= 1, "wave tile");

    __shared__ __attribute__((aligned(16))) float lds[(BM + BN) * LDS_LD];
    float* As = lds;
    float* Ws = lds + BM * LDS_LD;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave / WCOLS, wc = wave % WCOLS;
    const int m0 = blockIdx.y * BM;
    const int n0 = blockIdx.x * BN;

    // K-chunk range of this split
    int kc_begin = 0, kc_end = p.nchunks;
    if (p.ksplit > 1) {
        int per = (p.nchunks + p.ksplit - 1) / p.ksplit;
        kc_begin = blockIdx.z * per;
        kc_end = min(p.nchunks, kc_begin + per);
    }

    // loader coordinates: 8 threads x float4 per 32-float row, 32 rows per pass
    const int lrow = tid >> 3;
    const int lcol = (tid & 7) * 4;

    int ay0[PA], ax0[PA];
    size_t abase[PA];
    bool avalid[PA];
#pragma unroll
    for (int q = 0; q < PA; ++q) {
        int m = m0 + lrow + 32 * q;
        avalid[q] = m < p.M;
        int mm = avalid[q] ? m : 0;
        int b = mm / p.rpb;
        int r = mm - b * p.rpb;
        int oy = r / p.OW;
        int ox = r - oy * p.OW;
        ay0[q] = oy * p.stride - p.pad_h;
        ax0[q] = ox * p.stride - p.pad_w;
        abase[q] = (size_t)b * (size_t)p.a_bs;
    }
    const int vIH = p.IH << p.up, vIW = p.IW << p.up;

    float4 ra[PA], rb[PB];

    auto prefetch = [&](int kc) {
        const int k0 = kc * BK;
        if constexpr (!GENERIC) {
            const int tap = k0 / p.Cin;
            const int c0 = k0 - tap * p.Cin;
            const int ty = tap / p.KW;
            const int tx = tap - ty * p.KW;
            const int dy = ty * p.dil_h, dx = tx * p.dil_w;
#pragma unroll
            for (int q = 0; q < PA; ++q) {
                int iy = ay0[q] + dy, ix = ax0[q] + dx;
                bool ok = avalid[q] && (unsigned)iy < (unsigned)vIH && (unsigned)ix < (unsigned)vIW;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ok) {
                    const float* src = p.A + abase[q] +
                                       ((size_t)(iy >> p.up) * p.IW + (size_t)(ix >> p.up)) * p.lda + c0 + lcol;
                    v = *reinterpret_cast<const float4*>(src);
                    if (p.in_act) {
                        v.x = in_transform(v.x, p.in_act, p.in_slope);
                        v.y = in_transform(v.y, p.in_act, p.in_slope);
                        v.z = in_transform(v.z, p.in_act, p.in_slope);
                        v.w = in_transform(v.w, p.in_act, p.in_slope);
                    }
                }
                ra[q] = v;
            }
#pragma unroll
            for (int q = 0; q < PB; ++q) {
                int n = n0 + lrow + 32 * q;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (n < p.N) v = *reinterpret_cast<const float4*>(p.W + (size_t)n * p.K + k0 + lcol);
                rb[q] = v;
            }
        } else {
#pragma unroll
            for (int q = 0; q < PA; ++q) {
                float tmp[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    int k = k0 + lcol + e;
                    float v = 0.f;
                    if (k < p.K && avalid[q]) {
                        int tap = k / p.Cin;
                        int c = k - tap * p.Cin;
                        int ty = tap / p.KW;
                        int tx = tap - ty * p.KW;
                        int iy = ay0[q] + ty * p.dil_h, ix = ax0[q] + tx * p.dil_w;
                        if ((unsigned)iy < (unsigned)vIH && (unsigned)ix < (unsigned)vIW) {
                            v = p.A[abase[q] + ((size_t)(iy >> p.up) * p.IW + (size_t)(ix >> p.up)) * p.lda + c];
                            v = in_transform(v, p.in_act, p.in_slope);
                        }
                    }
                    tmp[e] = v;
                }
                ra[q] = make_float4(tmp[0], tmp[1], tmp[2], tmp[3]);
            }
#pragma unroll
            for (int q = 0; q < PB; ++q) {
                int n = n0 + lrow + 32 * q;
                float tmp[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    int k = k0 + lcol + e;
                    tmp[e] = (n < p.N && k < p.K) ? p.W[(size_t)n * p.K + k] : 0.f;
                }
                rb[q] = make_float4(tmp[0], tmp[1], tmp[2], tmp[3]);
            }
        }
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
    const float* a_frag = As + (wr * WM + fi) * LDS_LD + 4 * fh;
    const float* b_frag = Ws + (wc * WN + fi) * LDS_LD + 4 * fh;

    if (kc_begin < kc_end) prefetch(kc_begin);
    for (int kc = kc_begin; kc < kc_end; ++kc) {
#pragma unroll
        for (int q = 0; q < PA; ++q)
            *reinterpret_cast<float4*>(As + (lrow + 32 * q) * LDS_LD + lcol) = ra[q];
#pragma unroll
        for (int q = 0; q < PB; ++q)
            *reinterpret_cast<float4*>(Ws + (lrow + 32 * q) * LDS_LD + lcol) = rb[q];
        __syncthreads();
        if (kc + 1 < kc_end) prefetch(kc + 1);
#pragma unroll
        for (int kb = 0; kb < BK / 8; ++kb) {
            float4 af[TM], bf[TN];
#pragma unroll
            for (int a = 0; a < TM; ++a)
                af[a] = *reinterpret_cast<const float4*>(a_frag + a * 32 * LDS_LD + kb * 8);
#pragma unroll
            for (int b = 0; b < TN; ++b)
                bf[b] = *reinterpret_cast<const float4*>(b_frag + b * 32 * LDS_LD + kb * 8);
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b) {
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a].x, bf[b].x, acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a].y, bf[b].y, acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a].z, bf[b].z, acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a].w, bf[b].w, acc[a][b], 0, 0, 0);
                }
        }
        __syncthreads();
    }

    // epilogue: acc[a][b][r] is C[row = (r&3) + 8*(r>>2) + 4*fh][col = fi] of the 32x32 tile
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const int n = n0 + wc * WN + b * 32 + fi;
            if (n >= p.N) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wr * WM + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                if (m >= p.M) continue;
                if (p.ksplit > 1)
                    p.ws[((size_t)blockIdx.z * p.M + m) * p.N + n] = acc[a][b][r];
                else
                    store_out(p, m, n, acc[a][b][r]);
            }
        }
}

__global__ __launch_bounds__(256) void splitk_reduce_kernel(CGParams p) {
    size_t total = (size_t)p.M * p.N;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        int m = (int)(e / p.N);
        int n = (int)(e - (size_t)m * p.N);
        float v = 0.f;
        for (int z = 0; z < p.ksplit; ++z) v += p.ws[(size_t)z * total + e];
        store_out(p, m, n, v);
    }
}

static int fill_params(const aed_op* op, CGParams& p) {
    p.A = (const float*)op->p[0];
    p.W = (const float*)op->p[1];
    p.bias = (const float*)op->p[2];
    p.C = (float*)op->p[3];
    p.res = (const float*)op->p[4];
    p.rowvec = (const float*)op->p[5];
    p.ws = (float*)op->p[6];
    const int32_t* i = op->i;
    p.M = i[0]; p.N = i[1]; p.K = i[2]; p.lda = i[3]; p.ldc = i[4]; p.ldr = i[5]; p.ld_rv = i[6];
    p.IH = i[7]; p.IW = i[8]; p.OH = i[9]; p.OW = i[10]; p.Cin = i[11]; p.KH = i[12]; p.KW = i[13];
    p.stride = i[14]; p.pad_h = i[15]; p.pad_w = i[16]; p.dil_h = i[17]; p.dil_w = i[18]; p.up = i[19];
    p.a_bs = i[20]; p.o_mul = i[21]; p.o_add = i[22]; p.o_len = i[23]; p.out_bs = i[24];
    p.in_act = i[25]; p.out_act = i[26]; p.accumulate = i[27]; p.ksplit = i[28];
    p.in_slope = op->f[0]; p.out_p = op->f[1]; p.out_div = op->f[2];
    p.rpb = p.OH * p.OW;
    p.nchunks = (p.K + BK - 1) / BK;
    AED_REQUIRE(p.A && p.W && p.C, "conv_gemm: null operand");
    AED_REQUIRE(p.M > 0 && p.N > 0 && p.K > 0, "conv_gemm: bad shape M=%d N=%d K=%d", p.M, p.N, p.K);
    AED_REQUIRE(p.K == p.KH * p.KW * p.Cin, "conv_gemm: K=%d != KH*KW*Cin=%d", p.K, p.KH * p.KW * p.Cin);
    AED_REQUIRE(p.rpb > 0 && p.M % p.rpb == 0, "conv_gemm: M=%d not a multiple of OH*OW=%d", p.M, p.rpb);
    if (p.ksplit < 1) p.ksplit = 1;
    if (p.ksplit > 1) AED_REQUIRE(p.ws != nullptr, "conv_gemm: split-K needs a workspace");
    return 0;
}

template <int BM, int BN, int WR, int WC>
static void launch_cfg(const CGParams& p, bool generic, hipStream_t s) {
    dim3 grid(aed_cdiv(p.N, BN), aed_cdiv(p.M, BM), p.ksplit);
    if (generic)
        hipLaunchKernelGGL((conv_gemm_kernel<BM, BN, WR, WC, true>), grid, dim3(256), 0, s, p);
    else
        hipLaunchKernelGGL((conv_gemm_kernel<BM, BN, WR, WC, false>), grid, dim3(256), 0, s, p);
}

// tile_cfg: 0 auto, 1 = 128x128, 2 = 128x64, 3 = 64x128, 4 = 64x64, 5 = 128x32, 6 = 32x128
int launch_conv_gemm(const aed_op* op, hipStream_t s) {
    CGParams p;
    int rc = fill_params(op, p);
    if (rc) return rc;
    const bool generic = (p.Cin % BK != 0) || (p.lda % 4 != 0) || ((uintptr_t)p.A % 16 != 0) ||
                         ((uintptr_t)p.W % 16 != 0);
    int cfg = op->i[29];
    if (cfg == 0) {
        const int cus = aed_num_cus();
        auto blocks = [&](int bm, int bn) { return (long)aed_cdiv(p.M, bm) * aed_cdiv(p.N, bn) * p.ksplit; };
        const int n_small = p.N <= 32;
        if (n_small) cfg = 5;
        else if (p.M <= 32) cfg = 6;
        else if (blocks(128, 128) >= 2L * cus && p.N >= 128) cfg = 1;
        else if (blocks(128, 64) >= 2L * cus && p.N >= 64) cfg = 2;
        else cfg = 4;
        if (p.N < 64 && cfg != 5) cfg = 5;
    }
    switch (cfg) {
        case 1: launch_cfg<128, 128, 2, 2>(p, generic, s); break;
        case 2: launch_cfg<128, 64, 2, 2>(p, generic, s); break;
        case 3: launch_cfg<64, 128, 2, 2>(p, generic, s); break;
        case 4: launch_cfg<64, 64, 2, 2>(p, generic, s); break;
        case 5: launch_cfg<128, 32, 4, 1>(p, generic, s); break;
        case 6: launch_cfg<32, 128, 1, 4>(p, generic, s); break;
        default: AED_REQUIRE(false, "conv_gemm: bad tile cfg %d", cfg);
    }
    AED_CHECK_HIP(hipGetLastError());
    if (p.ksplit > 1) {
        size_t total = (size_t)p.M * p.N;
        int grid = (int)((total + 255) / 256);
        if (grid > 2048) grid = 2048;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(grid), dim3(256), 0, s, p);
        AED_CHECK_HIP(hipGetLastError());
    }
    return 0;
}

int launch_splitk_reduce(const aed_op* op, hipStream_t s) {
    CGParams p;
    int rc = fill_params(op, p);
    if (rc) return rc;
    size_t total = (size_t)p.M * p.N;
    int grid = (int)((total + 255) / 256);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(grid), dim3(256), 0, s, p);
    AED_CHECK_HIP(hipGetLastError());
    return 0;
}
