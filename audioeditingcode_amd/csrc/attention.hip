// attention.hip -- softmax(Q K^T * scale + key_bias) V, flash-style, on fp32 MFMA.
//
// Shapes on the path are small and odd: N_q in {1024,256,64}, 8 heads,
// d_head in {32,48,80} (AudioLDM2) / 64 (TANGO), N_k = N_q (self) or 8 / L<=~64 (cross, additive
// -10000 key mask, models.py:740-755).  So the kernel is tiled for occupancy, not for long
// sequences: a 256-thread workgroup owns 64 query rows of one (batch, head); each of its 4
// wavefronts owns 16 rows and runs v_mfma_f32_16x16x4_f32 for both QK^T and PV.  K tiles sit in
// LDS row-major, V tiles transposed (so both MFMA B-fragments are one ds_read_b128), P makes the
// accumulator -> A-operand layout change through a wave-private LDS slab.  Online softmax keeps
// (max, sum) per row in registers; nothing of size N_q x N_k ever reaches HBM.
#include "aed_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct AttnParams {
    const float* q; const float* k; const float* v; const float* bias; float* o;
    int Nq, Nk, H;
    int ldq, ldk, ldv, ldo, ld_bias;
    long bsq, bsk, bsv, bso;
    float scale;
};

#define KV_TILE 64
#define Q_TILE 64

template <int D>
__global__ __launch_bounds__(256) void attention_kernel(AttnParams p) {
    constexpr int KLD = D + 4;
    constexpr int VLD = KV_TILE + 4;
    constexpr int NB = D / 16;
    __shared__ __attribute__((aligned(16))) float Ks[KV_TILE * KLD];
    __shared__ __attribute__((aligned(16))) float Vt[D * VLD];
    __shared__ __attribute__((aligned(16))) float Ps[4 * 16 * VLD];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int fi = lane & 15, fh = lane >> 4;
    const int b = blockIdx.z, head = blockIdx.y;
    const int q0 = blockIdx.x * Q_TILE + wave * 16;
    const int hoff = head * D;

    const float* Q = p.q + (size_t)b * p.bsq + hoff;
    const float* K = p.k + (size_t)b * p.bsk + hoff;
    const float* V = p.v + (size_t)b * p.bsv + hoff;
    const float* bias = p.bias ? p.bias + (size_t)b * p.ld_bias : nullptr;

    float4 qf[NB];
    {
        const int qr = q0 + fi;
#pragma unroll
        for (int blk = 0; blk < NB; ++blk) {
            qf[blk] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (qr < p.Nq) qf[blk] = *reinterpret_cast<const float4*>(Q + (size_t)qr * p.ldq + 16 * blk + 4 * fh);
        }
    }
    f32x4 oacc[NB];
#pragma unroll
    for (int c = 0; c < NB; ++c) oacc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float m_run[4], l_run[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { m_run[r] = -INFINITY; l_run[r] = 0.f; }

    float* Pw = Ps + wave * 16 * VLD;

    for (int k0 = 0; k0 < p.Nk; k0 += KV_TILE) {
        // ---- stage K (row-major) and V (transposed) tiles
        for (int idx = tid; idx < KV_TILE * (D / 4); idx += 256) {
            const int key = idx / (D / 4), c4 = idx - key * (D / 4);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (k0 + key < p.Nk) v = *reinterpret_cast<const float4*>(K + (size_t)(k0 + key) * p.ldk + 4 * c4);
            *reinterpret_cast<float4*>(Ks + key * KLD + 4 * c4) = v;
        }
        {
            const int key = tid & 63, dg = tid >> 6;
            for (int dd = dg * 4; dd < D; dd += 16) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (k0 + key < p.Nk) v = *reinterpret_cast<const float4*>(V + (size_t)(k0 + key) * p.ldv + dd);
                Vt[(dd + 0) * VLD + key] = v.x;
                Vt[(dd + 1) * VLD + key] = v.y;
                Vt[(dd + 2) * VLD + key] = v.z;
                Vt[(dd + 3) * VLD + key] = v.w;
            }
        }
        __syncthreads();

        // ---- S = Q K^T  (16 rows x 64 keys per wave)
        f32x4 sacc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) sacc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int blk = 0; blk < NB; ++blk) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 kf = *reinterpret_cast<const float4*>(Ks + (16 * j + fi) * KLD + 16 * blk + 4 * fh);
                sacc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(qf[blk].x, kf.x, sacc[j], 0, 0, 0);
                sacc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(qf[blk].y, kf.y, sacc[j], 0, 0, 0);
                sacc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(qf[blk].z, kf.z, sacc[j], 0, 0, 0);
                sacc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(qf[blk].w, kf.w, sacc[j], 0, 0, 0);
            }
        }
        // sacc[j][r] = S[row 4*fh + r][key 16*j + fi]
        float tmax[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) tmax[r] = -INFINITY;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int key = k0 + 16 * j + fi;
            const bool ok = key < p.Nk;
            const float bv = (ok && bias) ? bias[key] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float s = ok ? sacc[j][r] * p.scale + bv : -INFINITY;
                sacc[j][r] = s;
                tmax[r] = fmaxf(tmax[r], s);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) tmax[r] = fmaxf(tmax[r], __shfl_xor(tmax[r], o, 16));
        }
        float alpha[4], rsum[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float mn = fmaxf(m_run[r], tmax[r]);
            alpha[r] = __expf(m_run[r] - mn);
            m_run[r] = mn;
            rsum[r] = 0.f;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pv = __expf(sacc[j][r] - m_run[r]);
                rsum[r] += pv;
                Pw[(4 * fh + r) * VLD + 16 * j + fi] = pv;
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) rsum[r] += __shfl_xor(rsum[r], o, 16);
            l_run[r] = l_run[r] * alpha[r] + rsum[r];
        }
#pragma unroll
        for (int c = 0; c < NB; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) oacc[c][r] *= alpha[r];

        // P slab is wave-private: LDS ops of one wave complete in issue order; only keep the
        // compiler from reordering the reads above the writes.
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();

        // ---- O += P V
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            const float4 pf = *reinterpret_cast<const float4*>(Pw + fi * VLD + 16 * kb + 4 * fh);
#pragma unroll
            for (int c = 0; c < NB; ++c) {
                const float4 vf = *reinterpret_cast<const float4*>(Vt + (16 * c + fi) * VLD + 16 * kb + 4 * fh);
                oacc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(pf.x, vf.x, oacc[c], 0, 0, 0);
                oacc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(pf.y, vf.y, oacc[c], 0, 0, 0);
                oacc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(pf.z, vf.z, oacc[c], 0, 0, 0);
                oacc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(pf.w, vf.w, oacc[c], 0, 0, 0);
            }
        }
        __syncthreads();
    }

    float* O = p.o + (size_t)b * p.bso + hoff;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int qr = q0 + 4 * fh + r;
        if (qr >= p.Nq) continue;
        const float inv = 1.0f / l_run[r];
#pragma unroll
        for (int c = 0; c < NB; ++c) O[(size_t)qr * p.ldo + 16 * c + fi] = oacc[c][r] * inv;
    }
}

// ---------------------------------------------------------------------------------------------------
// Split-KV variant for self-attention at small batch (N_k > 64).  With U-Net batch 2 the kernel above puts
// ONE wavefront on each SIMD of the chip and every wave walks all N_k/64 key tiles serially (QK^T MFMAs ->
// softmax VALU -> PV MFMAs, nothing to overlap with).  Here a block owns 32 query rows and its 4 waves are
// (query half) x (key half): each wave visits every second key tile, so the dependent chain per wave is
// half as long and two waves share each SIMD (MFMA of one overlaps softmax of the other).  The two key
// halves keep independent online-softmax states (m, l, O) and are merged once at the end through LDS:
//   m = max(m0, m1);  O = O0*exp(m0-m) + O1*exp(m1-m);  l likewise.
template <int D>
__global__ __launch_bounds__(256) void attention_split_kernel(AttnParams p) {
    constexpr int KLD = D + 4;
    constexpr int VLD = KV_TILE + 4;
    constexpr int NB = D / 16;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ks = smem;                               // [2][64][KLD]
    float* Vt = Ks + 2 * KV_TILE * KLD;             // [2][D][VLD]
    float* Ps = Vt + 2 * D * VLD;                   // [4][16][VLD]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int qg = wave & 1, kh = wave >> 1;
    const int fi = lane & 15, fh = lane >> 4;
    const int b = blockIdx.z, head = blockIdx.y;
    const int q0 = blockIdx.x * 32 + qg * 16;
    const int hoff = head * D;

    const float* Q = p.q + (size_t)b * p.bsq + hoff;
    const float* K = p.k + (size_t)b * p.bsk + hoff;
    const float* V = p.v + (size_t)b * p.bsv + hoff;
    const float* bias = p.bias ? p.bias + (size_t)b * p.ld_bias : nullptr;

    float4 qf[NB];
    {
        const int qr = q0 + fi;
#pragma unroll
        for (int blk = 0; blk < NB; ++blk) {
            qf[blk] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (qr < p.Nq) qf[blk] = *reinterpret_cast<const float4*>(Q + (size_t)qr * p.ldq + 16 * blk + 4 * fh);
        }
    }
    f32x4 oacc[NB];
#pragma unroll
    for (int c = 0; c < NB; ++c) oacc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float m_run[4], l_run[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { m_run[r] = -INFINITY; l_run[r] = 0.f; }

    float* Pw = Ps + wave * 16 * VLD;
    const float* Kh = Ks + kh * KV_TILE * KLD;
    const float* Vh = Vt + kh * D * VLD;

    for (int k00 = 0; k00 < p.Nk; k00 += 2 * KV_TILE) {
        // ---- stage both halves' K (row-major) and V (transposed) tiles: keys k00 .. k00+127
        for (int idx = tid; idx < 2 * KV_TILE * (D / 4); idx += 256) {
            const int key = idx / (D / 4), c4 = idx - key * (D / 4);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (k00 + key < p.Nk) v = *reinterpret_cast<const float4*>(K + (size_t)(k00 + key) * p.ldk + 4 * c4);
            *reinterpret_cast<float4*>(Ks + key * KLD + 4 * c4) = v;         // half = key / 64, contiguous
        }
        {
            const int key = tid & 127, dg = tid >> 7;                        // 128 keys x 2 channel groups
            for (int dd = dg * 4; dd < D; dd += 8) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (k00 + key < p.Nk) v = *reinterpret_cast<const float4*>(V + (size_t)(k00 + key) * p.ldv + dd);
                float* dst = Vt + (key >> 6) * D * VLD + (key & 63);
                dst[(dd + 0) * VLD] = v.x;
                dst[(dd + 1) * VLD] = v.y;
                dst[(dd + 2) * VLD] = v.z;
                dst[(dd + 3) * VLD] = v.w;
            }
        }
        __syncthreads();
        const int k0 = k00 + kh * KV_TILE;
        if (k0 < p.Nk) {            // wave-uniform: this half has keys in this round
            f32x4 sacc[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) sacc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int blk = 0; blk < NB; ++blk) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float4 kf = *reinterpret_cast<const float4*>(Kh + (16 * j + fi) * KLD + 16 * blk + 4 * fh);
                    sacc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(qf[blk].x, kf.x, sacc[j], 0, 0, 0);
                    sacc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(qf[blk].y, kf.y, sacc[j], 0, 0, 0);
                    sacc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(qf[blk].z, kf.z, sacc[j], 0, 0, 0);
                    sacc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(qf[blk].w, kf.w, sacc[j], 0, 0, 0);
                }
            }
            float tmax[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) tmax[r] = -INFINITY;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int key = k0 + 16 * j + fi;
                const bool ok = key < p.Nk;
                const float bv = (ok && bias) ? bias[key] : 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float s = ok ? sacc[j][r] * p.scale + bv : -INFINITY;
                    sacc[j][r] = s;
                    tmax[r] = fmaxf(tmax[r], s);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int o = 8; o > 0; o >>= 1) tmax[r] = fmaxf(tmax[r], __shfl_xor(tmax[r], o, 16));
            }
            float alpha[4], rsum[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float mn = fmaxf(m_run[r], tmax[r]);
                alpha[r] = __expf(m_run[r] - mn);
                m_run[r] = mn;
                rsum[r] = 0.f;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pv = __expf(sacc[j][r] - m_run[r]);
                    rsum[r] += pv;
                    Pw[(4 * fh + r) * VLD + 16 * j + fi] = pv;
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int o = 8; o > 0; o >>= 1) rsum[r] += __shfl_xor(rsum[r], o, 16);
                l_run[r] = l_run[r] * alpha[r] + rsum[r];
            }
#pragma unroll
            for (int c = 0; c < NB; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) oacc[c][r] *= alpha[r];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                const float4 pf = *reinterpret_cast<const float4*>(Pw + fi * VLD + 16 * kb + 4 * fh);
#pragma unroll
                for (int c = 0; c < NB; ++c) {
                    const float4 vf = *reinterpret_cast<const float4*>(Vh + (16 * c + fi) * VLD + 16 * kb + 4 * fh);
                    oacc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(pf.x, vf.x, oacc[c], 0, 0, 0);
                    oacc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(pf.y, vf.y, oacc[c], 0, 0, 0);
                    oacc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(pf.z, vf.z, oacc[c], 0, 0, 0);
                    oacc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(pf.w, vf.w, oacc[c], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }

    // ---- merge the two key halves (same query rows, same lane layout) through LDS
    constexpr int MSZ = 8 + 4 * NB;                 // floats per lane: m[4], l[4], O[NB][4]
    float* mrg = smem + (size_t)qg * 64 * MSZ;      // operand tiles are dead after the last barrier
    if (kh == 1) {
        float* dst = mrg + lane * MSZ;
#pragma unroll
        for (int r = 0; r < 4; ++r) { dst[r] = m_run[r]; dst[4 + r] = l_run[r]; }
#pragma unroll
        for (int c = 0; c < NB; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) dst[8 + 4 * c + r] = oacc[c][r];
    }
    __syncthreads();
    if (kh == 0) {
        const float* src = mrg + lane * MSZ;
        float* O = p.o + (size_t)b * p.bso + hoff;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int qr = q0 + 4 * fh + r;
            const float m1 = src[r], l1 = src[4 + r];
            const float m = fmaxf(m_run[r], m1);
            const float a0 = __expf(m_run[r] - m), a1 = (m1 == -INFINITY) ? 0.f : __expf(m1 - m);
            const float inv = 1.0f / (l_run[r] * a0 + l1 * a1);
            if (qr >= p.Nq) continue;
#pragma unroll
            for (int c = 0; c < NB; ++c)
                O[(size_t)qr * p.ldo + 16 * c + fi] = (oacc[c][r] * a0 + src[8 + 4 * c + r] * a1) * inv;
        }
    }
}

template <int D>
static int launch_split(const AttnParams& p, int B, hipStream_t s) {
    constexpr int KLD = D + 4, VLD = KV_TILE + 4;
    const size_t bytes = sizeof(float) * (2 * KV_TILE * KLD + 2 * D * VLD + 4 * 16 * VLD);
    static bool attr_set = false;
    if (!attr_set) {
        AED_CHECK_HIP(hipFuncSetAttribute((const void*)attention_split_kernel<D>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        attr_set = true;
    }
    dim3 grid(aed_cdiv(p.Nq, 32), p.H, B);
    hipLaunchKernelGGL(attention_split_kernel<D>, grid, dim3(256), bytes, s, p);
    return 0;
}

// slots: p0=q p1=k p2=v p3=bias(or null) p4=out
//        i0=B i1=H i2=Nq i3=Nk i4=D i5=ldq i6=ldk i7=ldv i8=ldo i9=ld_bias
//        i10=bsq i11=bsk i12=bsv i13=bso (elements) ; f0=scale
int launch_attention(const aed_op* op, hipStream_t s) {
    const int32_t* i = op->i;
    AttnParams p;
    p.q = (const float*)op->p[0]; p.k = (const float*)op->p[1]; p.v = (const float*)op->p[2];
    p.bias = (const float*)op->p[3]; p.o = (float*)op->p[4];
    AED_REQUIRE(p.q && p.k && p.v && p.o, "attention: null pointer");
    p.H = i[1]; p.Nq = i[2]; p.Nk = i[3];
    p.ldq = i[5]; p.ldk = i[6]; p.ldv = i[7]; p.ldo = i[8]; p.ld_bias = i[9];
    p.bsq = i[10]; p.bsk = i[11]; p.bsv = i[12]; p.bso = i[13];
    p.scale = op->f[0];
    AED_REQUIRE(p.ldq % 4 == 0 && p.ldk % 4 == 0 && p.ldv % 4 == 0, "attention: row strides must be multiples of 4");
    AED_REQUIRE(p.Nq > 0 && p.Nk > 0, "attention: empty sequence");
    if (p.Nk > KV_TILE && i[14] != 1) {        // split-KV variant (i14 = 1 forces the single-pass kernel)
        int rc = 0;
        switch (i[4]) {
            case 16: rc = launch_split<16>(p, i[0], s); break;
            case 32: rc = launch_split<32>(p, i[0], s); break;
            case 48: rc = launch_split<48>(p, i[0], s); break;
            case 64: rc = launch_split<64>(p, i[0], s); break;
            case 80: rc = launch_split<80>(p, i[0], s); break;
            default: AED_REQUIRE(false, "attention: unsupported head dim %d (16/32/48/64/80)", i[4]);
        }
        if (rc) return rc;
        AED_CHECK_HIP(hipGetLastError());
        return 0;
    }
    dim3 grid(aed_cdiv(p.Nq, Q_TILE), p.H, i[0]);
    switch (i[4]) {
        case 16: hipLaunchKernelGGL(attention_kernel<16>, grid, dim3(256), 0, s, p); break;
        case 32: hipLaunchKernelGGL(attention_kernel<32>, grid, dim3(256), 0, s, p); break;
        case 48: hipLaunchKernelGGL(attention_kernel<48>, grid, dim3(256), 0, s, p); break;
        case 64: hipLaunchKernelGGL(attention_kernel<64>, grid, dim3(256), 0, s, p); break;
        case 80: hipLaunchKernelGGL(attention_kernel<80>, grid, dim3(256), 0, s, p); break;
        default: AED_REQUIRE(false, "attention: unsupported head dim %d (16/32/48/64/80)", i[4]);
    }
    AED_CHECK_HIP(hipGetLastError());
    return 0;
}
