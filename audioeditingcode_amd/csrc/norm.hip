// norm.hip -- GroupNorm (+SiLU) on channels-last activations.  (LayerNorm, K8, has no kernel of its own any
// more: its statistics are gathered inside the consuming GEMM, conv_gemm.hip / lin_gemm.hip ln_mode.)
//
// GroupNorm over [B, HW, C] with G groups of C/G contiguous channels.  HBM-bound: the activation
// is read twice and written once.  Two launches:
//   gn_stats : grid (chunks, B); every block streams a slab of rows fully coalesced (float4 per
//              lane) and emits per-group partial (sum, sumsq) -- deterministic tree, no atomics;
//   gn_apply : every block re-reduces the <= few-hundred partials of its batch item in fp64,
//              then normalises + affine (+SiLU) its slab.
#include "aed_common.h"

// Two-source rows: channel c < C1 of row `row` lives in x (stride ldx), c >= C1 in x2 (stride ldx2) at c - C1 -- the
// (h | skip) concat of an up-block resnet (models.py:349-357 torch.cat) is never materialised.  c is a multiple of 4.
__device__ __forceinline__ const float* gn_src(const float* x, const float* x2, int C1, int ldx, int ldx2, size_t row,
                                               int c) {
    return (x2 != nullptr && c >= C1) ? x2 + row * ldx2 + (c - C1) : x + row * ldx + c;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gn_stats_kernel(const float* __restrict__ x, float* __restrict__ part,
                                                        int HW, int C, int G, int ldx, int rpc, int nchunks,
                                                        const float* __restrict__ x2, int C1, int ldx2) {
    __shared__ float sh[256][2];
    __shared__ float gacc[64][2];
    const int tid = threadIdx.x;
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int row0 = chunk * rpc;
    const int row1 = min(HW, row0 + rpc);
    const int Q = C >> 2;
    const int cpg4 = (C / G) >> 2;
    if (tid < G) { gacc[tid][0] = 0.f; gacc[tid][1] = 0.f; }
    __syncthreads();
    for (int cbase = 0; cbase < Q; cbase += 256) {
        const int ncol = min(256, Q - cbase);
        const int rpi = 256 / ncol;
        const int col = tid % ncol, rsub = tid / ncol;
        float s = 0.f, ss = 0.f;
        if (rsub < rpi) {
            for (int r = row0 + rsub; r < row1; r += rpi) {
                const float4 v = *reinterpret_cast<const float4*>(
                    gn_src(x, x2, C1, ldx, ldx2, (size_t)b * HW + r, 4 * (cbase + col)));
                s += (v.x + v.y) + (v.z + v.w);
                ss += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
            }
        }
        sh[tid][0] = s;
        sh[tid][1] = ss;
        __syncthreads();
        if (tid < G) {
            int c_lo = max(tid * cpg4, cbase), c_hi = min((tid + 1) * cpg4, cbase + ncol);
            float a0 = 0.f, a1 = 0.f;
            for (int rs = 0; rs < rpi; ++rs)
                for (int c = c_lo; c < c_hi; ++c) {
                    a0 += sh[rs * ncol + (c - cbase)][0];
                    a1 += sh[rs * ncol + (c - cbase)][1];
                }
            gacc[tid][0] += a0;
            gacc[tid][1] += a1;
        }
        __syncthreads();
    }
    if (tid < G) {
        float* dst = part + (((size_t)b * nchunks + chunk) * G + tid) * 2;
        dst[0] = gacc[tid][0];
        dst[1] = gacc[tid][1];
    }
}

__global__ __launch_bounds__(256) void gn_apply_kernel(const float* __restrict__ x, const float* __restrict__ part,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float* __restrict__ y,
                                                        int HW, int C, int G, int ldx, int ldy, int rpc, int nchunks,
                                                        float eps, int act, const float* __restrict__ x2, int C1,
                                                        int ldx2) {
    // rpc here is the APPLY slab height; nchunks the number of STATS partials per batch item.
    __shared__ double red_s[256], red_ss[256];
    __shared__ float mean_s[64], rstd_s[64];
    const int tid = threadIdx.x;
    const int b = blockIdx.y, chunk = blockIdx.x;
    {   // all 256 threads reduce the partials: thread -> (group g, lane j of 256/G), strided over chunks
        const int per = 256 / G;            // G in {8,16,32,64}
        const int g = tid % G, j = tid / G;
        double s = 0.0, ss = 0.0;
        if (j < per) {
            const float* src = part + ((size_t)b * nchunks * G + g) * 2;
            for (int c = j; c < nchunks; c += per) {
                s += (double)src[(size_t)c * G * 2];
                ss += (double)src[(size_t)c * G * 2 + 1];
            }
        }
        red_s[tid] = s;
        red_ss[tid] = ss;
        __syncthreads();
        if (tid < G) {
            for (int k = 1; k < per; ++k) { s += red_s[tid + k * G]; ss += red_ss[tid + k * G]; }
            const double n = (double)HW * (double)(C / G);
            const double mean = s / n;
            double var = ss / n - mean * mean;
            if (var < 0.0) var = 0.0;
            mean_s[tid] = (float)mean;
            rstd_s[tid] = (float)(1.0 / sqrt(var + (double)eps));
        }
        __syncthreads();
    }
    const int row0 = chunk * rpc;
    const int row1 = min(HW, row0 + rpc);
    const int Q = C >> 2;
    const int cpg4 = (C / G) >> 2;
    const int total = (row1 - row0) * Q;
    for (int e = tid; e < total; e += 256) {
        const int r = e / Q;
        const int c4 = e - r * Q;
        const int g = c4 / cpg4;
        const size_t row = (size_t)b * HW + row0 + r;
        float4 v = *reinterpret_cast<const float4*>(gn_src(x, x2, C1, ldx, ldx2, row, 4 * c4));
        const float4 ga = *reinterpret_cast<const float4*>(gamma + 4 * c4);
        const float4 be = *reinterpret_cast<const float4*>(beta + 4 * c4);
        const float m = mean_s[g], rs = rstd_s[g];
        v.x = (v.x - m) * rs * ga.x + be.x;
        v.y = (v.y - m) * rs * ga.y + be.y;
        v.z = (v.z - m) * rs * ga.z + be.z;
        v.w = (v.w - m) * rs * ga.w + be.w;
        if (act == AED_ACT_SILU) {
            v.x = v.x / (1.0f + expf(-v.x));
            v.y = v.y / (1.0f + expf(-v.y));
            v.z = v.z / (1.0f + expf(-v.z));
            v.w = v.w / (1.0f + expf(-v.w));
        }
        *reinterpret_cast<float4*>(y + row * ldy + 4 * c4) = v;
    }
}

// GroupNorm statistics folded into per-channel (scale, shift) vectors for the FUSED path: the consumer
// conv applies x' = act(x*a + d) inside its A-loader (conv_gemm.hip), so the normalised activation never
// exists in HBM and GroupNorm costs one small launch instead of two + a round trip.
//   a[b][c] = rstd[b,g(c)] * gamma[c]        d[b][c] = beta[c] - mean[b,g(c)] * a[b][c]
// grid (G, B): one block per (group, batch item); used for the U-Net's small feature maps.
__global__ __launch_bounds__(256) void gn_scale_shift_kernel(const float* __restrict__ x,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float* __restrict__ ab,
                                                              int HW, int C, int G, int ldx, float eps) {
    __shared__ double rs[4], rss[4];
    const int tid = threadIdx.x, g = blockIdx.x, b = blockIdx.y;
    const int cpg = C / G, cpg4 = cpg >> 2;
    const float* xb = x + (size_t)b * HW * ldx + g * cpg;
    const int total = HW * cpg4;
    float s = 0.f, ss = 0.f;
    for (int e = tid; e < total; e += 256) {
        const int row = e / cpg4, j = e - row * cpg4;
        const float4 v = *reinterpret_cast<const float4*>(xb + (size_t)row * ldx + 4 * j);
        s += (v.x + v.y) + (v.z + v.w);
        ss += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
    double ds = (double)s, dss = (double)ss;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { ds += __shfl_xor(ds, o, 64); dss += __shfl_xor(dss, o, 64); }
    if ((tid & 63) == 0) { rs[tid >> 6] = ds; rss[tid >> 6] = dss; }
    __syncthreads();
    ds = (rs[0] + rs[1]) + (rs[2] + rs[3]);
    dss = (rss[0] + rss[1]) + (rss[2] + rss[3]);
    const double n = (double)HW * (double)cpg;
    const double mean = ds / n;
    double var = dss / n - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    if (tid < cpg) {
        const int c = g * cpg + tid;
        const float a = rstd * gamma[c];
        ab[(size_t)b * 2 * C + c] = a;
        ab[(size_t)b * 2 * C + C + c] = beta[c] - (float)mean * a;
    }
}
// Single-launch GroupNorm(+SiLU) for the U-Net's small feature maps: one block per (group, batch item)
// computes the statistics of its slice and normalises it in the same launch (the slice -- at most a few
// hundred KB -- is re-read from L1/L2).  At U-Net batch 2 a GroupNorm is latency-bound, so one launch
// instead of two is what matters; large maps (VAE) keep the two-pass streaming kernels above.
// U loads are issued back to back (clamped addresses, masked use) before anything is consumed: round 1's first version
// waited on every load where it was issued (run-time-bounded loop, one load per trip); measured in round 2 this one is
// ~10 % faster per launch at U-Net batch 2.
template <int U>
__global__ __launch_bounds__(256) void gn_small_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float* __restrict__ y, int HW,
                                                         int C, int G, int ldx, int ldy, float eps, int act,
                                                         const float* __restrict__ x2, int C1, int ldx2) {
    __shared__ double rs[4], rss[4];
    const int tid = threadIdx.x, g = blockIdx.x, b = blockIdx.y;
    const int cpg = C / G, cpg4 = cpg >> 2;
    const size_t rb = (size_t)b * HW;
    float* yb = y + (size_t)b * HW * ldy + g * cpg;
    const int total = HW * cpg4;
    float s = 0.f, ss = 0.f;
    for (int e0 = tid; e0 < total; e0 += 256 * U) {
        float4 v[U];
        const float* src[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = min(e0 + 256 * u, total - 1);
            const int row = e / cpg4, j = e - row * cpg4;
            src[u] = gn_src(x, x2, C1, ldx, ldx2, rb + row, g * cpg + 4 * j);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = *reinterpret_cast<const float4*>(src[u]);
        __builtin_amdgcn_sched_barrier(0);          // keep the U loads together ahead of their first use
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (e0 + 256 * u < total) {
                s += (v[u].x + v[u].y) + (v[u].z + v[u].w);
                ss += (v[u].x * v[u].x + v[u].y * v[u].y) + (v[u].z * v[u].z + v[u].w * v[u].w);
            }
        }
    }
    double ds = (double)s, dss = (double)ss;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { ds += __shfl_xor(ds, o, 64); dss += __shfl_xor(dss, o, 64); }
    if ((tid & 63) == 0) { rs[tid >> 6] = ds; rss[tid >> 6] = dss; }
    __syncthreads();
    ds = (rs[0] + rs[1]) + (rs[2] + rs[3]);
    dss = (rss[0] + rss[1]) + (rss[2] + rss[3]);
    const double n = (double)HW * (double)cpg;
    const double dmean = ds / n;
    double var = dss / n - dmean * dmean;
    if (var < 0.0) var = 0.0;
    const float mean = (float)dmean;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    for (int e0 = tid; e0 < total; e0 += 256 * U) {
        float4 v[U], ga[U], be[U];
        int rowv[U], jv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = min(e0 + 256 * u, total - 1);
            rowv[u] = e / cpg4;
            jv[u] = e - rowv[u] * cpg4;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            v[u] = *reinterpret_cast<const float4*>(gn_src(x, x2, C1, ldx, ldx2, rb + rowv[u], g * cpg + 4 * jv[u]));
            ga[u] = *reinterpret_cast<const float4*>(gamma + g * cpg + 4 * jv[u]);
            be[u] = *reinterpret_cast<const float4*>(beta + g * cpg + 4 * jv[u]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (e0 + 256 * u >= total) continue;
            float4 w = v[u];
            w.x = (w.x - mean) * rstd * ga[u].x + be[u].x;
            w.y = (w.y - mean) * rstd * ga[u].y + be[u].y;
            w.z = (w.z - mean) * rstd * ga[u].z + be[u].z;
            w.w = (w.w - mean) * rstd * ga[u].w + be[u].w;
            if (act == AED_ACT_SILU) {
                w.x = w.x / (1.0f + expf(-w.x));
                w.y = w.y / (1.0f + expf(-w.y));
                w.z = w.z / (1.0f + expf(-w.z));
                w.w = w.w / (1.0f + expf(-w.w));
            }
            *reinterpret_cast<float4*>(yb + (size_t)rowv[u] * ldy + 4 * jv[u]) = w;
        }
    }
}
// Register-resident form for slices of at most 256*U float4 (every U-Net GroupNorm at levels 1-3 at batch 2): the slice is
// loaded ONCE, reduced, normalised from registers and stored -- no second read of the activation, one latency chain less
// (same arithmetic and summation order as gn_small_kernel: per-thread partial sums in element order, fp64 block combine).
template <int U>
__global__ __launch_bounds__(256) void gn_small_reg_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* __restrict__ y,
                                                            int HW, int C, int G, int ldx, int ldy, float eps, int act,
                                                            const float* __restrict__ x2, int C1, int ldx2) {
    __shared__ double rs[4], rss[4];
    const int tid = threadIdx.x, g = blockIdx.x, b = blockIdx.y;
    const int cpg = C / G, cpg4 = cpg >> 2;
    const size_t rb = (size_t)b * HW;
    float* yb = y + (size_t)b * HW * ldy + g * cpg;
    const int total = HW * cpg4;
    float4 v[U], ga[U], be[U];
    int rowv[U], jv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int e = min(tid + 256 * u, total - 1);
        rowv[u] = e / cpg4;
        jv[u] = e - rowv[u] * cpg4;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        v[u] = *reinterpret_cast<const float4*>(gn_src(x, x2, C1, ldx, ldx2, rb + rowv[u], g * cpg + 4 * jv[u]));
        ga[u] = *reinterpret_cast<const float4*>(gamma + g * cpg + 4 * jv[u]);
        be[u] = *reinterpret_cast<const float4*>(beta + g * cpg + 4 * jv[u]);
    }
    float s = 0.f, ss = 0.f;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        if (tid + 256 * u < total) {
            s += (v[u].x + v[u].y) + (v[u].z + v[u].w);
            ss += (v[u].x * v[u].x + v[u].y * v[u].y) + (v[u].z * v[u].z + v[u].w * v[u].w);
        }
    }
    double ds = (double)s, dss = (double)ss;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { ds += __shfl_xor(ds, o, 64); dss += __shfl_xor(dss, o, 64); }
    if ((tid & 63) == 0) { rs[tid >> 6] = ds; rss[tid >> 6] = dss; }
    __syncthreads();
    ds = (rs[0] + rs[1]) + (rs[2] + rs[3]);
    dss = (rss[0] + rss[1]) + (rss[2] + rss[3]);
    const double n = (double)HW * (double)cpg;
    const double dmean = ds / n;
    double var = dss / n - dmean * dmean;
    if (var < 0.0) var = 0.0;
    const float mean = (float)dmean;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
#pragma unroll
    for (int u = 0; u < U; ++u) {
        if (tid + 256 * u >= total) continue;
        float4 w = v[u];
        w.x = (w.x - mean) * rstd * ga[u].x + be[u].x;
        w.y = (w.y - mean) * rstd * ga[u].y + be[u].y;
        w.z = (w.z - mean) * rstd * ga[u].z + be[u].z;
        w.w = (w.w - mean) * rstd * ga[u].w + be[u].w;
        if (act == AED_ACT_SILU) {
            w.x = w.x / (1.0f + expf(-w.x));
            w.y = w.y / (1.0f + expf(-w.y));
            w.z = w.z / (1.0f + expf(-w.z));
            w.w = w.w / (1.0f + expf(-w.w));
        }
        *reinterpret_cast<float4*>(yb + (size_t)rowv[u] * ldy + 4 * jv[u]) = w;
    }
}
// Any channel count per group (TANGO at full size: 320 / 32 = 10 and, after the up-block concat, 960 / 32 = 30 channels per
// group -- not float4 granules; every kernel above loads float4 slices of a group).  One block per (group, batch item), scalar
// loads with a per-element source select, the same fp64 reduction of fp32 partial sums and the same affine / SiLU expression
// as gn_small_kernel.  Correctness path: the widths of the benchmark families never reach it.
__global__ __launch_bounds__(256) void gn_generic_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float* __restrict__ y, int HW,
                                                           int C, int G, int ldx, int ldy, float eps, int act,
                                                           const float* __restrict__ x2, int C1, int ldx2) {
    __shared__ double rs[4], rss[4];
    const int tid = threadIdx.x, g = blockIdx.x, b = blockIdx.y;
    const int cpg = C / G;
    const size_t rb = (size_t)b * HW;
    const int total = HW * cpg;
    auto src = [&](int e) -> const float* {
        const int row = e / cpg, c = g * cpg + (e - row * cpg);
        return (x2 != nullptr && c >= C1) ? x2 + (rb + row) * ldx2 + (c - C1) : x + (rb + row) * ldx + c;
    };
    float s = 0.f, ss = 0.f;
    for (int e = tid; e < total; e += 256) {
        const float v = *src(e);
        s += v;
        ss += v * v;
    }
    double ds = (double)s, dss = (double)ss;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { ds += __shfl_xor(ds, o, 64); dss += __shfl_xor(dss, o, 64); }
    if ((tid & 63) == 0) { rs[tid >> 6] = ds; rss[tid >> 6] = dss; }
    __syncthreads();
    ds = (rs[0] + rs[1]) + (rs[2] + rs[3]);
    dss = (rss[0] + rss[1]) + (rss[2] + rss[3]);
    const double n = (double)HW * (double)cpg;
    const double mean = ds / n;
    double var = dss / n - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps)), mu = (float)mean;
    for (int e = tid; e < total; e += 256) {
        const int row = e / cpg, c = g * cpg + (e - row * cpg);
        float w = (*src(e) - mu) * rstd * gamma[c] + beta[c];
        if (act == AED_ACT_SILU) w = w / (1.0f + expf(-w));
        y[(rb + row) * ldy + c] = w;
    }
}

// slots: p0=x p1=gamma p2=beta p3=y p4=x2(or null) ; i0=B i1=HW i2=C i3=G i4=ldx i5=ldy i6=act i7=1: never register-resident (A/B)
//        i8=C1 i9=ldx2 (two-source rows, see gn_src) ; f0=eps
int launch_gn_small(const aed_op* op, hipStream_t s) {
    const int32_t* i = op->i;
    AED_REQUIRE(op->p[0] && op->p[1] && op->p[2] && op->p[3], "gn_small: null pointer");
    AED_REQUIRE(i[3] > 0 && i[2] % i[3] == 0, "gn_small: C=%d G=%d", i[2], i[3]);
    AED_REQUIRE(!op->p[4] || (i[8] > 0 && i[8] < i[2]), "gn_small: bad two-source split");
    if (i[2] % (4 * i[3]) != 0 || i[4] % 4 != 0 || i[5] % 4 != 0 || (op->p[4] && (i[8] % 4 != 0 || i[9] % 4 != 0))) {
        // channels per group (or a row stride) that is not a float4 granule: the scalar kernel
        hipLaunchKernelGGL(gn_generic_kernel, dim3(i[3], i[0]), dim3(256), 0, s, (const float*)op->p[0], (const float*)op->p[1],
                           (const float*)op->p[2], (float*)op->p[3], i[1], i[2], i[3], i[4], i[5], op->f[0], i[6],
                           (const float*)op->p[4], i[8], i[9]);
        AED_CHECK_HIP(hipGetLastError());
        return 0;
    }
    const int total4 = i[1] * (i[2] / i[3] / 4);        // float4 per (group, batch item) slice
#define GN_ARGS (const float*)op->p[0], (const float*)op->p[1], (const float*)op->p[2], (float*)op->p[3], i[1], i[2], i[3], \
                i[4], i[5], op->f[0], i[6], (const float*)op->p[4], i[8], i[9]
    const dim3 grid(i[3], i[0]);
    if (i[7] != 1 && total4 <= 256 * 2) hipLaunchKernelGGL(gn_small_reg_kernel<2>, grid, dim3(256), 0, s, GN_ARGS);
    else if (i[7] != 1 && total4 <= 256 * 4) hipLaunchKernelGGL(gn_small_reg_kernel<4>, grid, dim3(256), 0, s, GN_ARGS);
    else if (i[7] != 1 && total4 <= 256 * 8) hipLaunchKernelGGL(gn_small_reg_kernel<8>, grid, dim3(256), 0, s, GN_ARGS);
    else hipLaunchKernelGGL(gn_small_kernel<4>, grid, dim3(256), 0, s, GN_ARGS);
#undef GN_ARGS
    AED_CHECK_HIP(hipGetLastError());
    return 0;
}

// slots: p0=x p1=gamma p2=beta p3=ab[B][2][C] ; i0=B i1=HW i2=C i3=G i4=ldx ; f0=eps
int launch_gn_scale_shift(const aed_op* op, hipStream_t s) {
    const int32_t* i = op->i;
    AED_REQUIRE(op->p[0] && op->p[1] && op->p[2] && op->p[3], "gn_scale_shift: null pointer");
    AED_REQUIRE(i[2] % (4 * i[3]) == 0 && i[2] / i[3] <= 256 && i[4] % 4 == 0, "gn_scale_shift: C=%d G=%d", i[2], i[3]);
    hipLaunchKernelGGL(gn_scale_shift_kernel, dim3(i[3], i[0]), dim3(256), 0, s, (const float*)op->p[0],
                       (const float*)op->p[1], (const float*)op->p[2], (float*)op->p[3], i[1], i[2], i[3], i[4],
                       op->f[0]);
    AED_CHECK_HIP(hipGetLastError());
    return 0;
}

// slots: p0=x p1=partials p2=x2(or null) ; i0=B i1=HW i2=C i3=G i4=ldx i5=rows_per_chunk i6=nchunks i7=C1 i8=ldx2
int launch_gn_stats(const aed_op* op, hipStream_t s) {
    const int32_t* i = op->i;
    AED_REQUIRE(op->p[0] && op->p[1], "gn_stats: null pointer");
    AED_REQUIRE(i[3] <= 64 && i[2] % (4 * i[3]) == 0, "gn_stats: C=%d must be a multiple of 4*G (G=%d<=64)", i[2], i[3]);
    AED_REQUIRE(i[4] % 4 == 0, "gn_stats: ldx %% 4");
    hipLaunchKernelGGL(gn_stats_kernel, dim3(i[6], i[0]), dim3(256), 0, s, (const float*)op->p[0], (float*)op->p[1],
                       i[1], i[2], i[3], i[4], i[5], i[6], (const float*)op->p[2], i[7], i[8]);
    AED_CHECK_HIP(hipGetLastError());
    return 0;
}

// slots: p0=x p1=partials p2=gamma p3=beta p4=y p5=x2(or null) ; i0..i4 as gn_stats, i5=apply rows/block,
//        i6=#stats partials, i7=act, i8=ldy, i9=#apply blocks per batch item, i10=C1 i11=ldx2 ; f0=eps
int launch_gn_apply(const aed_op* op, hipStream_t s) {
    const int32_t* i = op->i;
    AED_REQUIRE(op->p[0] && op->p[1] && op->p[2] && op->p[3] && op->p[4], "gn_apply: null pointer");
    AED_REQUIRE(i[3] <= 64 && 256 % i[3] == 0 && i[2] % (4 * i[3]) == 0, "gn_apply: C=%d G=%d", i[2], i[3]);
    hipLaunchKernelGGL(gn_apply_kernel, dim3(i[9], i[0]), dim3(256), 0, s, (const float*)op->p[0],
                       (const float*)op->p[1], (const float*)op->p[2], (const float*)op->p[3], (float*)op->p[4], i[1],
                       i[2], i[3], i[4], i[8], i[5], i[6], op->f[0], i[7], (const float*)op->p[5], i[10], i[11]);
    AED_CHECK_HIP(hipGetLastError());
    return 0;
}

