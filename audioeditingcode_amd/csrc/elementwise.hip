// elementwise.hip -- the HBM-bound small kernels of the path (SURVEY K1, K8, K9, K10, K2 tail).
// Compiled with -ffp-contract=off: the step-math kernels reproduce the reference's fp32
// expression order (models.py:85-158) bit-for-bit, so no FMA contraction is allowed here.
#include "aed_common.h"

static inline int grid_for(size_t n, int per_thread = 1) {
    size_t g = (n + 256 * (size_t)per_thread - 1) / (256 * (size_t)per_thread);
    if (g > 4096) g = 4096;
    if (g < 1) g = 1;
    return (int)g;
}

// ------------------------------------------------------------------------------------ copy2d
// Optional device-indexed source: src += (idx_off + idx_mul * state[0]) * idx_stride elements, so a
// captured graph can walk the xts trajectory (inversion_utils.py:78 `xt = xts[idx+1]`).
__global__ __launch_bounds__(256) void copy2d_kernel(const float* __restrict__ src, float* __restrict__ dst, int rows,
                                                      int cols, int lds_, int ldd, int vec, const int* state,
                                                      int idx_off, int idx_mul, long idx_stride, const float* coef,
                                                      int c_mul, int c_off, int c_stride, int c_col) {
    const int st = state ? state[0] : 0;
    if (state) src += (size_t)(idx_off + idx_mul * st) * (size_t)idx_stride;
    // optional per-step scale read from a device coefficient table (Stable Audio: scheduler.scale_model_input)
    const float sc = coef ? coef[(size_t)(st * c_mul + c_off) * c_stride + c_col] : 1.0f;
    if (vec) {
        const int q = cols >> 2;
        const size_t total = (size_t)rows * q;
        for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
            const int r = (int)(e / q);
            const int c = (int)(e - (size_t)r * q) * 4;
            float4 v = *reinterpret_cast<const float4*>(src + (size_t)r * lds_ + c);
            if (coef) { v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc; }
            *reinterpret_cast<float4*>(dst + (size_t)r * ldd + c) = v;
        }
    } else {
        const size_t total = (size_t)rows * cols;
        for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
            const int r = (int)(e / cols);
            const int c = (int)(e - (size_t)r * cols);
            const float v = src[(size_t)r * lds_ + c];
            dst[(size_t)r * ldd + c] = coef ? v * sc : v;
        }
    }
}
// slots: p0=src p1=dst p2=state(nullable) p3=coef table (nullable) ; i0=rows i1=cols i2=ld_src i3=ld_dst i4=idx_off
//        i5=idx_mul i6=idx_stride ; scale = coef[(state*i7 + i8)*i9 + i10]
int launch_copy2d(const aed_op* op, hipStream_t s) {
    const int32_t* i = op->i;
    AED_REQUIRE(op->p[0] && op->p[1], "copy2d: null pointer");
    const int vec = (i[1] % 4 == 0) && (i[2] % 4 == 0) && (i[3] % 4 == 0) && ((uintptr_t)op->p[0] % 16 == 0) &&
                    ((uintptr_t)op->p[1] % 16 == 0) && (i[6] % 4 == 0);
    hipLaunchKernelGGL(copy2d_kernel, dim3(grid_for((size_t)i[0] * i[1] / (vec ? 4 : 1))), dim3(256), 0, s,
                       (const float*)op->p[0], (float*)op->p[1], i[0], i[1], i[2], i[3], vec, (const int*)op->p[2],
                       i[4], i[5], (long)i[6], (const float*)op->p[3], i[7], i[8], i[9], i[10]);
    AED_CHECK_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------ timestep embedding
__global__ void time_embed_kernel(float* out, const long long* tt, const int* state, const float* freqs,
                                  const int* row_tidx, int B, int dim, int flip, int ld, int t_imm, int tgroup,
                                  float shift, float max_period, int float_table) {
    const int half = dim / 2;
    const int s = state ? state[0] : 0;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < B * half; e += gridDim.x * blockDim.x) {
        const int b = e / half, i = e - b * half;
        // row b of a timestep-batched call uses table entry s*tgroup + row_tidx[b]
        const int ti = s * tgroup + (row_tidx ? row_tidx[b] : 0);
        // (float_table: the table holds fp32 values -- Stable Audio's continuous timesteps, already times 2*pi)
        const float t = !tt ? (float)t_imm : float_table ? reinterpret_cast<const float*>(tt)[ti] : (float)tt[ti];
        // freqs (host table, computed exactly as diffusers' Timesteps does) keeps t*freq bit-identical
        const float fr = freqs ? freqs[i] : expf(-logf(max_period) * (float)i / ((float)half - shift));
        const float arg = t * fr;
        const float sv = sinf(arg), cv = cosf(arg);
        float* o = out + (size_t)b * ld;
        if (flip) { o[i] = cv; o[half + i] = sv; }
        else { o[i] = sv; o[half + i] = cv; }
    }
}
// slots: p0=out[B,dim] p1=timesteps(int64 dev, nullable) p2=state(int32 dev, nullable) p3=freqs[dim/2] (nullable)
//        p4=row_tidx(int32[B], nullable)   i5 = timesteps per call (tgroup, default 1)
//        i0=B i1=dim i2=flip_sin_to_cos i3=ld i4=t_imm i6=1: p1 is a float32 table ; f0=freq_shift f1=max_period
int launch_time_embed(const aed_op* op, hipStream_t s) {
    const int32_t* i = op->i;
    AED_REQUIRE(op->p[0] && i[1] % 2 == 0, "time_embed: bad args");
    hipLaunchKernelGGL(time_embed_kernel, dim3(aed_cdiv(i[0] * i[1] / 2, 256)), dim3(256), 0, s, (float*)op->p[0],
                       (const long long*)op->p[1], (const int*)op->p[2], (const float*)op->p[3], (const int*)op->p[4],
                       i[0], i[1], i[2], i[3], i[4], i[5] > 0 ? i[5] : 1, op->f[0],
                       op->f[1] > 0.f ? op->f[1] : 10000.0f, i[6]);
    AED_CHECK_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------ row softmax
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ x, float* __restrict__ y, int cols,
                                                            int ldx, int ldy, float scale) {
    __shared__ float red[4];
    const int row = blockIdx.x, tid = threadIdx.x;
    const float* xr = x + (size_t)row * ldx;
    float* yr = y + (size_t)row * ldy;
    float m = -INFINITY;
    for (int c = tid; c < cols; c += 256) m = fmaxf(m, xr[c] * scale);
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((tid & 63) == 0) red[tid >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float sum = 0.f;
    for (int c = tid; c < cols; c += 256) {
        const float e = expf(xr[c] * scale - m);
        yr[c] = e;
        sum += e;
    }
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    if ((tid & 63) == 0) red[tid >> 6] = sum;
    __syncthreads();
    sum = (red[0] + red[1]) + (red[2] + red[3]);
    const float inv = 1.0f / sum;
    for (int c = tid; c < cols; c += 256) yr[c] *= inv;
}
// slots: p0=x p1=y ; i0=rows i1=cols i2=ldx i3=ldy ; f0=scale
int launch_softmax_rows(const aed_op* op, hipStream_t s) {
    const int32_t* i = op->i;
    AED_REQUIRE(op->p[0] && op->p[1], "softmax_rows: null pointer");
    hipLaunchKernelGGL(softmax_rows_kernel, dim3(i[0]), dim3(256), 0, s, (const float*)op->p[0], (float*)op->p[1], i[1],
                       i[2], i[3], op->f[0]);
    AED_CHECK_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------ batched transpose
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ src, float* __restrict__ dst, int R,
                                                         int C, int lds_, int ldd, long bs_src, long bs_dst) {
    __shared__ float tile[32][33];
    const float* s = src + (size_t)blockIdx.z * bs_src;
    float* d = dst + (size_t)blockIdx.z * bs_dst;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    for (int j = ty; j < 32; j += 8) {
        const int r = r0 + j, c = c0 + tx;
        tile[j][tx] = (r < R && c < C) ? s[(size_t)r * lds_ + c] : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int c = c0 + j, r = r0 + tx;
        if (c < C && r < R) d[(size_t)c * ldd + r] = tile[tx][j];
    }
}
// slots: p0=src[Bt][R][C] p1=dst[Bt][C][R] ; i0=Bt i1=R i2=C i3=ld_src i4=ld_dst i5=bs_src i6=bs_dst
int launch_transpose(const aed_op* op, hipStream_t s) {
    const int32_t* i = op->i;
    AED_REQUIRE(op->p[0] && op->p[1], "transpose: null pointer");
    hipLaunchKernelGGL(transpose_kernel, dim3(aed_cdiv(i[2], 32), aed_cdiv(i[1], 32), i[0]), dim3(256), 0, s,
                       (const float*)op->p[0], (float*)op->p[1], i[1], i[2], i[3], i[4], (long)i[5], (long)i[6]);
    AED_CHECK_HIP(hipGetLastError());
    return 0;
}
int launch_layout(const aed_op* op, hipStream_t s) { return launch_transpose(op, s); }

// ------------------------------------------------------------------------------------ axpby
__global__ __launch_bounds__(256) void axpby_kernel(const float* __restrict__ x, float* __restrict__ y, size_t n, float a,
                                                     float b) {
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) {
        const float v = a * x[e];
        y[e] = (b == 0.0f) ? v : v + b * y[e];
    }
}
// slots: p0=x p1=y ; i0,i1 = numel lo/hi ; f0=a f1=b     (y = a*x + b*y)
int launch_axpby(const aed_op* op, hipStream_t s) {
    const size_t n = (size_t)(uint32_t)op->i[0] | ((size_t)(uint32_t)op->i[1] << 32);
    AED_REQUIRE(op->p[0] && op->p[1], "axpby: null pointer");
    hipLaunchKernelGGL(axpby_kernel, dim3(grid_for(n)), dim3(256), 0, s, (const float*)op->p[0], (float*)op->p[1], n,
                       op->f[0], op->f[1]);
    AED_CHECK_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------ K1: step math
struct StepParams {
    float* xts;            // device-indexed mode: base of xts [T+1][numel]; explicit mode: xt
    float* zs;             // device-indexed: base of zs [*][numel];        explicit: z (in for reverse, out for invert)
    float* xtm1;           // explicit mode only (invert): x_{t-1} (in/out)
    const float* eps_u;
    const float* eps_c;    // [P][numel] or null
    const float* cfg;      // [P][numel] or null -> cfg_scalar
    const float* coef;     // device table [steps][8] or null -> c[] immediates
    const int* state;      // device step counter or null -> s_imm
    float* out;            // reverse: x_{t-1} out ; invert: optional noise_pred out
    size_t numel;
    int P, T, s_imm, v_pred, fix, has_noise, s_mul, s_off;
    float cfg_scalar;
    float c[8];
};

__device__ __forceinline__ float cfg_combine(const StepParams& p, size_t e) {
    const float u = p.eps_u[e];
    if (!p.eps_c) return u;
    float acc = 0.f;
    for (int j = 0; j < p.P; ++j) {
        const float g = p.cfg ? p.cfg[(size_t)j * p.numel + e] : p.cfg_scalar;
        const float term = g * (p.eps_c[(size_t)j * p.numel + e] - u);
        acc = (j == 0) ? term : acc + term;
    }
    return u + acc;
}

__global__ __launch_bounds__(256) void invert_step_kernel(StepParams p) {
    const int s = p.state ? p.state[0] * p.s_mul + p.s_off : p.s_imm;
    float c0, c1, c2, c3, c4;
    if (p.coef) { const float* c = p.coef + (size_t)s * AED_COEF_STRIDE; c0 = c[0]; c1 = c[1]; c2 = c[2]; c3 = c[3]; c4 = c[4]; }
    else { c0 = p.c[0]; c1 = p.c[1]; c2 = p.c[2]; c3 = p.c[3]; c4 = p.c[4]; }
    const float* xt;
    float *xtm1, *z;
    if (p.xtm1) { xt = p.xts; xtm1 = p.xtm1; z = p.zs; }
    else {
        const int idx = p.T - s - 1;          // inversion_utils.py:75
        xt = p.xts + (size_t)(idx + 1) * p.numel;
        xtm1 = p.xts + (size_t)idx * p.numel;
        z = p.zs + (size_t)idx * p.numel;
    }
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < p.numel; e += (size_t)gridDim.x * 256) {
        const float eps = cfg_combine(p, e);
        const float x = xt[e];
        float x0, dir;
        if (!p.v_pred) { x0 = (x - c0 * eps) / c1; dir = eps; }
        else { x0 = c1 * x - c0 * eps; dir = c1 * eps + c0 * x; }
        const float mu = c2 * x0 + c3 * dir;
        const float zz = (xtm1[e] - mu) / c4;
        z[e] = zz;
        if (p.fix) xtm1[e] = mu + c4 * zz;
        if (p.out) p.out[e] = eps;
    }
}

__global__ __launch_bounds__(256) void reverse_step_kernel(StepParams p) {
    const int s = p.state ? p.state[0] * p.s_mul + p.s_off : p.s_imm;
    float c0, c1, c2, c3, c4;
    if (p.coef) { const float* c = p.coef + (size_t)s * AED_COEF_STRIDE; c0 = c[0]; c1 = c[1]; c2 = c[2]; c3 = c[3]; c4 = c[4]; }
    else { c0 = p.c[0]; c1 = p.c[1]; c2 = p.c[2]; c3 = p.c[3]; c4 = p.c[4]; }
    const float* xt = p.xts;
    const float* z = nullptr;
    if (p.has_noise) z = (p.T > 0) ? p.zs + (size_t)(p.T - s - 1) * p.numel : p.zs;   // T := number of zs (Z); 0 => explicit z
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < p.numel; e += (size_t)gridDim.x * 256) {
        const float eps = cfg_combine(p, e);
        const float x = xt[e];
        float x0, dir;
        if (!p.v_pred) { x0 = (x - c0 * eps) / c1; dir = eps; }
        else { x0 = c1 * x - c0 * eps; dir = c1 * eps + c0 * x; }
        float prev = c2 * x0 + c3 * dir;
        if (z) prev = prev + c4 * z[e];
        p.out[e] = prev;
    }
}

// slots (both): p0=xts base | xt   p1=zs base | z   p2=eps_u  p3=eps_c  p4=cfg  p5=coef table  p6=state  p7=out
//   i0,i1=numel lo/hi  i2=P  i3=T (invert: #steps; reverse: #zs, 0 => p1 is the explicit z)  i4=s_imm
//   i5=v_pred  i6=numerical_fix | has_noise   i7=s_mul i8=s_off (step = state*s_mul + s_off; timestep-batched loops)
//   f0=cfg_scalar  f1..f5 = c0..c4 immediates (used when p5 is null)
static void fill_step(const aed_op* op, StepParams& p) {
    p.xts = (float*)op->p[0]; p.zs = (float*)op->p[1]; p.xtm1 = nullptr;
    p.eps_u = (const float*)op->p[2]; p.eps_c = (const float*)op->p[3]; p.cfg = (const float*)op->p[4];
    p.coef = (const float*)op->p[5]; p.state = (const int*)op->p[6]; p.out = (float*)op->p[7];
    p.numel = (size_t)(uint32_t)op->i[0] | ((size_t)(uint32_t)op->i[1] << 32);
    p.P = op->i[2]; p.T = op->i[3]; p.s_imm = op->i[4]; p.v_pred = op->i[5];
    p.fix = op->i[6]; p.has_noise = op->i[6];
    p.s_mul = op->i[7] > 0 ? op->i[7] : 1; p.s_off = op->i[8];
    p.cfg_scalar = op->f[0];
    for (int k = 0; k < 5; ++k) p.c[k] = op->f[1 + k];
}
int launch_invert_step(const aed_op* op, hipStream_t s) {
    StepParams p;
    fill_step(op, p);
    AED_REQUIRE(p.xts && p.zs && p.eps_u, "invert_step: null pointer");
    AED_REQUIRE(p.eps_c == nullptr || p.P >= 1, "invert_step: P");
    hipLaunchKernelGGL(invert_step_kernel, dim3(grid_for(p.numel)), dim3(256), 0, s, p);
    AED_CHECK_HIP(hipGetLastError());
    return 0;
}
int launch_reverse_step(const aed_op* op, hipStream_t s) {
    StepParams p;
    fill_step(op, p);
    AED_REQUIRE(p.xts && p.eps_u && p.out, "reverse_step: null pointer");
    AED_REQUIRE(!p.has_noise || p.zs, "reverse_step: noise requested but zs is null");
    hipLaunchKernelGGL(reverse_step_kernel, dim3(grid_for(p.numel)), dim3(256), 0, s, p);
    AED_CHECK_HIP(hipGetLastError());
    return 0;
}
int launch_ddim_step(const aed_op* op, hipStream_t s) { return launch_reverse_step(op, s); }

// explicit-pointer entry points (the reference methods' signatures)
extern "C" int aed_get_zs_from_xts(const float* xt, float* xtm1, const float* eps_u, const float* eps_c,
                                   const float* cfg, float cfg_scalar, int n_prompts, const float* coef_host,
                                   int v_prediction, int numerical_fix, float* z, float* noise_pred_out,
                                   int64_t numel, void* stream) {
    StepParams p = {};
    p.s_mul = 1;
    p.xts = const_cast<float*>(xt); p.xtm1 = xtm1; p.zs = z; p.eps_u = eps_u; p.eps_c = eps_c; p.cfg = cfg;
    p.cfg_scalar = cfg_scalar; p.P = n_prompts; p.v_pred = v_prediction; p.fix = numerical_fix;
    p.out = noise_pred_out; p.numel = (size_t)numel;
    AED_REQUIRE(xt && xtm1 && eps_u && z && coef_host, "aed_get_zs_from_xts: null pointer");
    for (int k = 0; k < 5; ++k) p.c[k] = coef_host[k];
    hipLaunchKernelGGL(invert_step_kernel, dim3(grid_for(p.numel)), dim3(256), 0, (hipStream_t)stream, p);
    AED_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int aed_reverse_step_with_custom_noise(const float* xt, const float* eps_u, const float* eps_c,
                                                  const float* cfg, float cfg_scalar, int n_prompts,
                                                  const float* coef_host, int v_prediction, const float* z,
                                                  float* prev_out, int64_t numel, void* stream) {
    StepParams p = {};
    p.s_mul = 1;
    p.xts = const_cast<float*>(xt); p.zs = const_cast<float*>(z); p.eps_u = eps_u; p.eps_c = eps_c; p.cfg = cfg;
    p.cfg_scalar = cfg_scalar; p.P = n_prompts; p.v_pred = v_prediction; p.has_noise = z != nullptr; p.T = 0;
    p.out = prev_out; p.numel = (size_t)numel;
    AED_REQUIRE(xt && eps_u && prev_out && coef_host, "aed_reverse_step_with_custom_noise: null pointer");
    for (int k = 0; k < 5; ++k) p.c[k] = coef_host[k];
    hipLaunchKernelGGL(reverse_step_kernel, dim3(grid_for(p.numel)), dim3(256), 0, (hipStream_t)stream, p);
    AED_CHECK_HIP(hipGetLastError());
    return 0;
}

__global__ __launch_bounds__(256) void sample_xts_kernel(const float* __restrict__ x0, const float* __restrict__ noise,
                                                          const float* __restrict__ sa, const float* __restrict__ sb,
                                                          float* __restrict__ out, int nt, size_t numel) {
    const size_t total = (size_t)nt * numel;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const size_t r = e / numel;
        const size_t i = e - r * numel;
        out[e] = x0[i] * sa[r] + noise[e] * sb[r];      // models.py:81
    }
}
extern "C" int aed_sample_xts_from_x0(const float* x0, const float* noise, const float* sqrt_abar,
                                      const float* sqrt_1m_abar, float* xts_out, int n_t, int64_t numel,
                                      void* stream) {
    AED_REQUIRE(x0 && noise && sqrt_abar && sqrt_1m_abar && xts_out, "aed_sample_xts_from_x0: null pointer");
    hipLaunchKernelGGL(sample_xts_kernel, dim3(grid_for((size_t)n_t * numel)), dim3(256), 0, (hipStream_t)stream, x0,
                       noise, sqrt_abar, sqrt_1m_abar, xts_out, n_t, (size_t)numel);
    AED_CHECK_HIP(hipGetLastError());
    return 0;
}

__global__ void advance_kernel(int* state, int by) {
    if (threadIdx.x == 0 && blockIdx.x == 0) state[0] += by;
}
// slots: p0=state ; i0=increment
int launch_advance(const aed_op* op, hipStream_t s) {
    AED_REQUIRE(op->p[0], "advance: null state");
    hipLaunchKernelGGL(advance_kernel, dim3(1), dim3(64), 0, s, (int*)op->p[0], op->i[0] ? op->i[0] : 1);
    AED_CHECK_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------ STFT helpers
__global__ __launch_bounds__(256) void reflect_pad_kernel(const float* __restrict__ src, float* __restrict__ dst, int B,
                                                           int N, int pad, int ldd) {
    const int L = N + 2 * pad;
    const size_t total = (size_t)B * L;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const int b = (int)(e / L);
        int j = (int)(e - (size_t)b * L) - pad;
        if (j < 0) j = -j;
        if (j >= N) j = 2 * (N - 1) - j;
        dst[(size_t)b * ldd + (e - (size_t)b * L)] = src[(size_t)b * N + j];
    }
}
// slots: p0=src[B,N] p1=dst[B,ldd] ; i0=B i1=N i2=pad i3=ldd
int launch_reflect_pad(const aed_op* op, hipStream_t s) {
    const int32_t* i = op->i;
    AED_REQUIRE(op->p[0] && op->p[1] && i[2] < i[1], "reflect_pad: bad args");
    hipLaunchKernelGGL(reflect_pad_kernel, dim3(grid_for((size_t)i[0] * (i[1] + 2 * i[2]))), dim3(256), 0, s,
                       (const float*)op->p[0], (float*)op->p[1], i[0], i[1], i[2], i[3]);
    AED_CHECK_HIP(hipGetLastError());
    return 0;
}

__global__ __launch_bounds__(256) void magnitude_kernel(const float* __restrict__ ft, float* __restrict__ mag, int F,
                                                         int cut, int ldf, int ldm) {
    const size_t total = (size_t)F * ldm;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const int f = (int)(e / ldm);
        const int k = (int)(e - (size_t)f * ldm);
        float v = 0.f;
        if (k < cut) {
            const float re = ft[(size_t)f * ldf + k], im = ft[(size_t)f * ldf + cut + k];
            v = sqrtf(re * re + im * im);
        }
        mag[e] = v;
    }
}
// slots: p0=ft[F,2*cut] p1=mag[F,ldm] (zero padded cols) ; i0=F i1=cut i2=ld_ft i3=ld_mag
int launch_magnitude(const aed_op* op, hipStream_t s) {
    const int32_t* i = op->i;
    AED_REQUIRE(op->p[0] && op->p[1], "magnitude: null pointer");
    hipLaunchKernelGGL(magnitude_kernel, dim3(grid_for((size_t)i[0] * i[3])), dim3(256), 0, s, (const float*)op->p[0],
                       (float*)op->p[1], i[0], i[1], i[2], i[3]);
    AED_CHECK_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------ folded cross-attention operands
// Once per prompt set (context tape), per transformer block with text cross-attention (unet.py
// UNetEngine._folded_cross_attention; reference math: attention over encoder_hidden_states, models.py:806-894):
//   G  [b, h*Lk+j, c]  = sum_d k[b,j,hD+d] * xq[h][c][d]     xq = (gamma o Wq_h)^T       (scores = LN-folded x . G^T)
//   gs [b, h*Lk+j, 0/1] = sum_d k[b,j,hD+d] * xs[h][0/1][d]   xs = (Wq_h gamma, Wq_h beta) (the LayerNorm-fold vectors)
//   VOt[b, c, h*Lk+j]  = sum_d v[b,j,hD+d] * xo[h][c][d]     xo = Wo[:, hD:(h+1)D]        (out = P . VO + bo)
// One thread per (b, hj, c); D-long dot products, a few MFLOP in total -- launch count, not arithmetic, is the cost.
__global__ __launch_bounds__(256) void xattn_fold_kernel(const float* __restrict__ kv, const float* __restrict__ xq,
                                                          const float* __restrict__ xs, const float* __restrict__ xo,
                                                          float* __restrict__ G, float* __restrict__ gs,
                                                          float* __restrict__ VOt, int B, int Lk, int H, int C, int D,
                                                          int ldkv) {
    const int HL = H * Lk;
    const size_t total = (size_t)B * HL * C;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int c = (int)(idx % C);
        const int hj = (int)((idx / C) % HL);
        const int b = (int)(idx / ((size_t)C * HL));
        const int h = hj / Lk, j = hj - h * Lk;
        const float* k = kv + (size_t)(b * Lk + j) * ldkv + h * D;
        const float* v = k + C;
        const float* wq = xq + ((size_t)h * C + c) * D;
        const float* wo = xo + ((size_t)h * C + c) * D;
        float g = 0.f, o = 0.f;
        for (int d = 0; d < D; d += 4) {
            const float4 k4 = *reinterpret_cast<const float4*>(k + d), v4 = *reinterpret_cast<const float4*>(v + d);
            const float4 q4 = *reinterpret_cast<const float4*>(wq + d), o4 = *reinterpret_cast<const float4*>(wo + d);
            g += k4.x * q4.x + k4.y * q4.y + k4.z * q4.z + k4.w * q4.w;
            o += v4.x * o4.x + v4.y * o4.y + v4.z * o4.z + v4.w * o4.w;
        }
        G[idx] = g;
        VOt[((size_t)b * C + c) * HL + hj] = o;
        if (c < 2) {
            const float* ws = xs + ((size_t)h * 2 + c) * D;
            float s_ = 0.f;
            for (int d = 0; d < D; ++d) s_ += k[d] * ws[d];
            gs[((size_t)b * HL + hj) * 2 + c] = s_;
        }
    }
}
// slots: p0=kv [B*Lk, ldkv] (k | v halves of width C) p1=xq [H,C,D] p2=xs [H,2,D] p3=xo [H,C,D] p4=G [B,H*Lk,C]
//        p5=gs [B,H*Lk,2] p6=VOt [B,C,H*Lk] ; i0=B i1=Lk i2=H i3=C i4=D i5=ldkv
int launch_xattn_fold(const aed_op* op, hipStream_t s) {
    const int32_t* i = op->i;
    for (int k = 0; k < 7; ++k) AED_REQUIRE(op->p[k] != nullptr, "xattn_fold: null pointer (slot %d)", k);
    AED_REQUIRE(i[3] == i[2] * i[4] && i[4] % 4 == 0 && i[5] % 4 == 0 && i[5] >= 2 * i[3],
                "xattn_fold: C=%d must be H*D (H=%d, D=%d multiple of 4), ldkv=%d >= 2C", i[3], i[2], i[4], i[5]);
    const size_t total = (size_t)i[0] * i[2] * i[1] * i[3];
    hipLaunchKernelGGL(xattn_fold_kernel, dim3(grid_for(total)), dim3(256), 0, s, (const float*)op->p[0],
                       (const float*)op->p[1], (const float*)op->p[2], (const float*)op->p[3], (float*)op->p[4],
                       (float*)op->p[5], (float*)op->p[6], i[0], i[1], i[2], i[3], i[4], i[5]);
    AED_CHECK_HIP(hipGetLastError());
    return 0;
}
