// stable_audio.hip -- the elementwise kernels of the Stable Audio Open path (BASELINE config 5):
//   AED_OP_SA_STEP      CFG + StableAudWrapper.get_zs_from_xts / reverse_step_with_custom_noise
//                       (/root/reference/code/models.py:1209-1271, :1282-1329): SDE-DPM-Solver++ of order 1 / 2 with the
//                       previous data prediction kept ON THE DEVICE (the scheduler's `model_outputs` history)
//   AED_OP_ROTARY       partial rotary embedding of q and k inside the fused qkv buffer (DiT self-attention)
//   AED_OP_SNAKE        Snake1d activation of the Oobleck VAE: x + sin^2(a x) / (b + 1e-9), per channel
//   AED_OP_GAUSS_SAMPLE mean + (softplus(scale) + 1e-4) * noise (OobleckDiagonalGaussianDistribution.sample, models.py:1132-1133)
// All HBM-bound, one pass each.  Compiled with -ffp-contract=off: the step math keeps the reference's expression order
// so that it agrees with the torch-CPU arithmetic to the last bit wherever libm agrees.
#include "aed_common.h"

static inline int sa_grid(size_t n) {
    size_t b = (n + 255) / 256;
    const size_t cap = (size_t)aed_num_cus() * 8;
    return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

// ------------------------------------------------------------------------------------ solver step
struct SAStep {
    float* xts;           // invert: trajectory base [T+1, numel] | explicit x_t ; reverse: current sample
    float* xtm1;          // invert, explicit-pointer form only
    float* zs;            // invert: noise maps base [T, numel] | explicit z ; reverse: zs base [Z, numel] | explicit z | null
    const float* v_u;     // model output, unconditional pass
    const float* v_c;     // model output, conditional pass (nullable: empty source prompt)
    const float* coef;    // device table [steps, AED_SA_COEF_STRIDE] (nullable -> c[])
    const int* state;     // device step counter (nullable -> s_imm)
    float* hist;          // previous data prediction m1 (read), this step's data prediction (written)
    float* extra;         // invert: base [T, numel] receiving m1 per step (the reference's extra_info), nullable
    float* out;           // reverse: x_{t-1}
    size_t numel;
    int T, mode, s_mul, s_off, s_imm, fix, explicit_ptrs;
    float cfg;
    float c[AED_SA_COEF_STRIDE];
};

__global__ __launch_bounds__(256) void sa_step_kernel(SAStep p) {
    const int s = p.state ? p.state[0] * p.s_mul + p.s_off : p.s_imm;
    const float* c = p.coef ? p.coef + (size_t)s * AED_SA_COEF_STRIDE : p.c;
    const float c_skip = c[1], c_out = c[2], k1 = c[3], k2 = c[4], k3 = c[5], inv_r0 = c[6];
    const bool second = c[7] > 1.5f, zero_z = c[8] > 0.5f;
    const float half_k2 = 0.5f * k2;
    const float* xt;
    float *xtm1 = nullptr, *z = nullptr, *extra = nullptr;
    if (p.mode == 0) {
        if (p.explicit_ptrs) { xt = p.xts; xtm1 = p.xtm1; z = p.zs; extra = p.extra; }
        else {
            const int idx = p.T - s - 1;                          // inversion_utils.py:75
            xt = p.xts + (size_t)(idx + 1) * p.numel;
            xtm1 = p.xts + (size_t)idx * p.numel;
            z = p.zs + (size_t)idx * p.numel;
            extra = p.extra ? p.extra + (size_t)idx * p.numel : nullptr;
        }
    } else {
        xt = p.xts;
        if (p.zs) z = (p.explicit_ptrs || p.T <= 0) ? p.zs : p.zs + (size_t)(p.T - s - 1) * p.numel;   // T := number of zs
    }
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < p.numel; e += (size_t)gridDim.x * 256) {
        const float u = p.v_u[e];
        const float v = p.v_c ? u + p.cfg * (p.v_c[e] - u) : u;        // inversion_utils.py:97-102 with one prompt
        const float x = xt[e];
        const float d = c_skip * x + c_out * v;                        // scheduler.convert_model_output (v-prediction)
        const float m1 = p.hist[e];
        const float D1 = second ? inv_r0 * (d - m1) : 0.0f;
        if (p.mode == 0) {
            float zz;
            if (zero_z) zz = 0.0f;                                     // models.py:1235-1236
            else if (!second) zz = ((xtm1[e] - k1 * x) - k2 * d) / k3;                      // models.py:1238-1241
            else zz = (((xtm1[e] - k1 * x) - k2 * d) - half_k2 * D1) / k3;                  // models.py:1243-1255
            z[e] = zz;
            if (p.fix) xtm1[e] = second ? ((k1 * x + k2 * d) + half_k2 * D1) + k3 * zz : (k1 * x + k2 * d) + k3 * zz;
            if (extra) extra[e] = m1;
        } else {
            const float zz = z ? z[e] : 0.0f;
            p.out[e] = second ? ((k1 * x + k2 * d) + half_k2 * D1) + k3 * zz : (k1 * x + k2 * d) + k3 * zz;
        }
        p.hist[e] = d;
    }
}

// slots: p0=xts base | x_t   p1=zs base | z   p2=v_u  p3=v_c (nullable)  p4=coef table  p5=state  p6=hist  p7=out (reverse)
//        p8=extra base (invert, nullable)
//   i0,i1=numel lo/hi  i2=mode (0 invert, 1 reverse)  i3=T (invert: #steps; reverse: #zs)  i4=s_imm  i5=numerical_fix
//   i6=s_mul i7=s_off (step = state*s_mul + s_off; timestep-batched inversion)   f0=cfg scale
int launch_sa_step(const aed_op* op, hipStream_t s) {
    SAStep p = {};
    p.xts = (float*)op->p[0]; p.zs = (float*)op->p[1]; p.v_u = (const float*)op->p[2]; p.v_c = (const float*)op->p[3];
    p.coef = (const float*)op->p[4]; p.state = (const int*)op->p[5]; p.hist = (float*)op->p[6]; p.out = (float*)op->p[7];
    p.extra = (float*)op->p[8];
    p.numel = (size_t)(uint32_t)op->i[0] | ((size_t)(uint32_t)op->i[1] << 32);
    p.mode = op->i[2]; p.T = op->i[3]; p.s_imm = op->i[4]; p.fix = op->i[5];
    p.s_mul = op->i[6] > 0 ? op->i[6] : 1; p.s_off = op->i[7];
    p.cfg = op->f[0];
    AED_REQUIRE(p.xts && p.v_u && p.coef && p.hist, "sa_step: null pointer");
    AED_REQUIRE(p.mode == 0 ? (p.zs != nullptr) : (p.out != nullptr), "sa_step: missing zs (invert) / out (reverse)");
    hipLaunchKernelGGL(sa_step_kernel, dim3(sa_grid(p.numel)), dim3(256), 0, s, p);
    AED_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int aed_sa_get_zs_from_xts(const float* xt, float* xtm1, const float* v_u, const float* v_c, float cfg_scalar,
                                      const float* coef_host, float* hist, int numerical_fix, float* z,
                                      float* extra_out, int64_t numel, void* stream) {
    AED_REQUIRE(xt && xtm1 && v_u && coef_host && hist && z, "aed_sa_get_zs_from_xts: null pointer");
    SAStep p = {};
    p.xts = const_cast<float*>(xt); p.xtm1 = xtm1; p.zs = z; p.v_u = v_u; p.v_c = v_c; p.hist = hist; p.extra = extra_out;
    p.numel = (size_t)numel; p.mode = 0; p.fix = numerical_fix; p.explicit_ptrs = 1; p.s_mul = 1; p.cfg = cfg_scalar;
    for (int k = 0; k < AED_SA_COEF_STRIDE; ++k) p.c[k] = coef_host[k];
    hipLaunchKernelGGL(sa_step_kernel, dim3(sa_grid(p.numel)), dim3(256), 0, (hipStream_t)stream, p);
    AED_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int aed_sa_reverse_step_with_custom_noise(const float* xt, const float* v_u, const float* v_c,
                                                     float cfg_scalar, const float* coef_host, float* hist,
                                                     const float* z, float* prev_out, int64_t numel, void* stream) {
    AED_REQUIRE(xt && v_u && coef_host && hist && prev_out, "aed_sa_reverse_step_with_custom_noise: null pointer");
    SAStep p = {};
    p.xts = const_cast<float*>(xt); p.zs = const_cast<float*>(z); p.v_u = v_u; p.v_c = v_c; p.hist = hist; p.out = prev_out;
    p.numel = (size_t)numel; p.mode = 1; p.explicit_ptrs = 1; p.s_mul = 1; p.cfg = cfg_scalar;
    for (int k = 0; k < AED_SA_COEF_STRIDE; ++k) p.c[k] = coef_host[k];
    hipLaunchKernelGGL(sa_step_kernel, dim3(sa_grid(p.numel)), dim3(256), 0, (hipStream_t)stream, p);
    AED_CHECK_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------ rotary embedding (q and k in place)
// x: [M, ld] rows = (batch item, position); `nsec` sections of H heads x D features start at column sec*sec_stride
// (q at 0, k at C inside the fused qkv buffer).  The first R features of every head are rotated:
//   (re, im) = (x[j], x[R/2 + j])  ->  (re*cos - im*sin, im*cos + re*sin)   with cos/sin[pos, j]   (table [N, R/2])
// which is diffusers' apply_rotary_emb(use_real=True, use_real_unbind_dim=-2) on the table of
// get_1d_rotary_pos_embed(repeat_interleave_real=False) (both halves of that table are equal).
__global__ __launch_bounds__(256) void rotary_kernel(float* __restrict__ x, const float* __restrict__ ct,
                                                      const float* __restrict__ st, int M, int N, int H, int D, int R,
                                                      int ld, int nsec, int sec_stride) {
    const int hr = R >> 1;
    const size_t total = (size_t)M * nsec * H * hr;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const int j = (int)(e % hr);
        size_t r = e / hr;
        const int h = (int)(r % H); r /= H;
        const int sec = (int)(r % nsec);
        const int m = (int)(r / nsec);
        const int pos = m % N;
        float* px = x + (size_t)m * ld + (size_t)sec * sec_stride + h * D;
        const float cs = ct[(size_t)pos * hr + j], sn = st[(size_t)pos * hr + j];
        const float re = px[j], im = px[hr + j];
        px[j] = re * cs + (-im) * sn;
        px[hr + j] = im * cs + re * sn;
    }
}
// slots: p0=x p1=cos[N,R/2] p2=sin[N,R/2] ; i0=M i1=N (positions per batch item) i2=H i3=D i4=R i5=ld i6=nsec i7=sec_stride
int launch_rotary(const aed_op* op, hipStream_t s) {
    const int32_t* i = op->i;
    AED_REQUIRE(op->p[0] && op->p[1] && op->p[2], "rotary: null pointer");
    AED_REQUIRE(i[4] % 2 == 0 && i[4] <= i[3] && i[1] > 0 && i[6] >= 1, "rotary: bad geometry");
    hipLaunchKernelGGL(rotary_kernel, dim3(sa_grid((size_t)i[0] * i[6] * i[2] * (i[4] / 2))), dim3(256), 0, s,
                       (float*)op->p[0], (const float*)op->p[1], (const float*)op->p[2], i[0], i[1], i[2], i[3], i[4], i[5],
                       i[6], i[7]);
    AED_CHECK_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------ Snake1d
// y[r, c] = x[r, c] + inv_b[c] * sin(a[c] * x[r, c])^2 ; a = exp(alpha), inv_b = 1 / (exp(beta) + 1e-9) (host, once)
__global__ __launch_bounds__(256) void snake_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                     const float* __restrict__ a, const float* __restrict__ ib, size_t rows,
                                                     int C, int ldx, int ldy) {
    const int q = C >> 2;
    const size_t total = rows * q;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const size_t r = e / q;
        const int c = (int)(e - r * q) * 4;
        const float4 v = *reinterpret_cast<const float4*>(x + r * ldx + c);
        const float4 av = *reinterpret_cast<const float4*>(a + c);
        const float4 bv = *reinterpret_cast<const float4*>(ib + c);
        float4 o;
        float sx;
        sx = sinf(av.x * v.x); o.x = v.x + bv.x * (sx * sx);
        sx = sinf(av.y * v.y); o.y = v.y + bv.y * (sx * sx);
        sx = sinf(av.z * v.z); o.z = v.z + bv.z * (sx * sx);
        sx = sinf(av.w * v.w); o.w = v.w + bv.w * (sx * sx);
        *reinterpret_cast<float4*>(y + r * ldy + c) = o;
    }
}
// slots: p0=x p1=y p2=a[C] p3=inv_b[C] ; i0,i1=rows lo/hi i2=C i3=ldx i4=ldy
int launch_snake(const aed_op* op, hipStream_t s) {
    const int32_t* i = op->i;
    const size_t rows = (size_t)(uint32_t)i[0] | ((size_t)(uint32_t)i[1] << 32);
    AED_REQUIRE(op->p[0] && op->p[1] && op->p[2] && op->p[3], "snake: null pointer");
    AED_REQUIRE(i[2] % 4 == 0 && i[3] % 4 == 0 && i[4] % 4 == 0, "snake: channel count and row strides must be multiples of 4");
    hipLaunchKernelGGL(snake_kernel, dim3(sa_grid(rows * (i[2] / 4))), dim3(256), 0, s, (const float*)op->p[0],
                       (float*)op->p[1], (const float*)op->p[2], (const float*)op->p[3], rows, i[2], i[3], i[4]);
    AED_CHECK_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------ posterior sample
// moments: [rows, 2*C] (mean | scale) ; out[r, c] = mean + (softplus(scale) + 1e-4) * noise[r, c]
__global__ __launch_bounds__(256) void gauss_sample_kernel(const float* __restrict__ mom, const float* __restrict__ noise,
                                                            float* __restrict__ out, size_t rows, int C, int ldm) {
    const size_t total = rows * C;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const size_t r = e / C;
        const int c = (int)(e - r * C);
        const float mean = mom[r * ldm + c], sc = mom[r * ldm + C + c];
        const float sp = sc > 20.0f ? sc : log1pf(expf(sc));          // torch.nn.functional.softplus (threshold 20)
        out[e] = mean + (sp + 1e-4f) * noise[e];
    }
}
// slots: p0=moments p1=noise p2=out ; i0,i1=rows lo/hi i2=C i3=ld of moments
int launch_gauss_sample(const aed_op* op, hipStream_t s) {
    const int32_t* i = op->i;
    const size_t rows = (size_t)(uint32_t)i[0] | ((size_t)(uint32_t)i[1] << 32);
    AED_REQUIRE(op->p[0] && op->p[1] && op->p[2], "gauss_sample: null pointer");
    hipLaunchKernelGGL(gauss_sample_kernel, dim3(sa_grid(rows * i[2])), dim3(256), 0, s, (const float*)op->p[0],
                       (const float*)op->p[1], (float*)op->p[2], rows, i[2], i[3]);
    AED_CHECK_HIP(hipGetLastError());
    return 0;
}
