// cg_params.h -- kernel-side view of an AED_OP_CONV_GEMM record, shared by the two GEMM families:
//   conv_gemm.hip : LDS-staged block-tile kernels (throughput regime: inversion batches, VAE, vocoder)
//   lin_gemm.hip  : wave-split-K kernels with wave-private LDS staging (latency regime: the U-Net batch-2 edit loop)
#pragma once
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
    long long* dbg;          // optional in-kernel timeline (block 0, lane 0): s_memtime stamps
    const float* A2;         // second A source (channels [C1, Cin) of every tap): the skip half of an up-block concat
    int diag;                // A/B switches of conv_gemm_x6 (op flag bit 15 = 0x8000 -> bit 0: always the general epilogue)
    int gm;                  // tile order of conv_gemm_x6 (round 6): <= 1 = n fastest inside an XCD's range of tile ids; g > 1 = groups
                             // of g row panels, m fastest inside a group -- the workgroups that run together on an XCD then cover
                             // g A panels x (resident / g) W tiles instead of ~1 panel x ALL W tiles (wide Linears: W larger
                             // than the 4 MB L2 was re-streamed for every row panel; profiles/r06_tile_order.md)
    int M, N, K;
    int lda, ldc, ldr, ld_rv;
    int IH, IW, OH, OW, Cin;
    int KH, KW, stride, pad_h, pad_w, dil_h, dil_w, up;
    int a_bs;
    int vIH, vIW;            // virtual (nearest-upsampled) input grid actually convolved: <= IH<<up, IW<<up
    int o_mul, o_add, o_len, out_bs;
    int in_act, out_act, accumulate, ksplit;
    int rpb;                 // output rows per batch item = OH*OW
    int nchunks;             // ceil(K/BKT)
    // K traversal of the LDS-staged kernels (conv_gemm.hip, conv_gemm_x6.hip).  A chunk is (tap, channel offset); both operands
    // use the same map, so the order only decides WHEN a byte is fetched.  kgroup = channels per group: the chunks walk
    // [group][tap][chunk within group].  kgroup = Cin is the tap-major order of rounds 1-3 (all channels of tap 0, then tap 1 ...):
    // the 9 taps of a 3x3 conv re-read the same activation lines Cin/32 chunks apart and, at the inversion's batch, out of
    // the 4 MB L2 (measured 2.08x the algorithmic bytes).  kgroup = 32 (one 128-byte line per pixel; 64 for 64-wide chunks)
    // consumes all taps of a line group back to back: the re-reads hit the L2.
    int kgroup;
    float in_slope, out_p, out_div;
    int ln_mode;             // 1: A rows are LayerNorm inputs; W has gamma folded in, rowvec = sum_k W'[n,k], bias = W.beta (+bias)
    float ln_eps;
    int C1, lda2, a_bs2;     // two-source A: channel c < C1 comes from A (lda, a_bs), c >= C1 from A2 (lda2, a_bs2) at c - C1
    int late_epilogue;       // A/B switch (op flag 2): fetch bias / residual after the reduction instead of up front
    // per-batch-item operands (cross-attention folded into two skinny GEMMs, see unet.py): W, bias and rowvec advance by
    // these element strides with the batch item of the output row; vec_ld = stride between consecutive n in bias/rowvec
    int w_bs, vec_bs, vec_ld;
    // grouped softmax epilogue: out = softmax over each run of sm_group consecutive columns of (val * sm_scale + kbias[b][n % sm_group])
    int sm_group;
    float sm_scale;
    const float* kbias;
    int geglu;               // W rows are packed [32 value | 32 gate] per 32 output features; 1: out = value * gelu(gate)
                             // (GEGLU), 2: out = value * silu(gate) (SwiGLU, the Stable Audio DiT's feed-forward)
    // conv_gemm_f8.hip with pre-quantised weights (op flag bit 7): W as MX-FP8 -- e4m3 bytes [N][K] and one e8m0 scale byte per
    // 32 k [N][K/32] (aed_mx_quantize_rows); null otherwise
    const unsigned char* Wq;
    const unsigned char* Wsc;
};

__device__ __forceinline__ float in_transform(float v, int act, float slope) {
    if (act == AED_ACT_SILU) return v / (1.0f + __expf(-v));
    if (act == AED_ACT_LEAKY) return v > 0.0f ? v : v * slope;
    return v;
}

__device__ __forceinline__ float gelu_exact(float g) { return 0.5f * g * (1.0f + erff(g * 0.70710678118654752440f)); }
__device__ __forceinline__ float glu_gate(float g, int kind) { return kind == 2 ? g / (1.0f + expf(-g)) : gelu_exact(g); }

// general (one element per call) epilogue: bias, per-batch row vector, residual, activation, accumulate, row scatter
__device__ __forceinline__ void store_out(const CGParams& p, int m, int n, float v) {
    int b = m / p.rpb;
    int q = m - b * p.rpb;
    int o = q * p.o_mul + p.o_add;
    if ((unsigned)o >= (unsigned)p.o_len) return;
    size_t row = (size_t)b * p.out_bs + o;
    if (p.bias) v += p.bias[n];
    if (p.rowvec && !p.ln_mode) v += p.rowvec[(size_t)b * p.ld_rv + n];
    if (p.res) v += p.res[row * p.ldr + n];
    v = aed_apply_act(v, p.out_act, p.out_p);
    float* dst = p.C + row * p.ldc + n;
    if (p.accumulate == 1) v += *dst;
    else if (p.accumulate == 2) v = (*dst + v) / p.out_div;
    *dst = v;
}

int cg_fill_params(const aed_op* op, CGParams& p, int bkt);
int launch_lin_gemm(const CGParams& p, int cfg, hipStream_t s);
