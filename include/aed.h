/*
 * aed.h -- C ABI of libaed.so, the MI355X (gfx950) native engine underneath the
 * reference's models.py wrapper API (HilaManor/AudioEditingCode).
 *
 * The reference has NO native boundary: its hot path crosses from the editing loops into
 * torch/diffusers through the duck-typed Python class PipelineWrapper
 * (/root/reference/code/models.py:14-393 and the AudioLDM/AudioLDM2/TANGO subclasses).
 * This header is the boundary a maintainer would bind underneath that class (ctypes stub in
 * INTEGRATION.md).  Conventions:
 *   - extern "C", plain pointers and sizes, no torch types;
 *   - every pointer is a DEVICE pointer borrowed from the caller (never freed here) unless
 *     the name says _host;
 *   - tensors are fp32, contiguous, CHANNELS-LAST: feature maps [B,H,W,C] (== token matrix
 *     [B*H*W, C]), sequences [B,L,C]; weights [N,K] with K=(ky,kx,ci) for convolutions;
 *   - every entry point takes a hipStream_t (passed as void*) and is asynchronous on it;
 *   - return 0 = OK, non-zero = error with a message in aed_last_error().
 *
 * Two levels:
 *   (1) path-level entry points named after the reference methods they replace
 *       (aed_get_zs_from_xts, aed_reverse_step_with_custom_noise, ...);
 *   (2) the op tape: a model forward (U-Net / VAE / vocoder / STFT) is a flat array of
 *       aed_op records built once by the host and executed natively by aed_tape_run(),
 *       optionally captured into a hipGraph (aed_graph_*).
 */
#ifndef AED_H
#define AED_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AED_VERSION 4

/* ----------------------------------------------------------------------------------------
 * op tape
 * -------------------------------------------------------------------------------------- */
enum aed_opcode {
    AED_OP_NOP = 0,
    AED_OP_CONV_GEMM = 1,     /* implicit-GEMM conv / linear on fp32 MFMA (K5,K6,K11,K2,K3); optional two-source A
                                 (skip concat never materialised), fused LayerNorm, fused GEGLU gate (K8).
                                 flags bit 0: in-kernel timeline into p[7]; bit 1: late epilogue fetch (lin_gemm A/B);
                                 bit 2 (the product's arithmetic since round 4; tapes built under tape.arith_mode("bf16x6")):
                                 contract on split-bf16 MFMAs -- every fp32 operand is cut exactly into three bf16 pieces in the
                                 loader and the six piece products of relative size >= 2^-16 are accumulated in fp32
                                 (csrc/conv_gemm_x6.hip; as close to fp64 as the fp32 chain: tools/bf16_split_study.py,
                                 profiles/r03_x6_gemm.md; shapes that kernel does not take run the fp32 path);
                                 bit 3: with bit 2, interleave hints in the main loop (tapes set it; off = A/B);
                                 bit 4: with bit 2, three-term DIAGNOSTIC arithmetic (~4e-6 rel error);
                                 bit 8 (256): with bit 2 on the 512-thread tiles 8 / 9, 32-wide K chunks -- two bf16 MFMA k-blocks
                                 per LDS stage and barrier (needs 32 | Cin; other records ignore it);
                                 bit 10 (1024): with bit 2, keep the n-fastest tile order of rounds 1-5 (A/B; default since round 6:
                                 groups of row panels, m fastest inside a group -- csrc/conv_gemm_x6.hip); bits 11-13: forced
                                 group height 2^v (sweeps); bit 15 (0x8000): always the general epilogue (A/B of the simple-rows epilogue);
                                 bits 16-17: the stream this record runs on is masked to (all CUs) >> v compute units (sizes the
                                 tile groups; tapes built under tape.tile_regime set it);
                                 bit 6 (EXPERIMENT, tapes built under tape.arith_mode("fp8")): contract on the MX-FP8 matrix
                                 cores (csrc/conv_gemm_f8.hip: OCP microscaling e4m3, one e8m0 scale per 32 k of a row,
                                 quantised in the loader, v_mfma_scale_f32_32x32x64_f8f6f4, fp32 accumulate).  NOT a parity
                                 path (a few 1e-2 relative per GEMM); ops that kernel does not take run bits 2|3;
                                 bit 7: with bit 6, the weights are pre-quantised: p[7] = e4m3 bytes [N][K], p[9] = e8m0
                                 scale bytes [N][K/32] (aed_mx_quantize_rows); p[1] stays the fp32 weights (fallback)   */
    AED_OP_GN_STATS = 2,      /* GroupNorm partial sums (K4)                                   */
    AED_OP_GN_APPLY = 3,      /* GroupNorm normalise + affine (+SiLU) (K4)                     */
    AED_OP_LAYERNORM = 4,     /* RETIRED in v4 (returns an error): LayerNorm is fused into the consuming GEMM (K8)  */
    AED_OP_ATTENTION = 5,     /* softmax(QK^T*scale + bias)V, flash-style, fp32 MFMA (K7).  flags bit 2 (tapes built under
                                 tape.arith_mode("bf16x6")): in the throughput regime (every wave its own query tile) both
                                 contractions run on split-bf16 MFMAs (csrc/attention_x6.hip: K, V, Q and the probabilities cut
                                 exactly into three bf16 pieces, six piece products, fp32 accumulate and fp32 softmax)       */
    AED_OP_GEGLU = 6,         /* RETIRED in v4 (returns an error): the gate rides in the FF1 epilogue (K8)         */
    AED_OP_COPY2D = 7,        /* strided 2-D copy (skip concat, h-space tap/replace) (K10)     */
    AED_OP_TIME_EMBED = 8,    /* sinusoidal timestep embedding (K9)                            */
    AED_OP_SOFTMAX_ROWS = 9,  /* row softmax (VAE single-head attention)                       */
    AED_OP_TRANSPOSE = 10,    /* batched 2-D transpose                                         */
    AED_OP_AXPBY = 11,        /* y = a*x + b*y elementwise (h-space add, sample_xts)           */
    AED_OP_INVERT_STEP = 12,  /* CFG + get_zs_from_xts (K1; models.py:85-117)                  */
    AED_OP_REVERSE_STEP = 13, /* CFG + reverse_step_with_custom_noise (K1; models.py:119-158)  */
    AED_OP_DDIM_STEP = 14,    /* CFG + DDIM next_step / scheduler.step(eta=0)                  */
    AED_OP_ADVANCE = 15,      /* device step counter += 1                                      */
    AED_OP_REFLECT_PAD = 16,  /* 1-D reflect pad (stft.py:60-65)                               */
    AED_OP_MAGNITUDE = 17,    /* sqrt(re^2+im^2) (stft.py:78)                                  */
    AED_OP_NCHW_TO_NHWC = 18, /* boundary layout change                                        */
    AED_OP_NHWC_TO_NCHW = 19,
    AED_OP_SPLITK_REDUCE = 20,
    AED_OP_GN_SCALE_SHIFT = 21, /* GroupNorm stats -> per-(batch,channel) scale/shift vectors [B,2,C]         */
    AED_OP_GN_SMALL = 22,     /* single-launch GroupNorm(+SiLU) for small feature maps (K4)       */
    AED_OP_XATTN_FOLD = 23,   /* per-prompt operands of the folded cross-attention: G' = k_h.(gamma o Wq_h), its two
                                 LayerNorm-fold vectors, VO^T = (v_h.Wo_h^T)^T -- once per prompt, not per step      */
    AED_OP_ROTARY = 24,       /* partial rotary embedding of q and k in place (Stable Audio DiT self-attention)           */
    AED_OP_SNAKE = 25,        /* Snake1d activation of the Oobleck VAE                                                     */
    AED_OP_SA_STEP = 26,      /* CFG + StableAudWrapper.get_zs_from_xts / reverse_step_with_custom_noise (SDE-DPM-Solver++
                                 order 1 / 2, history on the device; models.py:1209-1271, :1282-1329)                      */
    AED_OP_GAUSS_SAMPLE = 27, /* mean + (softplus(scale) + 1e-4) * noise (Oobleck posterior sample, models.py:1132-1133)    */
    AED_OP_COUNT
};

/* One record of the tape.  Which slots an opcode reads is documented next to its launcher in
 * audioeditingcode_amd/csrc (and mirrored by audioeditingcode_amd/tape.py). */
typedef struct aed_op {
    int32_t code;
    int32_t flags;
    int32_t i[40];
    float   f[8];
    void*   p[10];
} aed_op;

/* activation / transform codes used in aed_op slots */
enum aed_act { AED_ACT_NONE = 0, AED_ACT_SILU = 1, AED_ACT_LEAKY = 2, AED_ACT_TANH = 3, AED_ACT_LOGCLAMP = 4 };

int         aed_version(void);
const char* aed_last_error(void);
/* number of CUs, LDS bytes per CU and gcn arch name of the current device */
int         aed_device_info(int* cu_count, int* lds_bytes, char* arch, int arch_len);

/* Execute one op / a tape of n ops on `stream` (asynchronous). */
int aed_launch(const aed_op* op, void* stream);
int aed_tape_run(const aed_op* ops, int n, void* stream);
/* Timed variant: records a hipEvent pair around every op on `stream`, synchronises, and
 * writes per-op milliseconds into ms_host[n] (host pointer).  Used by bench.py's roofline leg. */
int aed_tape_profile(const aed_op* ops, int n, void* stream, float* ms_host);

/* hipGraph capture of everything launched on `stream` between begin and end. */
int aed_graph_begin(void* stream);
int aed_graph_end(void* stream, void** graph_exec_out);
int aed_graph_launch(void* graph_exec, void* stream);
int aed_graph_destroy(void* graph_exec);

/* Streams restricted to a subset of the chip's compute units (two-clip pipeline: the latency-bound edit loop of clip i
 * and the throughput-bound inversion of clip i+1 run on DISJOINT CU sets, so neither queues behind the other's
 * workgroups).  mask_words: n_words x 32 bits, bit k = CU k in the driver's enumeration (consecutive bits rotate over the
 * 8 XCDs, so a contiguous bit range is spread evenly over them).  priority: 0 normal, <0 higher (HIP convention); used
 * only when n_words == 0 (an unmasked stream).  The caller destroys the stream with aed_stream_destroy. */
int aed_stream_create_cu_mask(void** stream_out, const uint32_t* mask_words, int n_words, int priority);
int aed_stream_destroy(void* stream);
/* Diagnostic: n_blocks one-wave workgroups, each idling ~spin_clocks shader cycles, write their {HW_ID, XCC_ID} hardware
 * registers to out_dev[2*n_blocks]: which physical CUs (xcc, se, sh, cu) a stream's work lands on (profiles/ evidence
 * for the CU partition). */
int aed_cu_census(uint32_t* out_dev, int n_blocks, int spin_clocks, void* stream);

/* ----------------------------------------------------------------------------------------
 * tape images: a compiled model for hosts without the Python graph compiler
 * --------------------------------------------------------------------------------------
 * audioeditingcode_amd/image.py writes a model's op tapes, every buffer they reference (weights, tables, activations, inputs,
 * outputs) and a snapshot of those buffers into ONE relocatable file.  aed_image_load() puts the arena on the device and
 * relocates the ops; a host then fills the named inputs, runs a named program (= aed_tape_run on the image's ops) and reads
 * the named outputs -- e.g. for a U-Net image: copy_in "x_in" / "ehs0" / "ehs1" / "bias1" / "timesteps", run "context" once
 * per prompt, run "forward" per step, copy_out "eps".  (This is the model-level boundary SURVEY 8(b) sketched as
 * aed_create / aed_unet_forward / aed_vae_encode / ...: one loader and one runner instead of one entry point per model.)
 * flags bit 0: keep the arena in HOST memory (inspection and tests; aed_image_run refuses such an image).             */
int aed_image_load(const char* path, int flags, void** image_out);
int aed_image_free(void* image);
int aed_image_run(void* image, const char* program, void* stream);
int aed_image_program(void* image, const char* program, const aed_op** ops_out, int* n_out);   /* the relocated ops */
int aed_image_buffer(void* image, const char* name, void** ptr_out, uint64_t* nbytes_out);
int aed_image_copy_in(void* image, const char* name, const void* host, uint64_t nbytes, void* stream);     /* synchronous */
int aed_image_copy_out(void* image, const char* name, void* host, uint64_t nbytes, void* stream);          /* synchronous */

/* HIP-event helpers so hosts without a HIP binding can time a stream region. */
int aed_event_create(void** ev_out);
int aed_event_record(void* ev, void* stream);
int aed_event_elapsed_ms(void* ev_start, void* ev_stop, float* ms_out);   /* synchronises on ev_stop */
int aed_event_destroy(void* ev);

/* ----------------------------------------------------------------------------------------
 * path-level entry points (reference method each one replaces)
 * -------------------------------------------------------------------------------------- */

/* Per-step scheduler coefficients, computed on the host in fp32 with the reference's
 * expression order (models.py:91-113, :124-150) so that the device arithmetic is bit-exact:
 *   c[0]=sqrt(1-abar_t)  c[1]=sqrt(abar_t)  c[2]=sqrt(abar_prev)
 *   c[3]=sqrt(1-abar_prev-eta*var)          c[4]=eta*sqrt(var)     c[5..7] reserved        */
#define AED_COEF_STRIDE 8

/* PipelineWrapper.get_zs_from_xts (models.py:85-117) fused with the CFG combine of
 * inversion_utils.py:97-102.  eps_c may be NULL (empty source prompt, inversion_utils.py:86).
 * cfg: per-element guidance tensor [P,numel] (inversion_utils.py:29-51) or NULL -> cfg_scalar.
 * Writes z[numel] and, when numerical_fix, overwrites xtm1 in place with mu + sigma*z.      */
int aed_get_zs_from_xts(const float* xt, float* xtm1, const float* eps_u, const float* eps_c,
                        const float* cfg, float cfg_scalar, int n_prompts, const float* coef_host,
                        int v_prediction, int numerical_fix, float* z, float* noise_pred_out,
                        int64_t numel, void* stream);

/* PipelineWrapper.reverse_step_with_custom_noise (models.py:119-158) fused with the CFG
 * combine of inversion_utils.py:276-281.  z may be NULL when eta == 0.                      */
int aed_reverse_step_with_custom_noise(const float* xt, const float* eps_u, const float* eps_c,
                                       const float* cfg, float cfg_scalar, int n_prompts,
                                       const float* coef_host, int v_prediction, const float* z,
                                       float* prev_out, int64_t numel, void* stream);

/* PipelineWrapper.sample_xts_from_x0 inner statement (models.py:81):
 * out = x0*sqrt_abar + noise*sqrt_1m_abar, for n_t rows (noise drawn by the host RNG).      */
int aed_sample_xts_from_x0(const float* x0, const float* noise, const float* sqrt_abar,
                           const float* sqrt_1m_abar, float* xts_out, int n_t, int64_t numel,
                           void* stream);

/* EXPERIMENT (fp8 path of BASELINE config 5; no reference counterpart -- the reference's DiT call, models.py:1331-1354, is fp32):
 * rows of an fp32 matrix [rows][K] (K % 32 == 0) -> OCP MX-FP8: e4m3 bytes q[rows][K] and one e8m0 scale byte per 32 k,
 * scales[rows][K/32]; scale = 2^(floor(log2(block amax)) - 8), element = RNE_e4m3(clamp(x / scale, +-448)).  Weights quantised
 * once with this feed AED_OP_CONV_GEMM records with flag bits 6|7 (p[7] = q, p[9] = scales); csrc/conv_gemm_f8.hip.          */
int aed_mx_quantize_rows(const float* src, void* q, void* scales, long long rows, int K, void* stream);

/* ----------------------------------------------------------------------------------------
 * Stable Audio Open (StableAudWrapper, models.py:1051-1354)
 * -------------------------------------------------------------------------------------- */

/* Per-step coefficients of the CosineDPMSolver++ SDE update, computed on the host in fp32 with the expression order of
 * models.py:1238-1255 / the scheduler's update functions (sigma_s = sigmas[i], sigma_t = sigmas[i+1], h = ln sigma_s - ln sigma_t):
 *   c[0]=c_in = 1/sqrt(sigma_s^2+sd^2) (scale_model_input)   c[1]=c_skip  c[2]=c_out (v-prediction -> data prediction)
 *   c[3]=sigma_t/sigma_s*exp(-h)   c[4]=1-exp(-2h)   c[5]=sigma_t*sqrt(1-exp(-2h))   c[6]=1/r0 (r0 = h_prev/h)
 *   c[7]=solver order of this step (1 or 2)   c[8]=1 when z is defined as 0 (final step, sigma_t = 0)
 *   c[9]=2*pi*timestep (the DiT's Fourier-feature argument)   c[10..11] reserved                                        */
#define AED_SA_COEF_STRIDE 12

/* StableAudWrapper.get_zs_from_xts (models.py:1209-1271) fused with the CFG combine of inversion_utils.py:97-102 (one
 * prompt).  `hist` holds the previous step's data prediction on entry (the scheduler's model_outputs[-2]; ignored when
 * c[7] == 1) and this step's on exit.  Writes z, overwrites xtm1 when numerical_fix, and copies the previous data
 * prediction to extra_out (the reference's third return value) when it is not NULL.  v_c may be NULL.                   */
int aed_sa_get_zs_from_xts(const float* xt, float* xtm1, const float* v_u, const float* v_c, float cfg_scalar,
                           const float* coef_host, float* hist, int numerical_fix, float* z, float* extra_out,
                           int64_t numel, void* stream);

/* StableAudWrapper.reverse_step_with_custom_noise (models.py:1282-1329) fused with the CFG combine of
 * inversion_utils.py:276-281.  z may be NULL (treated as zero noise).                                                   */
int aed_sa_reverse_step_with_custom_noise(const float* xt, const float* v_u, const float* v_c, float cfg_scalar,
                                          const float* coef_host, float* hist, const float* z, float* prev_out,
                                          int64_t numel, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AED_H */
