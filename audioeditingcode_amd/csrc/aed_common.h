// Shared declarations for libaed.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/aed.h"

void aed_set_error(const char* fmt, ...);

#define AED_CHECK_HIP(expr)                                                                   \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            aed_set_error("%s:%d %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            return 1;                                                                         \
        }                                                                                     \
    } while (0)

#define AED_REQUIRE(cond, ...)                 \
    do {                                       \
        if (!(cond)) {                         \
            aed_set_error(__VA_ARGS__);        \
            return 2;                          \
        }                                      \
    } while (0)

static inline int aed_cdiv(int a, int b) { return (a + b - 1) / b; }

// launchers (one per opcode family), defined in the .hip files
int launch_conv_gemm(const aed_op* op, hipStream_t s);
int launch_splitk_reduce(const aed_op* op, hipStream_t s);
int launch_splitk_reduce_slices(const aed_op* op, int slices, hipStream_t s);   // slices > 0: the producer's slice count
int launch_gn_stats(const aed_op* op, hipStream_t s);
int launch_gn_apply(const aed_op* op, hipStream_t s);
int launch_gn_scale_shift(const aed_op* op, hipStream_t s);
int launch_gn_small(const aed_op* op, hipStream_t s);
int launch_attention(const aed_op* op, hipStream_t s);
int launch_copy2d(const aed_op* op, hipStream_t s);
int launch_time_embed(const aed_op* op, hipStream_t s);
int launch_softmax_rows(const aed_op* op, hipStream_t s);
int launch_transpose(const aed_op* op, hipStream_t s);
int launch_axpby(const aed_op* op, hipStream_t s);
int launch_invert_step(const aed_op* op, hipStream_t s);
int launch_reverse_step(const aed_op* op, hipStream_t s);
int launch_ddim_step(const aed_op* op, hipStream_t s);
int launch_advance(const aed_op* op, hipStream_t s);
int launch_reflect_pad(const aed_op* op, hipStream_t s);
int launch_magnitude(const aed_op* op, hipStream_t s);
int launch_layout(const aed_op* op, hipStream_t s);
int launch_xattn_fold(const aed_op* op, hipStream_t s);
int launch_rotary(const aed_op* op, hipStream_t s);
int launch_snake(const aed_op* op, hipStream_t s);
int launch_sa_step(const aed_op* op, hipStream_t s);
int launch_gauss_sample(const aed_op* op, hipStream_t s);

int aed_num_cus();

// ---- device helpers -------------------------------------------------------------------
__device__ __forceinline__ float aed_silu(float x) { return x / (1.0f + __expf(-x)); }

__device__ __forceinline__ float aed_apply_act(float v, int act, float p) {
    switch (act) {
        case AED_ACT_SILU: return v / (1.0f + expf(-v));
        case AED_ACT_LEAKY: return v > 0.0f ? v : v * p;
        case AED_ACT_TANH: return tanhf(v);
        case AED_ACT_LOGCLAMP: return logf(fmaxf(v, p));
        default: return v;
    }
}
