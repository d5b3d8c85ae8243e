// api.hip -- C-ABI surface: error state, op dispatch, tape executor, hipGraph + event helpers.
#include <stdarg.h>
#include <string.h>
#include <vector>
#include "aed_common.h"

static thread_local char g_err[512] = "";
void aed_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

static int g_cus = 0;
int aed_num_cus() {
    if (g_cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
            g_cus = prop.multiProcessorCount;
        if (g_cus <= 0) g_cus = 256;
    }
    return g_cus;
}

// Where did a stream's workgroups run?  Each block idles ~spin clocks (so the grid spreads over every CU the stream may
// use instead of recycling the first free one) and records its HW_ID and XCC_ID registers.
__global__ void cu_census_kernel(uint32_t* out, int spin) {
    long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < spin) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));       // HW_REG_HW_ID
        out[2 * blockIdx.x + 1] = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));  // HW_REG_XCC_ID
    }
}

typedef int (*launcher_t)(const aed_op*, hipStream_t);
static int launch_nop(const aed_op*, hipStream_t) { return 0; }
// opcodes whose standalone kernel was retired (ABI v4): LayerNorm lives in the consuming GEMM (fused row statistics),
// the GEGLU / SwiGLU gate in the FF1 epilogue.  The enum slots stay so that older tapes fail loudly instead of shifting.
static int launch_retired(const aed_op* op, hipStream_t) {
    aed_set_error("opcode %d was retired in ABI v4 (fused into conv_gemm: ln_mode / geglu slots)", op->code);
    return 3;
}
// CONV_GEMM, flag bit 2: contraction on split-bf16 MFMAs (conv_gemm_x6.hip; experimental, see include/aed.h)
int launch_conv_gemm_x6(const aed_op* op, hipStream_t s);
// flag bit 6: EXPERIMENT -- contraction on the MX-FP8 matrix cores (conv_gemm_f8.hip; not a parity path, see include/aed.h)
int launch_conv_gemm_f8(const aed_op* op, hipStream_t s);
static int launch_conv_gemm_any(const aed_op* op, hipStream_t s) {
    if (op->flags & 64) return launch_conv_gemm_f8(op, s);
    return (op->flags & 4) ? launch_conv_gemm_x6(op, s) : launch_conv_gemm(op, s);
}
static launcher_t g_table[AED_OP_COUNT] = {
    launch_nop,           // NOP
    launch_conv_gemm_any, // CONV_GEMM
    launch_gn_stats, launch_gn_apply, launch_retired, launch_attention, launch_retired, launch_copy2d,
    launch_time_embed, launch_softmax_rows, launch_transpose, launch_axpby, launch_invert_step,
    launch_reverse_step, launch_ddim_step, launch_advance, launch_reflect_pad, launch_magnitude,
    launch_layout, launch_layout, launch_splitk_reduce, launch_gn_scale_shift, launch_gn_small, launch_xattn_fold,
    launch_rotary, launch_snake, launch_sa_step, launch_gauss_sample,
};

extern "C" {

int aed_version(void) { return AED_VERSION; }
const char* aed_last_error(void) { return g_err; }

int aed_device_info(int* cu_count, int* lds_bytes, char* arch, int arch_len) {
    int dev = 0;
    hipDeviceProp_t prop;
    AED_CHECK_HIP(hipGetDevice(&dev));
    AED_CHECK_HIP(hipGetDeviceProperties(&prop, dev));
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (lds_bytes) *lds_bytes = (int)prop.maxSharedMemoryPerMultiProcessor;
    if (arch && arch_len > 0) {
        strncpy(arch, prop.gcnArchName, arch_len - 1);
        arch[arch_len - 1] = 0;
    }
    return 0;
}

int aed_launch(const aed_op* op, void* stream) {
    AED_REQUIRE(op != nullptr, "aed_launch: null op");
    AED_REQUIRE(op->code >= 0 && op->code < AED_OP_COUNT, "aed_launch: bad opcode %d", op->code);
    return g_table[op->code](op, (hipStream_t)stream);
}

int aed_tape_run(const aed_op* ops, int n, void* stream) {
    AED_REQUIRE(ops != nullptr || n == 0, "aed_tape_run: null tape");
    for (int k = 0; k < n; ++k) {
        const aed_op* op = ops + k;
        if (op->code < 0 || op->code >= AED_OP_COUNT) {
            aed_set_error("aed_tape_run: bad opcode %d at op %d", op->code, k);
            return 2;
        }
        int rc = g_table[op->code](op, (hipStream_t)stream);
        if (rc) {
            char tmp[400];
            strncpy(tmp, g_err, sizeof(tmp) - 1);
            tmp[sizeof(tmp) - 1] = 0;
            aed_set_error("op %d (code %d): %s", k, op->code, tmp);
            return rc;
        }
    }
    return 0;
}

int aed_tape_profile(const aed_op* ops, int n, void* stream, float* ms_host) {
    AED_REQUIRE(ops && ms_host, "aed_tape_profile: null argument");
    std::vector<hipEvent_t> ev(n + 1);
    for (int k = 0; k <= n; ++k) AED_CHECK_HIP(hipEventCreate(&ev[k]));
    AED_CHECK_HIP(hipEventRecord(ev[0], (hipStream_t)stream));
    for (int k = 0; k < n; ++k) {
        int rc = aed_launch(ops + k, stream);
        if (rc) return rc;
        AED_CHECK_HIP(hipEventRecord(ev[k + 1], (hipStream_t)stream));
    }
    AED_CHECK_HIP(hipEventSynchronize(ev[n]));
    for (int k = 0; k < n; ++k) AED_CHECK_HIP(hipEventElapsedTime(&ms_host[k], ev[k], ev[k + 1]));
    for (int k = 0; k <= n; ++k) (void)hipEventDestroy(ev[k]);
    return 0;
}

int aed_graph_begin(void* stream) {
    AED_CHECK_HIP(hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal));
    return 0;
}
int aed_graph_end(void* stream, void** graph_exec_out) {
    hipGraph_t graph = nullptr;
    AED_CHECK_HIP(hipStreamEndCapture((hipStream_t)stream, &graph));
    hipGraphExec_t exec = nullptr;
    hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) {
        aed_set_error("hipGraphInstantiate: %s", hipGetErrorString(e));
        return 1;
    }
    *graph_exec_out = (void*)exec;
    return 0;
}
int aed_graph_launch(void* graph_exec, void* stream) {
    AED_CHECK_HIP(hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream));
    return 0;
}
int aed_graph_destroy(void* graph_exec) {
    if (graph_exec) AED_CHECK_HIP(hipGraphExecDestroy((hipGraphExec_t)graph_exec));
    return 0;
}

int aed_stream_create_cu_mask(void** stream_out, const uint32_t* mask_words, int n_words, int priority) {
    AED_REQUIRE(stream_out != nullptr, "aed_stream_create_cu_mask: null output");
    hipStream_t st = nullptr;
    if (n_words > 0) {
        AED_REQUIRE(mask_words != nullptr, "aed_stream_create_cu_mask: null mask");
        uint32_t any = 0;
        for (int k = 0; k < n_words; ++k) any |= mask_words[k];
        AED_REQUIRE(any != 0, "aed_stream_create_cu_mask: empty CU mask");
        AED_CHECK_HIP(hipExtStreamCreateWithCUMask(&st, (uint32_t)n_words, mask_words));
    } else {
        AED_CHECK_HIP(hipStreamCreateWithPriority(&st, hipStreamNonBlocking, priority));
    }
    *stream_out = (void*)st;
    return 0;
}
int aed_stream_destroy(void* stream) {
    if (stream) AED_CHECK_HIP(hipStreamDestroy((hipStream_t)stream));
    return 0;
}

int aed_cu_census(uint32_t* out_dev, int n_blocks, int spin_clocks, void* stream) {
    AED_REQUIRE(out_dev != nullptr && n_blocks > 0, "aed_cu_census: bad arguments");
    hipLaunchKernelGGL(cu_census_kernel, dim3(n_blocks), dim3(64), 0, (hipStream_t)stream, out_dev, spin_clocks);
    AED_CHECK_HIP(hipGetLastError());
    return 0;
}

int aed_event_create(void** ev_out) {
    hipEvent_t ev;
    AED_CHECK_HIP(hipEventCreate(&ev));
    *ev_out = (void*)ev;
    return 0;
}
int aed_event_record(void* ev, void* stream) {
    AED_CHECK_HIP(hipEventRecord((hipEvent_t)ev, (hipStream_t)stream));
    return 0;
}
int aed_event_elapsed_ms(void* a, void* b, float* ms_out) {
    AED_CHECK_HIP(hipEventSynchronize((hipEvent_t)b));
    AED_CHECK_HIP(hipEventElapsedTime(ms_out, (hipEvent_t)a, (hipEvent_t)b));
    return 0;
}
int aed_event_destroy(void* ev) {
    if (ev) AED_CHECK_HIP(hipEventDestroy((hipEvent_t)ev));
    return 0;
}

}  // extern "C"
