// Which lane / byte of the scale operand of v_mfma_scale_f32_32x32x64_f8f6f4 scales which part of the data?  (No ISA manual in
// this image: measured.)  A data: lane < 32 bytes 0-15 = 1.0, bytes 16-31 = 2.0; lane >= 32 bytes 0-15 = 4.0, bytes 16-31 = 8.0
// (e4m3); B data = 1.0; all scales 1.0 (0x7f) except ONE byte of ONE lane's A-scale VGPR = x8.  C[5][0] = 240 + 7 * (weights of
// the data groups that byte scaled): 112 = (lane 5, bytes 0-15), 224 = (lane 5, 16-31), 448 = (lane 37, 0-15), 896 = (lane 37, 16-31).
//   hipcc --offload-arch=gfx950 -O2 tools/f8_probe.cpp -o audioeditingcode_amd/f8_probe && ./audioeditingcode_amd/f8_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int OPSEL>
__global__ void probe(float* c, int L, int byte, int on_b) {
    const int lane = threadIdx.x;
    const unsigned lo = lane < 32 ? 0x38383838u : 0x48484848u, hi = lane < 32 ? 0x40404040u : 0x50505050u;
    i32x8 a = {(int)lo, (int)lo, (int)lo, (int)lo, (int)hi, (int)hi, (int)hi, (int)hi};
    i32x8 one = {0x38383838, 0x38383838, 0x38383838, 0x38383838, 0x38383838, 0x38383838, 0x38383838, 0x38383838};
    unsigned s = 0x7f7f7f7fu, t = 0x7f7f7f7fu;
    if (lane == L) s = (s & ~(0xffu << (8 * byte))) | (130u << (8 * byte));
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (on_b) acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(one, a, acc, 0, 0, OPSEL, (int)t, OPSEL, (int)s);
    else acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, one, acc, 0, 0, OPSEL, (int)s, OPSEL, (int)t);
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), col = lane & 31;
        c[row * 32 + col] = acc[r];
    }
}

int main() {
    float* d;
    hipMalloc(&d, 32 * 32 * 4);
    std::vector<float> h(32 * 32);
    for (int on_b = 0; on_b < 2; ++on_b)
        for (int opsel = 0; opsel < 4; ++opsel)
            for (int L : {5, 37})
                for (int byte = 0; byte < 4; ++byte) {
                    switch (opsel) {
                        case 0: probe<0><<<1, 64>>>(d, L, byte, on_b); break;
                        case 1: probe<1><<<1, 64>>>(d, L, byte, on_b); break;
                        case 2: probe<2><<<1, 64>>>(d, L, byte, on_b); break;
                        default: probe<3><<<1, 64>>>(d, L, byte, on_b); break;
                    }
                    hipMemcpy(h.data(), d, 32 * 32 * 4, hipMemcpyDeviceToHost);
                    // scaled operand A: its row index is the C row; scaled operand B: its row index is the C column
                    const float v5 = on_b ? h[0 * 32 + 5] : h[5 * 32 + 0], v6 = on_b ? h[0 * 32 + 6] : h[6 * 32 + 0];
                    printf("{\"operand\": \"%s\", \"opsel\": %d, \"lane\": %d, \"byte\": %d, \"C5\": %.0f, \"delta_over_7\": %.0f, \"C6\": %.0f}\n",
                           on_b ? "B" : "A", opsel, L, byte, v5, (v5 - 240.f) / 7.f, v6);
                }
    return 0;
}
