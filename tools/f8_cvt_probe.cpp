// What does v_cvt_pk_fp8_f32 do at ties, below the normal range and above 448 on gfx950?  (Decides the rounding rule of
// oracle/mxfp8.py.)   hipcc --offload-arch=gfx950 -O2 tools/f8_cvt_probe.cpp -o audioeditingcode_amd/f8_cvt_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float* x, unsigned* q, int n) {
    const int i = threadIdx.x;
    if (i < n) q[i] = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(x[i], 0.f, 0, false) & 0xffu;
}
int main() {
    const float xs[] = {1.0f, 1.0625f, 1.125f, 1.1875f, 1.25f, 1.3125f, -1.0625f, -1.1875f, 1.06250012f, 1.06249988f,
                        0.015625f, 0.0146484375f, 0.013671875f, 0.001953125f, 0.0009765625f, 0.00097656256f, 0.0029296875f,
                        448.f, 464.f, 479.9f, 480.f, 500.f, 1000.f, -500.f, 0.f, 416.f, 432.f, 240.f, 248.f};
    const int n = sizeof(xs) / sizeof(xs[0]);
    float* dx; unsigned* dq; unsigned hq[64];
    hipMalloc(&dx, sizeof(xs)); hipMalloc(&dq, 64 * 4);
    hipMemcpy(dx, xs, sizeof(xs), hipMemcpyHostToDevice);
    k<<<1, 64>>>(dx, dq, n);
    hipMemcpy(hq, dq, n * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i) {
        const unsigned b = hq[i], s = b >> 7, e = (b >> 3) & 15, m = b & 7;
        double v = e == 0 ? m / 8.0 * 0.015625 : (1 + m / 8.0) * (e >= 7 ? (double)(1u << (e - 7)) : 1.0 / (double)(1u << (7 - e)));
        if (e == 15 && m == 7) v = 0.0 / 0.0;
        printf("{\"x\": %.10g, \"byte\": \"0x%02x\", \"value\": %.10g}\n", xs[i], b, s ? -v : v);
    }
    return 0;
}
