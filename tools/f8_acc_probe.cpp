// How exact is ONE v_mfma_scale_f32_32x32x64_f8f6f4 (64 e4m3 products + block scales + fp32 accumulator)?  Random e4m3 data with
// a wide spread of magnitudes, scales 2^0: device result against the exact sum on the host (fp64).  Operand layout as measured by
// tools/f8_probe.cpp.   hipcc --offload-arch=gfx950 -O2 tools/f8_acc_probe.cpp -o audioeditingcode_amd/f8_acc_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(const unsigned char* A, const unsigned char* B, float* C) {     // A, B: [32 rows][64 k] bytes
    const int lane = threadIdx.x, i = lane & 31, h = lane >> 5;
    i32x8 a, b;
    for (int v = 0; v < 8; ++v) {
        const int k0 = (v < 4 ? 16 * h + 4 * v : 32 + 16 * h + 4 * (v - 4));
        a[v] = *reinterpret_cast<const int*>(A + i * 64 + k0);
        b[v] = *reinterpret_cast<const int*>(B + i * 64 + k0);
    }
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + i] = acc[r];
}
static double dec(unsigned b) {
    const unsigned s = b >> 7, e = (b >> 3) & 15, m = b & 7;
    const double v = e == 0 ? m / 8.0 * std::ldexp(1.0, -6) : (1 + m / 8.0) * std::ldexp(1.0, (int)e - 7);
    return s ? -v : v;
}
int main() {
    unsigned char hA[32 * 64], hB[32 * 64];
    srand(7);
    for (int spread = 0; spread < 3; ++spread) {
        for (int i = 0; i < 32 * 64; ++i) {
            // spread 0: exponents 6..9 (similar magnitudes); 1: 3..12; 2: the whole range 0..15 (no NaN byte)
            const int lo = spread == 0 ? 6 : spread == 1 ? 3 : 0, hi = spread == 0 ? 9 : spread == 1 ? 12 : 15;
            auto gen = [&]() { unsigned e = lo + rand() % (hi - lo + 1), m = rand() % 8, s = rand() % 2; if (e == 15 && m == 7) m = 6; return (unsigned char)((s << 7) | (e << 3) | m); };
            hA[i] = gen(); hB[i] = gen();
        }
        unsigned char *dA, *dB; float* dC; float hC[32 * 32];
        hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dC, sizeof(hC));
        hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
        k<<<1, 64>>>(dA, dB, dC);
        hipMemcpy(hC, dC, sizeof(hC), hipMemcpyDeviceToHost);
        double num = 0, den = 0, worst = 0;
        for (int i = 0; i < 32; ++i)
            for (int j = 0; j < 32; ++j) {
                double ref = 0, mag = 0;
                for (int kk = 0; kk < 64; ++kk) { const double p = dec(hA[i * 64 + kk]) * dec(hB[j * 64 + kk]); ref += p; mag += std::fabs(p); }
                const double d = hC[i * 32 + j] - ref;
                num += d * d; den += ref * ref;
                if (std::fabs(d) / mag > worst) worst = std::fabs(d) / mag;
            }
        printf("{\"exponent_spread\": %d, \"rel_l2_vs_exact\": %.3e, \"worst_abs_err_over_sum_of_magnitudes\": %.3e}\n", spread, std::sqrt(num / den), worst);
    }
    return 0;
}
