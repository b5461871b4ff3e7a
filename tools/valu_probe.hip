// What does the fp64 VALU of one gfx950 SIMD sustain?  CH independent fma chains per wave, W waves per SIMD.
// Reports cycles per wave-instruction seen by one SIMD (4.0 = the 16-lane SIMD saturated by wave64 fp64 ops).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CH, int OP>
__global__ __launch_bounds__(256) void k(double *out, double a, double b, int iters) {
    double x[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) x[c] = threadIdx.x * 1e-3 + c;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                if (OP == 0) x[c] = __builtin_fma(x[c], a, b);
                else if (OP == 1) x[c] = x[c] * a;
                else if (OP == 2) x[c] = x[c] + b;
                else { float y = (float)x[c]; y = __builtin_fmaf(y, (float)a, (float)b); x[c] = y; }
            }
    }
    double s = 0;
#pragma unroll
    for (int c = 0; c < CH; ++c) s += x[c];
    if (s == 12345.678) out[0] = s;
}
template <int CH, int OP> void run(const char *name, double *out) {
    for (int W = 1; W <= 4; ++W) {
        const int iters = 2000, blocks = 256 * W;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL((k<CH, OP>), dim3(blocks), dim3(256), 0, 0, out, 1.0000001, 1e-9, iters);
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<CH, OP>), dim3(blocks), dim3(256), 0, 0, out, 1.0000001, 1e-9, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double instr_per_simd = (double)iters * 8 * CH * W;     // wave-instructions one SIMD executes
        printf("%-8s chains=%d waves/SIMD=%d : %.3f ms  -> %.2f ns per wave-instr per SIMD (x2.4 GHz = %.2f cycles)\n", name, CH, W, ms,
               ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
    }
}
int main() {
    double *out; hipMalloc(&out, 8);
    run<1, 0>("fma64", out); run<2, 0>("fma64", out); run<4, 0>("fma64", out); run<8, 0>("fma64", out);
    run<1, 1>("mul64", out); run<4, 1>("mul64", out);
    run<1, 2>("add64", out); run<4, 2>("add64", out);
    return 0;
}
