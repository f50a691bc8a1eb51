// fp64 issue-rate microbenchmark for gfx950: v_fma_f64 (VALU) vs v_mfma_f64_16x16x4_f64 (matrix core).
// Evidence for DESIGN.md "Why no MFMA": both reach the same fp64 rate, so a matrix formulation buys no throughput.
//   hipcc -O3 --offload-arch=gfx950 -o fp64_peak fp64_peak.hip && ./fp64_peak
#include <hip/hip_runtime.h>
#include <cstdio>

typedef double double4_t __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) k_valu(double* out, int iters)
{
    double a0 = threadIdx.x * 1e-9, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const double b = 1.0000001, c = 1e-9;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            a0 = __builtin_fma(a0, b, c); a1 = __builtin_fma(a1, b, c); a2 = __builtin_fma(a2, b, c); a3 = __builtin_fma(a3, b, c);
            a4 = __builtin_fma(a4, b, c); a5 = __builtin_fma(a5, b, c); a6 = __builtin_fma(a6, b, c); a7 = __builtin_fma(a7, b, c);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

__global__ void __launch_bounds__(256) k_mfma(double* out, int iters)
{
    double4_t c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    const double a = 1.0 + threadIdx.x * 1e-9, b = 1e-9;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}

int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount, blocks = cus * 8, iters = 20000;
    double* d; hipMalloc(&d, sizeof(double) * blocks * 256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int which = 0; which < 2; ++which) {
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(e0);
            if (which == 0) hipLaunchKernelGGL(k_valu, dim3(blocks), dim3(256), 0, 0, d, iters);
            else hipLaunchKernelGGL(k_mfma, dim3(blocks), dim3(256), 0, 0, d, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        // VALU: 64 FMA per thread-iteration (8 x 8), 2 flop each.  MFMA 16x16x4: 2*16*16*4 flop per wave instruction, 32 per iteration.
        const double flop = which == 0 ? (double)blocks * 256 * iters * 64 * 2 : (double)blocks * 4 * iters * 32 * (2.0 * 16 * 16 * 4);
        printf("%s: %.3f ms, %.1f TFLOP/s fp64 (%d CUs, clock %.0f MHz)\n", which == 0 ? "v_fma_f64            " : "v_mfma_f64_16x16x4f64",
               best, flop / best / 1e9, cus, p.clockRate / 1000.0);
    }
    return 0;
}
