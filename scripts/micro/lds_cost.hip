// lds_cost.hip — LDS pipe cost of the instructions the solver passes are made of, as a function of the EXEC mask.
// 8 wavefronts per CU (2 workgroups x 4, like relmc_eval_kernel<.., Tile24>) issue the same LDS instruction back to back on
// conflict-free addresses; cycles per instruction per CU = (kernel cycles) / (instructions per wave x 8 waves).
//   hipcc -O3 --offload-arch=gfx950 -o lds_cost lds_cost.hip && ./lds_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template <int OP>
__global__ void __launch_bounds__(256) lds_kernel(unsigned long long* out, int iters, unsigned long long mask, int stride16)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // per-lane address: 16-byte slots, stride16 slots apart (1 = consecutive = conflict free for b128)
    unsigned addr = (unsigned)(wave * 8192 + lane * 16 * stride16);
    const bool act = (mask >> lane) & 1ull;
    double a0 = 1.0, a1 = 2.0, b0 = 0, b1 = 0, c0 = 0, c1 = 0, d0 = 0, d1 = 0, e0 = 0, e1 = 0;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (act) {
        for (int i = 0; i < iters; ++i) {
            if (OP == 0)       // ds_read_b128 x 8
                __asm__ volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:1024\n ds_read_b128 %2, %4 offset:2048\n ds_read_b128 %3, %4 offset:3072\n"
                                 "ds_read_b128 %0, %4 offset:4096\n ds_read_b128 %1, %4 offset:5120\n ds_read_b128 %2, %4 offset:6144\n ds_read_b128 %3, %4 offset:7168\n s_waitcnt lgkmcnt(0)"
                                 : "=&v"(*(reinterpret_cast<__attribute__((ext_vector_type(2))) double*>(&b0))), "=&v"(*(reinterpret_cast<__attribute__((ext_vector_type(2))) double*>(&c0))),
                                   "=&v"(*(reinterpret_cast<__attribute__((ext_vector_type(2))) double*>(&d0))), "=&v"(*(reinterpret_cast<__attribute__((ext_vector_type(2))) double*>(&e0)))
                                 : "v"(addr) : "memory");
            else if (OP == 1)  // ds_write_b128 x 8
                __asm__ volatile("ds_write_b128 %0, %1\n ds_write_b128 %0, %1 offset:1024\n ds_write_b128 %0, %1 offset:2048\n ds_write_b128 %0, %1 offset:3072\n"
                                 "ds_write_b128 %0, %1 offset:4096\n ds_write_b128 %0, %1 offset:5120\n ds_write_b128 %0, %1 offset:6144\n ds_write_b128 %0, %1 offset:7168\n s_waitcnt lgkmcnt(0)"
                                 :: "v"(addr), "v"(*(reinterpret_cast<__attribute__((ext_vector_type(2))) double*>(&a0))) : "memory");
            else if (OP == 2)  // ds_read_b64 x 8
                __asm__ volatile("ds_read_b64 %0, %4\n ds_read_b64 %1, %4 offset:1024\n ds_read_b64 %2, %4 offset:2048\n ds_read_b64 %3, %4 offset:3072\n"
                                 "ds_read_b64 %0, %4 offset:4096\n ds_read_b64 %1, %4 offset:5120\n ds_read_b64 %2, %4 offset:6144\n ds_read_b64 %3, %4 offset:7168\n s_waitcnt lgkmcnt(0)"
                                 : "=&v"(b0), "=&v"(c0), "=&v"(d0), "=&v"(e0) : "v"(addr) : "memory");
            else               // ds_write_b64 x 8
                __asm__ volatile("ds_write_b64 %0, %1\n ds_write_b64 %0, %1 offset:1024\n ds_write_b64 %0, %1 offset:2048\n ds_write_b64 %0, %1 offset:3072\n"
                                 "ds_write_b64 %0, %1 offset:4096\n ds_write_b64 %0, %1 offset:5120\n ds_write_b64 %0, %1 offset:6144\n ds_write_b64 %0, %1 offset:7168\n s_waitcnt lgkmcnt(0)"
                                 :: "v"(addr), "v"(a0) : "memory");
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    __syncthreads();
    if (lane == 0) out[(size_t)blockIdx.x * 4 + wave] = t1 - t0;
    if (b0 + b1 + c0 + c1 + d0 + d1 + e0 + e1 + a1 == 123.456) out[0] = 0;
}

template <int OP>
double run(unsigned long long mask, int stride16, int iters, unsigned long long* dout, int blocks)
{
    hipFuncSetAttribute(reinterpret_cast<const void*>(&lds_kernel<OP>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    hipLaunchKernelGGL(lds_kernel<OP>, dim3(blocks), dim3(256), 80 * 1024, 0, dout, iters, mask, stride16);
    hipLaunchKernelGGL(lds_kernel<OP>, dim3(blocks), dim3(256), 80 * 1024, 0, dout, iters, mask, stride16);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h((size_t)blocks * 4);
    hipMemcpy(h.data(), dout, h.size() * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : h) s += (double)v;
    return s / h.size() / (iters * 8.0) / 8.0;      // cycles per instruction per CU with 8 waves sharing the pipe
}

int main()
{
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int blocks = prop.multiProcessorCount * 2;
    unsigned long long* dout; hipMalloc(&dout, sizeof(unsigned long long) * blocks * 4);
    struct { const char* name; unsigned long long mask; } masks[] = {
        {"64 lanes", ~0ull}, {"48 lanes (0-47)", 0x0000ffffffffffffull}, {"32 lanes (0-31)", 0xffffffffull}, {"16 lanes (0-15)", 0xffffull},
        {"32 lanes (even)", 0x5555555555555555ull}, {"16 lanes (every 4th)", 0x1111111111111111ull}, {"12 of each 16", 0x0fff0fff0fff0fffull}, {"4 lanes (0-3)", 0xfull}};
    const char* ops[] = {"ds_read_b128", "ds_write_b128", "ds_read_b64", "ds_write_b64"};
    for (int op = 0; op < 4; ++op)
        for (auto& m : masks) {
            double c1 = 0;
            if (op == 0) c1 = run<0>(m.mask, 1, 2000, dout, blocks); else if (op == 1) c1 = run<1>(m.mask, 1, 2000, dout, blocks);
            else if (op == 2) c1 = run<2>(m.mask, 1, 2000, dout, blocks); else c1 = run<3>(m.mask, 1, 2000, dout, blocks);
            printf("%-14s %-22s %6.2f cycles per instruction (pipe, 8 waves per CU)\n", ops[op], m.name, c1);
        }
    return 0;
}
