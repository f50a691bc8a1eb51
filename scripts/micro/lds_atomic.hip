// lds_atomic.hip — LDS pipe cost of ds_add_f64 (no return) against ds_write_b64 / ds_read_b64, 8 wavefronts per CU like relmc_eval_kernel:
// would block updates that ADD into their target (no read-modify-write through registers, no write-after-write ordering between the
// updates of one block) be affordable?  Address patterns: every lane its own 8-byte word; groups of 2 / 4 lanes on one word.
//   hipcc -O3 --offload-arch=gfx950 -o lds_atomic lds_atomic.hip && ./lds_atomic
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int OP>
__global__ void __launch_bounds__(256) k(unsigned long long* out, int iters, int share, unsigned long long mask)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned addr = (unsigned)(wave * 8192 + (lane / share) * 8);
    double a0 = 1.0, b0 = 0, c0 = 0, d0 = 0, e0 = 0;
    for (int i = threadIdx.x; i < 8192; i += 256) reinterpret_cast<double*>(smem)[i] = 0.0;
    __syncthreads();
    const bool act = (mask >> lane) & 1ull;
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (act) for (int i = 0; i < iters; ++i) {
        if (OP == 0)
            __asm__ volatile("ds_add_f64 %0, %1\n ds_add_f64 %0, %1 offset:1024\n ds_add_f64 %0, %1 offset:2048\n ds_add_f64 %0, %1 offset:3072\n"
                             "ds_add_f64 %0, %1 offset:4096\n ds_add_f64 %0, %1 offset:5120\n ds_add_f64 %0, %1 offset:6144\n ds_add_f64 %0, %1 offset:7168\n s_waitcnt lgkmcnt(0)"
                             :: "v"(addr), "v"(a0) : "memory");
        else if (OP == 1)
            __asm__ volatile("ds_write_b64 %0, %1\n ds_write_b64 %0, %1 offset:1024\n ds_write_b64 %0, %1 offset:2048\n ds_write_b64 %0, %1 offset:3072\n"
                             "ds_write_b64 %0, %1 offset:4096\n ds_write_b64 %0, %1 offset:5120\n ds_write_b64 %0, %1 offset:6144\n ds_write_b64 %0, %1 offset:7168\n s_waitcnt lgkmcnt(0)"
                             :: "v"(addr), "v"(a0) : "memory");
        else
            __asm__ volatile("ds_read_b64 %0, %4\n ds_read_b64 %1, %4 offset:1024\n ds_read_b64 %2, %4 offset:2048\n ds_read_b64 %3, %4 offset:3072\n"
                             "ds_read_b64 %0, %4 offset:4096\n ds_read_b64 %1, %4 offset:5120\n ds_read_b64 %2, %4 offset:6144\n ds_read_b64 %3, %4 offset:7168\n s_waitcnt lgkmcnt(0)"
                             : "=&v"(b0), "=&v"(c0), "=&v"(d0), "=&v"(e0) : "v"(addr) : "memory");
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    __syncthreads();
    if (lane == 0) out[(size_t)blockIdx.x * 4 + wave] = t1 - t0;
    if (b0 + c0 + d0 + e0 == 123.456) out[0] = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0 && OP == 0) out[(size_t)gridDim.x * 4] = (unsigned long long)reinterpret_cast<double*>(smem)[0];   // sum check
}

template <int OP>
double run(int share, unsigned long long mask, int iters, unsigned long long* dout, int blocks, unsigned long long* chk)
{
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<OP>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 80 * 1024, 0, dout, iters, share, mask);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 80 * 1024, 0, dout, iters, share, mask);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h((size_t)blocks * 4 + 1);
    hipMemcpy(h.data(), dout, h.size() * 8, hipMemcpyDeviceToHost);
    double s = 0; for (size_t i = 0; i + 1 < h.size(); ++i) s += (double)h[i];
    if (chk) *chk = h.back();
    return s / (h.size() - 1) / (iters * 8.0) / 8.0;
}

int main()
{
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int blocks = prop.multiProcessorCount * 2;
    unsigned long long* dout; hipMalloc(&dout, sizeof(unsigned long long) * (blocks * 4 + 1));
    const char* ops[] = {"ds_add_f64", "ds_write_b64", "ds_read_b64"};
    struct { const char* name; unsigned long long mask; } masks[] = {{"64 lanes", ~0ull}, {"16 lanes (0-15)", 0xffffull}, {"every 4th lane", 0x1111111111111111ull}};
    for (auto& m : masks)
        for (int share : {1, 2, 4, 16})
            for (int op = 0; op < 3; ++op) {
                unsigned long long chk = 0;
                double c = op == 0 ? run<0>(share, m.mask, 1000, dout, blocks, &chk) : (op == 1 ? run<1>(share, m.mask, 1000, dout, blocks, nullptr) : run<2>(share, m.mask, 1000, dout, blocks, nullptr));
                printf("%-13s %-18s %2d lanes per word: %7.2f cycles per instruction (pipe, 8 waves per CU)%s\n", ops[op], m.name, share, c, op == 0 ? (chk ? "  [sum ok]" : "  [sum 0?]") : "");
            }
    return 0;
}
