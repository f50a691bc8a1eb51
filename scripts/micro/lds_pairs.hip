// lds_pairs.hip — which lanes of a wavefront share an LDS service group, and how wide is the bank window, per DS instruction.
// For every lane pair (i, j): all 64 lanes access distinct bank slots except that j is moved onto i's slot at a different address
// (i's address + `delta` bytes).  If i and j are served in the same group the instruction takes one more LDS cycle.
//   hipcc -O3 --offload-arch=gfx950 -o lds_pairs lds_pairs.hip && ./lds_pairs > pairs.txt
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template <int OP>
__global__ void __launch_bounds__(256) k(const unsigned* __restrict__ addr_tab, unsigned long long* out, int iters, unsigned long long mask)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned addr = addr_tab[lane] + (unsigned)wave * 16384u;
    double a0 = 1.0, a1 = 2.0, b0 = 0, b1 = 0;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    if ((mask >> lane) & 1ull)
    for (int i = 0; i < iters; ++i) {
        if (OP == 0)
            __asm__ volatile("ds_read_b128 %0, %1\n ds_read_b128 %0, %1\n ds_read_b128 %0, %1\n ds_read_b128 %0, %1\n"
                             "ds_read_b128 %0, %1\n ds_read_b128 %0, %1\n ds_read_b128 %0, %1\n ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)"
                             : "=&v"(*(reinterpret_cast<__attribute__((ext_vector_type(2))) double*>(&b0))) : "v"(addr) : "memory");
        else if (OP == 1)
            __asm__ volatile("ds_write_b128 %0, %1\n ds_write_b128 %0, %1\n ds_write_b128 %0, %1\n ds_write_b128 %0, %1\n"
                             "ds_write_b128 %0, %1\n ds_write_b128 %0, %1\n ds_write_b128 %0, %1\n ds_write_b128 %0, %1\n s_waitcnt lgkmcnt(0)"
                             :: "v"(addr), "v"(*(reinterpret_cast<__attribute__((ext_vector_type(2))) double*>(&a0))) : "memory");
        else if (OP == 2)
            __asm__ volatile("ds_read_b64 %0, %1\n ds_read_b64 %0, %1\n ds_read_b64 %0, %1\n ds_read_b64 %0, %1\n"
                             "ds_read_b64 %0, %1\n ds_read_b64 %0, %1\n ds_read_b64 %0, %1\n ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)"
                             : "=&v"(b0) : "v"(addr) : "memory");
        else
            __asm__ volatile("ds_write_b64 %0, %1\n ds_write_b64 %0, %1\n ds_write_b64 %0, %1\n ds_write_b64 %0, %1\n"
                             "ds_write_b64 %0, %1\n ds_write_b64 %0, %1\n ds_write_b64 %0, %1\n ds_write_b64 %0, %1\n s_waitcnt lgkmcnt(0)"
                             :: "v"(addr), "v"(a0) : "memory");
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0) out[(size_t)blockIdx.x * 4 + wave] = t1 - t0;
    if (b0 + b1 + a1 == 123.456) out[0] = 0;
}

template <int OP>
double run(const std::vector<unsigned>& tab, unsigned* dtab, unsigned long long* dout, int blocks, int iters, unsigned long long mask = ~0ull)
{
    hipMemcpy(dtab, tab.data(), 64 * sizeof(unsigned), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 72 * 1024, 0, dtab, dout, iters, mask);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h((size_t)blocks * 4);
    hipMemcpy(h.data(), dout, h.size() * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : h) s += (double)v;
    return s / h.size() / (iters * 8.0) / 8.0;           // cycles per instruction per CU (8 waves per CU)
}

int main()
{
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int blocks = prop.multiProcessorCount * 2, iters = 300;
    unsigned* dtab; unsigned long long* dout;
    hipMalloc(&dtab, 64 * sizeof(unsigned)); hipMalloc(&dout, sizeof(unsigned long long) * blocks * 4);
    for (int op = 0; op < 4; ++op) {
        auto set_attr = [&](auto fn) { hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024); };
        if (op == 0) set_attr(&k<0>); else if (op == 1) set_attr(&k<1>); else if (op == 2) set_attr(&k<2>); else set_attr(&k<3>);
        const unsigned width = (op == 0 || op == 1) ? 16u : 8u;          // bytes per lane
        auto go = [&](const std::vector<unsigned>& tab, unsigned long long mask = ~0ull) {
            return op == 0 ? run<0>(tab, dtab, dout, blocks, iters, mask) : op == 1 ? run<1>(tab, dtab, dout, blocks, iters, mask)
                 : op == 2 ? run<2>(tab, dtab, dout, blocks, iters, mask) : run<3>(tab, dtab, dout, blocks, iters, mask);
        };
        std::vector<unsigned> base(64);
        for (int l = 0; l < 64; ++l) base[l] = l * width;                 // consecutive: conflict free
        const double c0 = go(base);
        printf("op %d base %.3f\n", op, c0);
        // bank window: lane 1 moved onto lane 0's slot at +delta
        for (unsigned delta : {64u, 128u, 256u, 512u, 1024u, 2048u}) {
            std::vector<unsigned> t = base; t[1] = base[0] + delta;
            printf("op %d window delta %u lanes(0,1) %.3f\n", op, delta, go(t));
        }
        // pair matrix with ONLY lanes i and j active: both on bank slot 0, at different addresses -> one more cycle iff served in the same group
        if (op == 1 || op == 3) continue;
        {
            std::vector<unsigned> t(64, 0);
            const double c1 = go(t, 1ull);
            printf("op %d one lane %.3f\n", op, c1);
            for (int i = 0; i < 64; ++i) {
                printf("op %d row %2d:", op, i);
                for (int j = 0; j < 64; ++j) {
                    if (i == j) { printf(" ."); continue; }
                    std::vector<unsigned> t2(64, 0); t2[j] = 4096u;
                    const double c = go(t2, (1ull << i) | (1ull << j));
                    printf(" %c", c > c1 + 0.4 ? 'X' : '-');
                }
                printf("\n");
            }
        }
    }
    return 0;
}
