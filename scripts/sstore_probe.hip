// Probe: cost of scalar stores (s_store_dwordx2 / x4) issued by one wavefront per SIMD, and their visibility to a later kernel.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -w scripts/sstore_probe.hip -o scripts/sstore_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
template <int WIDTH>
__global__ __launch_bounds__(256, 1) void writer(unsigned long long *out, unsigned long long *clk, int reps) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned long long *base = out + ((size_t)blockIdx.x * 4 + wave) * 128 * reps;
    float x = (float)(threadIdx.x & 63) - 31.5f;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r) {
        unsigned long long *p = base + (size_t)r * 128;
#pragma unroll
        for (int k = 0; k < 128; k += 2) {
            const unsigned long long m0 = __builtin_amdgcn_ballot_w64(x > (float)(k - 64) * 0.5f);
            const unsigned long long m1 = __builtin_amdgcn_ballot_w64(x > (float)(k - 63) * 0.5f);
            if (WIDTH == 2) {
                asm volatile("s_store_dwordx2 %0, %1, %2" : : "s"(m0), "s"(p), "n"(0) : "memory");
                asm volatile("s_store_dwordx2 %0, %1, %2" : : "s"(m1), "s"(p), "n"(8) : "memory");
            } else {
                typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
                const u64x2 both = {m0, m1};
                asm volatile("s_store_dwordx4 %0, %1, %2" : : "s"(both), "s"(p), "n"(0) : "memory");
            }
            p += 2;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_dcache_wb" ::: "memory");
    unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}
__global__ void reader(const unsigned long long *in, size_t n, unsigned long long *sum) {
    unsigned long long s = 0;
    for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += __builtin_popcountll(in[i]);
    atomicAdd(sum, s);
}
int main() {
    const int reps = 64, blocks = 256;
    const size_t n = (size_t)blocks * 4 * 128 * reps;
    unsigned long long *out, *clk, *sum;
    hipMalloc(&out, n * 8); hipMalloc(&clk, 64); hipMalloc(&sum, 8);
    for (int width : {2, 4}) {
        hipMemset(out, 0, n * 8); hipMemset(sum, 0, 8);
        if (width == 2) hipLaunchKernelGGL(writer<2>, dim3(blocks), dim3(256), 0, 0, out, clk, reps);
        else hipLaunchKernelGGL(writer<4>, dim3(blocks), dim3(256), 0, 0, out, clk, reps);
        hipLaunchKernelGGL(reader, dim3(1024), dim3(256), 0, 0, out, n, sum);
        unsigned long long h, c; hipMemcpy(&h, sum, 8, hipMemcpyDeviceToHost); hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
        // expected popcount: for threshold t_k = (k-64)*0.5, lanes with x > t_k
        unsigned long long want = 0;
        for (int k = 0; k < 128; ++k) for (int l = 0; l < 64; ++l) want += ((float)l - 31.5f) > (float)(k - 64) * 0.5f;
        want *= (unsigned long long)blocks * 4 * reps;
        printf("s_store_dwordx%d: %.1f cycles per 128 lane-masks (1 KiB) per wave incl. the ballots; popcount %llu, expected %llu -> %s\n",
               width, (double)c / reps, h, want, h == want ? "visible, correct" : "MISMATCH");
    }
    return 0;
}
