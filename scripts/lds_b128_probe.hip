// Which lanes does ds_read_b128 serve together?  Times a stream of ds_read_b128 for lane -> address maps and reports cycles
// per instruction (one wavefront, nothing else on the CU): 4 x 16 lanes x 16 B = 1 KiB needs >= 4 LDS cycles at 256 B /
// clock (guide: 128 B / clock / CU -> 8); every bank conflict adds passes.  Used to pick the fragment swizzle of the
// split-f16 stream (mlp_layout.h): A fragment of v_mfma_f32_16x16x32_f16 = lane (row n = lane & 15, k-group g = lane >> 4).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -w scripts/lds_b128_probe.hip -o scripts/lds_b128_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(const int *lane_off, int reps, unsigned long long *out, float *sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)lds + (unsigned)lane_off[threadIdx.x];
    for (int e = threadIdx.x; e < 16384; e += 64) reinterpret_cast<float *>(lds)[e] = (float)e;
    __syncthreads();
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r) {
        f32x4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[j]) : "v"(base), "n"(j * 1024));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += v[j];
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    sink[threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}
static double run(const std::vector<int> &off, int *d_off, unsigned long long *d_out, float *d_sink) {
    hipMemcpy(d_off, off.data(), 256, hipMemcpyHostToDevice);
    const int reps = 2000;
    unsigned long long best = ~0ull, h;
    for (int t = 0; t < 3; ++t) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 65536, 0, d_off, reps, d_out, d_sink);
        hipMemcpy(&h, d_out, 8, hipMemcpyDeviceToHost);
        best = std::min(best, h);
    }
    return (double)best / (reps * 8.0);
}
int main() {
    int *d_off; unsigned long long *d_out; float *d_sink;
    hipMalloc(&d_off, 256); hipMalloc(&d_out, 8); hipMalloc(&d_sink, 256);
    hipFuncSetAttribute(reinterpret_cast<const void *>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    std::vector<int> off(64);
    for (int l = 0; l < 64; ++l) off[l] = l * 16;
    printf("linear (lane * 16)                                   %6.2f cycles / ds_read_b128\n", run(off, d_off, d_out, d_sink));
    for (int l = 0; l < 64; ++l) off[l] = 0;
    printf("broadcast (all lanes one address)                    %6.2f\n", run(off, d_off, d_out, d_sink));
    for (int l = 0; l < 64; ++l) off[l] = (l & 15) * 64;
    printf("16 rows x 64 B, same slot in every lane group        %6.2f\n", run(off, d_off, d_out, d_sink));
    for (int l = 0; l < 64; ++l) { const int i = l & 31, h = l >> 5; off[l] = i * 64 + (((0 + h) ^ ((i >> 2) & 3)) << 4); }
    printf("bf16 kernel: row = lane & 31, slot (h) ^ (row>>2 & 3)   %6.2f\n", run(off, d_off, d_out, d_sink));
    for (int l = 0; l < 64; ++l) { const int n = l & 15, g = l >> 4; off[l] = n * 64 + ((g ^ ((n >> 2) & 3)) << 4); }
    printf("f16x2 round-6 first cut: slot g ^ (row>>2 & 3)          %6.2f\n", run(off, d_off, d_out, d_sink));
    int sig[4] = {0, 1, 2, 3};
    double best = 1e9; int bs[4] = {0, 0, 0, 0};
    do {
        for (int l = 0; l < 64; ++l) { const int n = l & 15, g = l >> 4; off[l] = n * 64 + ((g ^ sig[(n >> 2) & 3]) << 4); }
        const double c = run(off, d_off, d_out, d_sink);
        printf("f16x2 slot g ^ sigma(row>>2 & 3), sigma = %d %d %d %d        %6.2f\n", sig[0], sig[1], sig[2], sig[3], c);
        if (c < best) { best = c; for (int q = 0; q < 4; ++q) bs[q] = sig[q]; }
    } while (std::next_permutation(sig, sig + 4));
    printf("best sigma = %d %d %d %d : %.2f cycles\n", bs[0], bs[1], bs[2], bs[3], best);
    // other shapes of swizzle: slot (g + rot(row)) & 3
    for (int a = 0; a < 4; ++a) {
        for (int l = 0; l < 64; ++l) { const int n = l & 15, g = l >> 4; off[l] = n * 64 + (((g + ((n >> 2) & 3) * a) & 3) << 4); }
        printf("f16x2 slot (g + %d * (row>>2 & 3)) & 3                     %6.2f\n", a, run(off, d_off, d_out, d_sink));
    }
    return 0;
}
