// Microbenchmark: how fast can one CU stream an L2-resident weight image into LDS by LDS-DMA
// (global_load_lds_dwordx4) next to the bf16 MFMA + ds_read_b128 consumer of the fused MLP kernel, as a function
// of the geometry: waves per workgroup, column blocks per wave (MFMAs per A fragment), read-ahead depth, ring.
// All CUs stream the SAME 1.34 MB image (21 x 64 KiB), like the fused MLP kernels do.  Also reports the shader
// clock the chip sustains under each load (s_memtime ticks per 100 MHz s_memrealtime tick).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -w scripts/ldsdma_stream.hip -o scripts/ldsdma_stream.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void lds_dma_16s(const char *src, unsigned lane_off, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(lane_off), "s"(src), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void wait_lgkm() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(%0)" : : "n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}

constexpr int IMAGE_BYTES = 21 * 65536;

// MODE 0 = DMA only, 1 = DMA + consumer, 2 = consumer only (no DMA), 3 = MFMAs only (no LDS reads, no DMA)
template <int WAVES, int SLOT, int NSLOTS, int MODE, int NCB, int DEPTH, int SPREAD, int CHAIN>
__global__ __launch_bounds__(WAVES * 64, 1) void stream_kernel(const char *__restrict__ image, int steps, float *out,
                                                                unsigned long long *clk) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int PIECES = SLOT / 1024 / WAVES;      // 1-KiB pieces per wave per slot
    constexpr int IMG_SLOTS = IMAGE_BYTES / SLOT;
    constexpr int R = SLOT / 1024;                   // A fragments (ds_read_b128) per wave per slot
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lane_off = lane * 16u;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)lds;
    const char *src_wave = image + wave * (PIECES * 1024);
    f32x16 acc[NCB][8];
#pragma unroll
    for (int c = 0; c < NCB; ++c)
#pragma unroll
        for (int f = 0; f < 8; ++f)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][f][r] = 0.f;
    bf16x8 b[NCB];
#pragma unroll
    for (int c = 0; c < NCB; ++c)
#pragma unroll
        for (int e = 0; e < 8; ++e) b[c][e] = (__bf16)(0.01f * (float)((lane * 7 + e * 3 + c) % 97) - 0.4f);

    unsigned long long t0 = 0, r0 = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        t0 = __builtin_readcyclecounter();
        r0 = wall_clock64();
    }
    auto issue_piece = [&](int step, int p) {
        const int pos = step % IMG_SLOTS, slot = step % NSLOTS;
        lds_dma_16s(src_wave + (size_t)pos * SLOT + p * 1024, lane_off, lds0 + slot * SLOT + wave * (PIECES * 1024) + p * 1024);
    };
    auto issue = [&](int step) {
        if (MODE >= 2) return;
#pragma unroll
        for (int p = 0; p < PIECES; ++p) issue_piece(step, p);
    };
#pragma unroll
    for (int s = 0; s < NSLOTS - 1; ++s) issue(s);
    for (int step = 0; step < steps; ++step) {
        if (MODE < 2) wait_vm<PIECES * (NSLOTS - 2)>();
        __builtin_amdgcn_s_barrier();
        if (!(SPREAD && MODE == 1)) issue(step + NSLOTS - 1);
        if (MODE == 3) {
#pragma unroll
            for (int k = 0; k < R; ++k)
#pragma unroll
                for (int c = 0; c < NCB; ++c)
                    acc[c][(k / CHAIN) & 7] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[c ^ 1 & (NCB - 1)], b[c], acc[c][(k / CHAIN) & 7], 0, 0, 0);
        } else if (MODE >= 1) {
            const unsigned base = lds0 + (step % NSLOTS) * SLOT + lane * 16u;
            bf16x8 a[DEPTH];
#pragma unroll
            for (int k = 0; k < DEPTH; ++k)
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a[k]) : "v"(base), "n"(k * 1024));
#pragma unroll
            for (int k = 0; k < R; ++k) {
                if (R - 1 - k >= DEPTH - 1) wait_lgkm<DEPTH - 1>();
                else if (R - 1 - k == 2) wait_lgkm<2>();
                else if (R - 1 - k == 1) wait_lgkm<1>();
                else wait_lgkm<0>();
                const bf16x8 av = a[k % DEPTH];
#pragma unroll
                for (int c = 0; c < NCB; ++c)
                    acc[c][(k / CHAIN) & 7] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, b[c], acc[c][(k / CHAIN) & 7], 0, 0, 0);
                if (k + DEPTH < R)
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a[k % DEPTH]) : "v"(base + ((k + DEPTH) / 32) * 32768u), "n"(((k + DEPTH) % 32) * 1024));
                // SPREAD: one DMA piece every R/PIECES fragments, the two waves of a SIMD half a period apart
                if (SPREAD && MODE == 1 && (k % (R / PIECES)) == (SPREAD == 2 && wave >= 4 ? R / PIECES / 2 : 0))
                    issue_piece(step + NSLOTS - 1, k / (R / PIECES));
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        clk[0] = __builtin_readcyclecounter() - t0;
        clk[1] = wall_clock64() - r0;
    }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NCB; ++c)
#pragma unroll
        for (int f = 0; f < 8; ++f) s += acc[c][f][0] + acc[c][f][7];
    if (s == 12345.678f) out[0] = s;
}

template <int WAVES, int SLOT, int NSLOTS, int MODE, int NCB, int DEPTH, int SPREAD = 0, int CHAIN = 1>
void run(const char *image, float *out, unsigned long long *clk, int cus) {
    auto kern = stream_kernel<WAVES, SLOT, NSLOTS, MODE, NCB, DEPTH, SPREAD, CHAIN>;
    const int lds = SLOT * NSLOTS;
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
        printf("attr failed\n");
        return;
    }
    const int steps = (int)(64ll * IMAGE_BYTES / SLOT);   // 64 passes over the image = 88 MB per CU
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e30f;
    unsigned long long h[2] = {0, 1};
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(cus), dim3(WAVES * 64), lds, 0, image, steps, out, clk);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) {
            best = ms;
            hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
        }
    }
    if (hipGetLastError() != hipSuccess) printf("launch failed\n");
    const double bytes = (double)steps * SLOT;
    const double mhz = (double)h[0] / (double)h[1] * 100.0;
    const double us_step = best * 1e3 / (bytes / 65536.0);
    // MFMAs per wave per 64 KiB = 64 * NCB at 32 cycles each, WAVES/4 waves per SIMD; samples per step = 32 * NCB * WAVES
    const double mfma_cycles = 64.0 * NCB * 32 * (WAVES / 4);
    printf("w%d ncb%d depth%d slot %2dK ring%d mode%d spread%d chain%d: %7.3f ms %6.1f GB/s/CU  %.3f us/64KiB = %.3f us per 256 samples; clock %4.0f MHz; "
           "MFMA busy %.3f (of the measured clock), %.0f TF/s equivalent\n",
           WAVES, NCB, DEPTH, SLOT / 1024, NSLOTS, MODE, SPREAD, CHAIN, best, bytes / (best * 1e-3) / 1e9, us_step,
           us_step * 256.0 / (32.0 * NCB * WAVES), mhz, MODE ? mfma_cycles / (us_step * mhz) : 0.0,
           MODE ? (double)cus * 4 * (mfma_cycles / 32) * 32768.0 / (us_step * 1e-6) / 1e12 / (WAVES / 4) * (WAVES / 4) : 0.0);
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    char *image;
    float *out;
    unsigned long long *clk;
    hipMalloc(&image, IMAGE_BYTES + 65536);
    hipMalloc(&out, 64);
    hipMalloc(&clk, 64);
    std::vector<unsigned short> host((IMAGE_BYTES + 65536) / 2);
    for (size_t i = 0; i < host.size(); ++i) host[i] = (unsigned short)(((rand() & 1) << 15) | 0x3c00 | (rand() & 0x3ff));  // bf16 ~ +-0.01
    hipMemcpy(image, host.data(), host.size() * 2, hipMemcpyHostToDevice);
    printf("%s, %d CUs; every CU streams the same %.2f MB image\n", prop.name, cus, IMAGE_BYTES / 1e6);
#define RUN(W, S, N, M, C, D, SP, CH) run<W, S, N, M, C, D, SP, CH>(image, out, clk, cus);
    RUN(8, 32768, 4, 3, 1, 4, 0, 1)   // warm-up (discard)
    // bare MFMAs, + ds_read_b128 A fragments, + the 64 KiB / 256 samples weight stream: 8 waves x 1 column block ...
    RUN(8, 32768, 4, 3, 1, 4, 0, 1)
    RUN(8, 32768, 4, 2, 1, 4, 0, 1)
    RUN(8, 32768, 4, 1, 1, 4, 0, 1)
    RUN(8, 32768, 4, 1, 1, 4, 1, 1)
    RUN(8, 65536, 2, 1, 1, 4, 1, 1)
    // ... and 4 waves x 2 column blocks (one A fragment feeds two MFMAs)
    RUN(4, 32768, 4, 3, 2, 4, 0, 1)
    RUN(4, 32768, 4, 2, 2, 4, 0, 1)
    RUN(4, 32768, 4, 1, 2, 4, 0, 1)
    RUN(4, 32768, 4, 1, 2, 4, 1, 1)
    // the round-1 geometry: 4 waves x 1 column block (64 KiB per 128 samples)
    RUN(4, 65536, 2, 1, 1, 4, 1, 1)
    // the stream alone
    RUN(8, 32768, 4, 0, 1, 4, 0, 1)
    RUN(4, 65536, 2, 0, 1, 4, 0, 1)
    return 0;
}
