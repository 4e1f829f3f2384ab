// What is the streaming-store ceiling of this GPU?  The split-f16 record forward and reverse chain write 8 - 10 KB per sample
// and run at 3.1 TB/s: store-bound, or short of what stores can do?  Pure store kernels, 8 GiB per launch, patterns:
//   run1k : a wavefront store instruction writes 1 KiB contiguous (global_store_dwordx4, lane * 16)
//   run512: two 512-byte runs 4 KiB apart per instruction (what Recorder::store of mlp_forward_f16x2.hip issues)
//   run1k nt / sc1: the same with the nontemporal / system-coherent cache policy bits
//   copy  : read 1 KiB + write 1 KiB per instruction pair (for reference: HBM read + write mix)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -w scripts/hbm_write_probe.hip -o scripts/hbm_write_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256) void writer(char *out, const char *in, size_t bytes) {
    const size_t wave_global = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6, waves = (size_t)gridDim.x * 4;
    const unsigned lane = threadIdx.x & 63;
    const f32x4 v = {(float)lane, 1.0f, 2.0f, (float)blockIdx.x};
    for (size_t kb = wave_global; kb < bytes / 1024; kb += waves) {
        char *p = out + kb * 1024 + lane * 16;
        if (MODE == 1) {                    // lanes 0..31 write half-KiB unit k of an 8 KiB group, lanes 32..63 unit k + 8 (4 KiB further)
            const size_t grp = kb >> 3, k = kb & 7;
            p = out + grp * 8192 + (k + 8 * (lane >> 5)) * 512 + (lane & 31) * 16;
        }
        if (MODE == 0 || MODE == 1) asm volatile("global_store_dwordx4 %0, %1, off" : : "v"(p), "v"(v) : "memory");
        else if (MODE == 2) asm volatile("global_store_dwordx4 %0, %1, off nt" : : "v"(p), "v"(v) : "memory");
        else if (MODE == 3) asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
        else if (MODE == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(p), "v"(v) : "memory");
        else {
            f32x4 r;
            asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(in + kb * 1024 + lane * 16) : "memory");
            asm volatile("global_store_dwordx4 %0, %1, off" : : "v"(p), "v"(r) : "memory");
        }
    }
}
template <int MODE>
static void run(const char *name, char *out, char *in, size_t bytes, int blocks) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e30f;
    for (int t = 0; t < 4; ++t) {
        hipEventRecord(a);
        hipLaunchKernelGGL(writer<MODE>, dim3(blocks), dim3(256), 0, 0, out, in, bytes);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (t && ms < best) best = ms;
    }
    const double moved = MODE == 5 ? 2.0 * bytes : (double)bytes;
    printf("%-22s blocks %5d  %8.3f ms  %7.1f GB/s%s\n", name, blocks, best, moved / best / 1e6, MODE == 5 ? "  (read + write)" : "");
}
int main() {
    const size_t bytes = (size_t)8 << 30;
    char *out, *in;
    if (hipMalloc(&out, bytes) != hipSuccess || hipMalloc(&in, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(in, 1, bytes);
    for (int blocks : {256, 1024, 4096}) {
        run<0>("run1k", out, in, bytes, blocks);
        run<1>("run512 x 2", out, in, bytes, blocks);
        run<2>("run1k nt", out, in, bytes, blocks);
        run<3>("run1k sc1", out, in, bytes, blocks);
        run<4>("run1k sc0 sc1", out, in, bytes, blocks);
        run<5>("copy", out, in, bytes, blocks);
    }
    return 0;
}
