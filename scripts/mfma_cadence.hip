#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
// MODE 4: MODE 1 + an LDS-DMA block (M0 save/set/restore) every 2 groups; MODE 5: same without M0 save/restore
// MODE 0: pure MFMA, 8 accumulators, groups of 4 dependent; MODE 1: + one ds_read_b128 per group (compiler placed)
// MODE 2: groups of 4 on the SAME acc for all (fully dependent chain); MODE 3: 16 dependent per acc (fb-major)
template <int MODE>
__global__ __launch_bounds__(256, 1) void k(float *out, unsigned long long *cyc, const float *in) {
    __shared__ __attribute__((aligned(16))) float lds[8192 + 4096];
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = in[i];
    __syncthreads();
    f32x16 acc[8];
    for (int f = 0; f < 8; ++f) for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;
    float b[16];
    for (int r = 0; r < 16; ++r) b[r] = in[threadIdx.x + r];
    f32x4 a = {in[threadIdx.x], in[threadIdx.x + 1], in[threadIdx.x + 2], in[threadIdx.x + 3]};
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < 64; ++it) {
#pragma unroll
        for (int g = 0; g < 32; ++g) {
            const int q = g / 8, fb = MODE == 2 ? 0 : (MODE == 3 ? g / 4 : g % 8);
            if ((MODE == 4 || MODE == 5) && (g & 1) == 0) {
                const char *src = reinterpret_cast<const char *>(in) + ((it * 32 + g) & 31) * 1024 + threadIdx.x % 64 * 16;
                unsigned dst = 16384 + (g & 15) * 1024; unsigned keep;
                if (MODE == 4) asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
                else asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(src), "s"(dst) : "memory");
            }
            if (MODE == 1 || MODE == 4 || MODE == 5) a = *reinterpret_cast<const f32x4 *>(&lds[((g * 64 + threadIdx.x) * 4) & 8191]);
            acc[fb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b[4 * q + 0], acc[fb], 0, 0, 0);
            acc[fb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b[4 * q + 1], acc[fb], 0, 0, 0);
            acc[fb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b[4 * q + 2], acc[fb], 0, 0, 0);
            acc[fb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b[4 * q + 3], acc[fb], 0, 0, 0);
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0; for (int f = 0; f < 8; ++f) for (int r = 0; r < 16; ++r) s += acc[f][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[MODE] = t1 - t0;
}
int main() {
    float *out, *in; unsigned long long *cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&in, 65536 * 4); hipMalloc(&cyc, 64); hipMemset(in, 0, 65536 * 4); hipMemset(cyc, 0, 64);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, out, cyc, in);
        hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, out, cyc, in);
        hipLaunchKernelGGL(k<2>, dim3(256), dim3(256), 0, 0, out, cyc, in);
        hipLaunchKernelGGL(k<3>, dim3(256), dim3(256), 0, 0, out, cyc, in);
        hipLaunchKernelGGL(k<4>, dim3(256), dim3(256), 0, 0, out, cyc, in);
        hipLaunchKernelGGL(k<5>, dim3(256), dim3(256), 0, 0, out, cyc, in);
        hipDeviceSynchronize();
    }
    unsigned long long h[8]; hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
    for (int m = 0; m < 6; ++m) printf("mode %d: %.2f cycles per MFMA\n", m, (double)h[m] / (64.0 * 128));
    return 0;
}
