// Probe for DESIGN 4.4 (round 6): is the f16 matrix pipe good enough to carry fp32 operands split in two f16 parts?
//   1. subnormal f16 A / B inputs of v_mfma_f32_32x32x16_f16: flushed or kept?  (the low part of a value below 2^-3 is an
//      f16 subnormal; scripts/split_emulate.py shows the arrangement only works if they are kept)
//   2. accumulation: K = 256 (a whole layer), three part products per k-step chained into one fp32 accumulator, against
//      the exact (fp64) dot product of the fp32 operands and against the fp32 MFMA chain (v_mfma_f32_32x32x2_f32)
//   3. the conversion idiom: (_Float16) casts must round to nearest even and the residual x - hi must be exact
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -w scripts/f16_split_probe.hip -o scripts/f16_split_probe.bin
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// A (32 x K) row-major, B (K x 32) stored as Bt (32 x K) row-major (column j of B = row j of Bt), out 32 x 32
__global__ void split_dot(const float *A, const float *Bt, int K, float scaleA, float *out_split, float *out_f32, float *out_plain) {
    const int lane = threadIdx.x, i = lane & 31, h = lane >> 5;
    f32x16 acc, acc32, accp;
    for (int r = 0; r < 16; ++r) acc[r] = acc32[r] = accp[r] = 0.f;
    for (int k0 = 0; k0 < K; k0 += 16) {
        f16x8 ah, al, bh, bl;
        for (int e = 0; e < 8; ++e) {
            const float a = A[i * K + k0 + 8 * h + e] * scaleA, b = Bt[i * K + k0 + 8 * h + e];
            ah[e] = (_Float16)a;
            al[e] = (_Float16)(a - (float)ah[e]);
            bh[e] = (_Float16)b;
            bl[e] = (_Float16)(b - (float)bh[e]);
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
        accp = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, accp, 0, 0, 0);
        for (int kk = 0; kk < 16; kk += 2)
            acc32 = __builtin_amdgcn_mfma_f32_32x32x2f32(A[i * K + k0 + kk + h] * scaleA, Bt[i * K + k0 + kk + h], acc32, 0, 0, 0);
    }
    for (int r = 0; r < 16; ++r) {
        const int row = (r >> 2) * 8 + h * 4 + (r & 3);
        out_split[row * 32 + i] = acc[r] / scaleA;
        out_f32[row * 32 + i] = acc32[r] / scaleA;
        out_plain[row * 32 + i] = accp[r] / scaleA;
    }
}

// one product a * b at k = 0 (everything else zero) through the f16 MFMA
__global__ void one_product(float a, float b, float *out) {
    const int lane = threadIdx.x;
    f16x8 av, bv;
    for (int e = 0; e < 8; ++e) av[e] = bv[e] = (_Float16)0.f;
    if (lane < 32) { av[0] = (_Float16)a; bv[0] = (_Float16)b; }
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc, 0, 0, 0);
    if (lane == 0) out[0] = acc[0];
}

__global__ void split_check(const float *x, int n, float *hi, float *lo) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const _Float16 h = (_Float16)x[t];
    const _Float16 l = (_Float16)(x[t] - (float)h);
    hi[t] = (float)h;
    lo[t] = (float)l;
}

int main() {
    // ---- 1. subnormals
    float *out;
    hipMalloc(&out, 4096 * 4);
    struct { float a, b; const char *what; } cases[] = {
        {ldexpf(1.f, -20), 1.0f, "subnormal A (2^-20) x 1"},
        {1.0f, ldexpf(1.f, -20), "1 x subnormal B (2^-20)"},
        {ldexpf(1.f, -24), 1.0f, "smallest subnormal A (2^-24) x 1"},
        {ldexpf(1.f, -20), ldexpf(1.f, -20), "subnormal x subnormal (2^-40)"},
        {ldexpf(1.f, -14), ldexpf(1.f, -14), "min normal squared (2^-28)"},
    };
    for (auto &c : cases) {
        hipLaunchKernelGGL(one_product, dim3(1), dim3(64), 0, 0, c.a, c.b, out);
        float got;
        hipMemcpy(&got, out, 4, hipMemcpyDeviceToHost);
        printf("subnormal  %-36s got %.9g  want %.9g  %s\n", c.what, got, c.a * c.b, got == c.a * c.b ? "KEPT" : "FLUSHED/DIFFERENT");
    }
    // ---- 3. conversion idiom
    {
        const int n = 1 << 16;
        std::vector<float> x(n), hi(n), lo(n);
        srand(3);
        for (int t = 0; t < n; ++t) x[t] = ldexpf((float)rand() / RAND_MAX * 2.f - 1.f, rand() % 12 - 8);
        float *dx, *dh, *dl;
        hipMalloc(&dx, n * 4); hipMalloc(&dh, n * 4); hipMalloc(&dl, n * 4);
        hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(split_check, dim3(n / 256), dim3(256), 0, 0, dx, n, dh, dl);
        hipMemcpy(hi.data(), dh, n * 4, hipMemcpyDeviceToHost);
        hipMemcpy(lo.data(), dl, n * 4, hipMemcpyDeviceToHost);
        double worst = 0; int bad_rne = 0;
        for (int t = 0; t < n; ++t) {
            const double rel = fabs((double)x[t] - hi[t] - lo[t]) / fmax(fabs((double)x[t]), ldexp(1.0, -3));
            if (rel > worst) worst = rel;
            if ((float)(_Float16)x[t] != hi[t]) ++bad_rne;
        }
        printf("split      max |x - hi - lo| / max(|x|, 2^-3) = %.3g (2^-22 = %.3g)  host/device hi mismatches %d\n", worst, ldexp(1.0, -22), bad_rne);
    }
    // ---- 2. K = 256 accumulation
    for (int trial = 0; trial < 3; ++trial) {
        const int K = 256;
        std::vector<float> A(32 * K), Bt(32 * K), o1(1024), o2(1024), o3(1024);
        srand(11 + trial);
        for (auto &v : A) v = ((float)rand() / RAND_MAX * 2.f - 1.f) / 16.f;
        for (auto &v : Bt) { v = ((float)rand() / RAND_MAX * 2.f - 0.8f); if (trial != 1 && v < 0) v = 0.f; if (trial == 2) v *= 0.05f; }
        float *dA, *dB, *d1, *d2, *d3;
        hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, Bt.size() * 4); hipMalloc(&d1, 4096); hipMalloc(&d2, 4096); hipMalloc(&d3, 4096);
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dB, Bt.data(), Bt.size() * 4, hipMemcpyHostToDevice);
        for (float scale : {1.0f, 131072.0f}) {
            hipLaunchKernelGGL(split_dot, dim3(1), dim3(64), 0, 0, dA, dB, K, scale, d1, d2, d3);
            hipMemcpy(o1.data(), d1, 4096, hipMemcpyDeviceToHost);
            hipMemcpy(o2.data(), d2, 4096, hipMemcpyDeviceToHost);
            hipMemcpy(o3.data(), d3, 4096, hipMemcpyDeviceToHost);
            double e1 = 0, e2 = 0, e3 = 0, b1 = 0, b2 = 0;
            for (int r = 0; r < 32; ++r)
                for (int c = 0; c < 32; ++c) {
                    double exact = 0, mag = 0;
                    for (int k = 0; k < K; ++k) { exact += (double)A[r * K + k] * Bt[c * K + k]; mag += fabs((double)A[r * K + k] * Bt[c * K + k]); }
                    e1 = fmax(e1, fabs(o1[r * 32 + c] - exact) / mag);
                    e2 = fmax(e2, fabs(o2[r * 32 + c] - exact) / mag);
                    e3 = fmax(e3, fabs(o3[r * 32 + c] - exact) / mag);
                    b1 += (o1[r * 32 + c] - exact) / mag; b2 += (o2[r * 32 + c] - exact) / mag;
                }
            printf("K=256 trial %d weight scale %-8g max err / sum|ab|: split-f16x2/3p %.3g (mean signed %.2g)  fp32 MFMA %.3g (mean signed %.2g)  plain f16 %.3g\n",
                   trial, scale, e1, b1 / 1024, e2, b2 / 1024, e3);
        }
    }
    return 0;
}
