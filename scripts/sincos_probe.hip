#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <math.h>
__global__ void k(const float *x, int n, unsigned long long *diff) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s, c;
    sincosf(x[i], &s, &c);
    float s2 = sinf(x[i]), c2 = cosf(x[i]);
    if (__float_as_uint(s) != __float_as_uint(s2) || __float_as_uint(c) != __float_as_uint(c2)) atomicAdd(diff, 1ull);
}
int main() {
    const int n = 1 << 24;
    float *h = (float *)malloc(n * 4);
    uint64_t st = 88172645463325252ull;
    for (int i = 0; i < n; ++i) {
        st ^= st << 13; st ^= st >> 7; st ^= st << 17;
        double u = (st >> 11) * (1.0 / 9007199254740992.0);
        int f = i % 16;                       // octaves 2^0 .. 2^15 of a coordinate in [-6, 6]
        h[i] = ldexpf((float)(u * 12.0 - 6.0), f);
    }
    float *d; unsigned long long *dd, hd = 0;
    hipMalloc(&d, n * 4); hipMalloc(&dd, 8);
    hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice); hipMemcpy(dd, &hd, 8, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(d, n, dd);
    hipMemcpy(&hd, dd, 8, hipMemcpyDeviceToHost);
    printf("sincosf vs sinf/cosf: %llu of %d differ\n", hd, n);
    return 0;
}
