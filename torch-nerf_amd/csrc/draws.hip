// Counter-based uniform draws for the ray-sharded paths (torch_nerf/amd/shard.py, train.py).
//
// The reference draws its jitter with torch.rand / rand_like inside the sampler
// (ray_samplers/stratified_sampler.py:77, :109; ray_samplers/utils.py:43, :56): a stream that depends
// on how many rays the process renders.  A frame or a training batch that is cut across GPUs needs draws
// that are a pure function of (key, global element index) instead, so that the picture does not depend on
// the number of GPUs; this kernel evaluates that function -- the splitmix64 finaliser of
// torch_nerf.amd.synth.counter_uniform, bit for bit -- in one launch (as torch tensor ops it is ~20
// launches per call, four calls per pass).  HBM-bound: 4 B written per draw.
#include "common.h"

namespace {

__global__ void counter_uniform_kernel(uint64_t key, int64_t first, int64_t count, float *__restrict__ out) {
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < count;
         g += (int64_t)gridDim.x * blockDim.x) {
        uint64_t x = (uint64_t)(first + g) * 0xD1342543DE82EF95ull + key;
        x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
        x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
        x ^= x >> 31;
        out[g] = (float)(x >> 40) * (1.0f / 16777216.0f);   // 24 random bits: exact in fp32, in [0, 1)
    }
}

}  // namespace

NERF_API int nerf_counter_uniform(uint64_t key, int64_t first, int64_t count, float *out, nerf_stream_t stream) {
    NERF_REQUIRE(count >= 0 && first >= 0, "nerf_counter_uniform: negative range");
    if (count == 0) return NERF_OK;
    NERF_REQUIRE(out, "nerf_counter_uniform: null pointer");
    int64_t grid = (count + 255) / 256;
    if (grid > 256 * 16) grid = 256 * 16;
    hipLaunchKernelGGL(counter_uniform_kernel, dim3((unsigned)grid), dim3(256), 0, nerf::as_stream(stream), key, first,
                       count, out);
    return nerf::check_launch("nerf_counter_uniform");
}
