// a13 (MLP part): parameter gradients.  Placeholder until the MFMA backward lands.
#include "common.h"
#include "mlp_layout.h"

NERF_API int64_t nerf_mlp_backward_workspace_bytes(int64_t M) { (void)M; return 0; }

NERF_API int nerf_mlp_backward(const void *, const float *, const float *, const float *, int64_t, int,
                               const float *, const float *, const void *, const float *, const float *,
                               float *, void *, nerf_stream_t) {
    return nerf::fail(NERF_ERR_UNSUPPORTED, "nerf_mlp_backward: not implemented yet");
}
