// nerf_net_t (include/nerf_amd.h) -> the kernels' view of the network instance.
#pragma once
#include "common.h"
#include "mlp_layout.h"

namespace nerf {

// The description the ABI call works on; NULL = the reference's shipped yaml (63 / 27 / 256, levels 10 / 4,
// include_input).  Returns NERF_PATH_FUSED / NERF_PATH_LAYERED, or -1 with the error message set.
inline int net_describe(const nerf_net_t *abi, nerf_net_t &out) {
    static const nerf_net_t shipped = {63, 27, 256, 10, 1, 4, 1};
    out = abi ? *abi : shipped;
    if (out.pos_dim < 1 || out.view_dir_dim < 1 || out.feat_dim < 2) {
        fail(NERF_ERR_ARG, "nerf_net_t: pos_dim, view_dir_dim >= 1 and feat_dim >= 2 expected");
        return -1;
    }
    // a declared PositionalEncoder(3, L, include_input) must produce the declared width (positional_encoder.py:43-47)
    if ((out.pos_levels >= 0 && out.pos_dim != 6 * out.pos_levels + (out.pos_include_input ? 3 : 0)) ||
        (out.dir_levels >= 0 && out.view_dir_dim != 6 * out.dir_levels + (out.dir_include_input ? 3 : 0))) {
        fail(NERF_ERR_ARG, "nerf_net_t: encode levels / include_input do not give pos_dim / view_dir_dim");
        return -1;
    }
    return (out.feat_dim == mlp::FEAT && out.pos_dim <= mlp::MAX_E_POS && out.view_dir_dim <= mlp::MAX_E_DIR)
               ? NERF_PATH_FUSED : NERF_PATH_LAYERED;
}

// For the entry points of the fused family: fills `net`, or returns an error code.
inline int fused_net(const nerf_net_t *abi, mlp::Net &net, const char *who) {
    nerf_net_t d;
    const int path = net_describe(abi, d);
    if (path < 0) return NERF_ERR_ARG;
    if (path != NERF_PATH_FUSED) {
        snprintf(error_buffer(), 256, "%s: NeRF(%d, %d, %d) is outside the fused kernels (feat_dim 256, pos_dim <= 64, "
                 "view_dir_dim <= 32): use nerf_mlp_layered_*", who, d.pos_dim, d.view_dir_dim, d.feat_dim);
        return NERF_ERR_UNSUPPORTED;
    }
    net.e_pos = d.pos_dim; net.e_dir = d.view_dir_dim;
    net.l_pos = d.pos_levels; net.l_dir = d.dir_levels;
    net.inc_pos = d.pos_include_input ? 1 : 0; net.inc_dir = d.dir_include_input ? 1 : 0;
    return NERF_OK;
}

inline bool raw_inputs_ok(const mlp::Net &net) { return net.l_pos >= 0 && net.l_dir >= 0; }

// For the split-f16 entries: feat_dim 256 with up to four position and two direction k-blocks (the fused family and the
// wider inputs of coord_encode_level <= 20 / dir_encode_level <= 10).
inline int f16x2_net(const nerf_net_t *abi, mlp::Net &net, const char *who) {
    nerf_net_t d;
    if (net_describe(abi, d) < 0) return NERF_ERR_ARG;
    if (d.feat_dim != mlp::FEAT || d.pos_dim > mlp::F2_MAX_E_POS || d.view_dir_dim > mlp::F2_MAX_E_DIR) {
        snprintf(error_buffer(), 256, "%s: NeRF(%d, %d, %d) is outside the split-f16 kernel (feat_dim 256, pos_dim <= 128, "
                 "view_dir_dim <= 64)", who, d.pos_dim, d.view_dir_dim, d.feat_dim);
        return NERF_ERR_UNSUPPORTED;
    }
    net.e_pos = d.pos_dim; net.e_dir = d.view_dir_dim;
    net.l_pos = d.pos_levels; net.l_dir = d.dir_levels;
    net.inc_pos = d.pos_include_input ? 1 : 0; net.inc_dir = d.dir_include_input ? 1 : 0;
    return NERF_OK;
}

}  // namespace nerf
