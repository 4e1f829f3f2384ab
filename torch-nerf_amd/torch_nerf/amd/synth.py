"""Deterministic synthetic inputs for benchmarks, fixtures and tests (numpy only).

There is no dataset on the GPU box, so the workload is synthetic Blender / LLFF
geometry: analytic orbit poses, a seeded pixel batch and
counter-generated MLP weights.  Everything here is a pure function of its integer
seed so that the golden-fixture script, the tests and bench.py agree bit for bit
without committing weight blobs.
"""
import math

import numpy as np

# state_dict order of the reference network (torch_nerf/src/network/nerf.py:49-59)
LAYER_NAMES = ("fc_in", "fc_1", "fc_2", "fc_3", "fc_4", "fc_5", "fc_6", "fc_7", "fc_8", "fc_9",
               "fc_out")

BLENDER_CAMERA_ANGLE_X = 0.6911112070083618  # NeRF-synthetic transforms_*.json


def layer_shapes(pos_dim=63, view_dir_dim=27, feat_dim=256):
    """(out, in) of the 11 Linear layers, nerf.py:49-59."""
    F = feat_dim
    ins = (pos_dim, F, F, F, F, F + pos_dim, F, F, F, F + view_dir_dim, F // 2)
    outs = (F, F, F, F, F, F, F, F, F + 1, F // 2, 3)
    return tuple(zip(outs, ins))


def param_count(pos_dim=63, view_dir_dim=27, feat_dim=256):
    return sum(o * i + o for o, i in layer_shapes(pos_dim, view_dir_dim, feat_dim))


def _mix(x):
    """splitmix64 finaliser on uint64 arrays (wrapping arithmetic)."""
    x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return x ^ (x >> np.uint64(31))


def counter_uniform(seed, stream, n):
    """n fp32 uniforms in [0,1) as a pure function of (seed, stream, index)."""
    with np.errstate(over="ignore"):
        base = _mix(np.uint64(seed) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(stream))
        idx = np.arange(n, dtype=np.uint64)
        bits = _mix(base + idx * np.uint64(0xD1342543DE82EF95) + np.uint64(1))
    return ((bits >> np.uint64(40)).astype(np.float64) * (1.0 / 16777216.0)).astype(np.float32)


def nerf_flat_params(seed=0, pos_dim=63, view_dir_dim=27, feat_dim=256, sigma_bias=0.0,
                     sigma_gain=1.0, gain=1.0):
    """Flat fp32 parameter blob in state_dict order (weight (out,in) then bias per layer).

    Values follow nn.Linear's default ranges U(-1/sqrt(in), 1/sqrt(in)) scaled by
    ``gain``; ``sigma_bias`` is added to fc_8.bias[0] so that densities are positive
    and transmittance actually decays (non-uniform CDF for the fine pass);
    ``sigma_gain`` scales the density row fc_8.weight[0] so density varies in space.
    """
    chunks = []
    for li, (o, i) in enumerate(layer_shapes(pos_dim, view_dir_dim, feat_dim)):
        bound = gain / math.sqrt(i)
        w = (counter_uniform(seed, 2 * li, o * i) * 2.0 - 1.0).astype(np.float32) * np.float32(bound)
        b = (counter_uniform(seed, 2 * li + 1, o) * 2.0 - 1.0).astype(np.float32) * np.float32(bound)
        if li == 8:
            b[0] += np.float32(sigma_bias)
            w[:i] *= np.float32(sigma_gain)
        chunks += [w.astype(np.float32), b.astype(np.float32)]
    return np.concatenate(chunks)


def split_flat_params(flat, pos_dim=63, view_dir_dim=27, feat_dim=256):
    """{'fc_in.weight': (out,in), 'fc_in.bias': (out,), ...} views of a flat blob."""
    out, off = {}, 0
    for name, (o, i) in zip(LAYER_NAMES, layer_shapes(pos_dim, view_dir_dim, feat_dim)):
        out[name + ".weight"] = flat[off:off + o * i].reshape(o, i)
        off += o * i
        out[name + ".bias"] = flat[off:off + o]
        off += o
    return out


def pose_spherical(theta_deg, phi_deg, radius):
    """Analytic orbit pose (4,4) fp32; math of utils/data/load_blender.py:78-109.

    Built step by step in fp32 matrices like the reference (each factor is an fp32
    tensor there), so the result is bit-identical to it.
    """
    def f32(m):
        return np.array(m, dtype=np.float32)

    th, ph = theta_deg / 180.0 * np.pi, phi_deg / 180.0 * np.pi
    trans = f32([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, radius], [0, 0, 0, 1]])
    rot_x = f32([[1, 0, 0, 0], [0, np.cos(ph), -np.sin(ph), 0], [0, np.sin(ph), np.cos(ph), 0],
                 [0, 0, 0, 1]])
    rot_y = f32([[np.cos(th), 0, -np.sin(th), 0], [0, 1, 0, 0], [np.sin(th), 0, np.cos(th), 0],
                 [0, 0, 0, 1]])
    flip = f32([[-1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]])
    c2w = rot_x @ trans
    c2w = rot_y @ c2w
    return (flip @ c2w).astype(np.float32)


def blender_focal(width, camera_angle_x=BLENDER_CAMERA_ANGLE_X):
    """focal = 0.5 W / tan(0.5 angle), utils/data/load_blender.py:170-171."""
    return 0.5 * width / np.tan(0.5 * camera_angle_x)


def blender_orbit_poses(num=40):
    """pose_spherical(theta, -30, 4) for theta in linspace(-180,180,num+1)[:-1] (load_blender.py:174-176)."""
    return [pose_spherical(float(t), -30.0, 4.0) for t in np.linspace(-180, 180, num + 1)[:-1]]


def llff_like_pose():
    """A 3x4 near-identity forward-facing pose (LLFF poses are 3x4, llff_dataset.py)."""
    a = 0.05
    R = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]], np.float32)
    t = np.array([[0.1], [-0.05], [0.2]], np.float32)
    return np.concatenate([R, t], axis=1).astype(np.float32)


def pixel_batch(seed, height, width, num_pixels):
    """A seeded sample of distinct pixel indices (int64), stand-in for np.random.choice."""
    rng = np.random.RandomState(seed)
    return rng.permutation(height * width)[:num_pixels].astype(np.int64)
