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


# ---------------------------------------------------------------- LLFF render poses
# Pure functions of (poses, depth bounds): what utils/data/load_llff.py computes between loading the images
# and returning `render_poses`.  The image / COLMAP loading around them stays out of scope.
def unit(vec):
    """vec / |vec| (load_llff.py:213-226)."""
    vec = np.asarray(vec)
    return vec / np.linalg.norm(vec)


def build_extrinsic(z_vec, up_vec, camera_position):
    """(3,4) camera-to-world [x | y | z | position] from a viewing axis and an up vector (load_llff.py:229-258):
    x = unit(up x z), y = unit(z x x)."""
    z = unit(z_vec)
    x = unit(np.cross(up_vec, z))
    y = unit(np.cross(z, x))
    return np.stack([x, y, z, np.asarray(camera_position)], axis=1)


def poses_avg(poses):
    """The "central" pose of a set of (N,3,4) poses: mean position, summed z axes, summed y axes as the up
    vector (load_llff.py:281-307)."""
    poses = np.asarray(poses)
    return build_extrinsic(unit(poses[:, :3, 2].sum(axis=0)), poses[:, :3, 1].sum(axis=0),
                           poses[:, :3, 3].mean(axis=0))


def recenter_poses(poses):
    """Express every pose in the frame of the central pose (load_llff.py:353-376)."""
    poses = np.asarray(poses)
    out = poses + 0
    last_row = np.array([[0.0, 0.0, 0.0, 1.0]])
    centre = np.concatenate([poses_avg(poses), last_row], axis=0)
    full = np.concatenate([poses[:, :3, :4], np.broadcast_to(last_row, (poses.shape[0], 1, 4))], axis=1)
    out[:, :3, :4] = (np.linalg.inv(centre) @ full)[:, :3, :4]
    return out


def render_path_spiral(camera_to_world, up_vec, radiuses, focal, z_rate, rots, num_keyframe):
    """Spiral of (3,4) poses around `camera_to_world`, every one looking at the point `focal` in front of it
    (load_llff.py:310-350).  Returns a list, like the reference."""
    c2w = np.asarray(camera_to_world)[:3, :4]
    radii = np.array(list(radiuses) + [1.0])
    target = c2w @ np.array([0.0, 0.0, -focal, 1.0])
    poses = []
    for theta in np.linspace(0.0, 2.0 * np.pi * rots, num_keyframe + 1)[:-1]:
        position = c2w @ (np.array([np.cos(theta), -np.sin(theta), -np.sin(theta * z_rate), 1.0]) * radii)
        poses.append(build_extrinsic(unit(position - target), up_vec, position))
    return poses


def llff_spiral_poses(extrinsics, z_bounds, path_zflat=False):
    """`render_poses` exactly as load_llff_data derives them from the (already rescaled / recentred) poses and
    depth bounds (load_llff.py:519-559): focus depth from the bounds, radii = 90th percentile of |position|,
    120 key frames over 2 rotations (60 over 1, z-radius 0, for path_zflat).  fp32 (N,3,4)."""
    extrinsics, z_bounds = np.asarray(extrinsics), np.asarray(z_bounds)
    centre = poses_avg(extrinsics)
    up = unit(extrinsics[:, :, 1].sum(0))
    close_depth, inf_depth = z_bounds.min() * 0.9, z_bounds.max() * 5.0
    dt = 0.75
    focal = 1.0 / ((1.0 - dt) / close_depth + dt / inf_depth)
    radii = np.percentile(np.abs(extrinsics[:, :, 3]), 90, 0)
    keyframes, rotations = 120, 2
    if path_zflat:
        centre[:3, 3] = centre[:3, 3] + (-close_depth * 0.1) * centre[:3, 2]
        radii[2] = 0.0
        keyframes, rotations = 60, 1
    return np.array(render_path_spiral(centre, up, radii, focal, z_rate=0.5, rots=rotations,
                                       num_keyframe=keyframes)).astype(np.float32)


def llff_like_pose_set(num=20, seed=0):
    """(poses (num,3,4) float64, z_bounds (num,2)): a forward-facing capture like an LLFF scene -- cameras on a
    jittered grid one unit in front of the z = 0 plane, all looking roughly down -z -- from the build's counter
    generator.  (One unit away: the reference's NDC mapping divides by the origin's z, sampler_base.py:199-257.)"""
    u = counter_uniform(seed, 77, num * 8).astype(np.float64).reshape(num, 8)
    poses = []
    for k in range(num):
        position = np.array([(u[k, 0] - 0.5) * 2.0, (u[k, 1] - 0.5) * 1.2, 1.0 + (u[k, 2] - 0.5) * 0.3])
        look = np.array([(u[k, 3] - 0.5) * 0.2, (u[k, 4] - 0.5) * 0.2, 1.0])      # camera z axis points backwards
        poses.append(build_extrinsic(look, np.array([(u[k, 5] - 0.5) * 0.1, 1.0, 0.0]), position))
    z_bounds = np.stack([1.2 + u[:, 6], 8.0 + 6.0 * u[:, 7]], axis=1)
    return np.array(poses), z_bounds


def pixel_batch(seed, height, width, num_pixels):
    """A seeded sample of distinct pixel indices (int64), stand-in for np.random.choice."""
    rng = np.random.RandomState(seed)
    return rng.permutation(height * width)[:num_pixels].astype(np.int64)
