"""Row f3: LLFF render-pose generators (utils/data/load_llff.py:213-376, :519-559) against golden F10, which
tests/golden/make_golden.py captured from the reference's own functions on a synthetic forward-facing pose set."""
import numpy as np

from torch_nerf.amd import synth


def test_pose_set_generator_is_what_the_golden_was_made_from(golden):
    g = golden("f10_llff_poses")
    poses, z_bounds = synth.llff_like_pose_set(20, seed=0)
    assert np.array_equal(poses, g["poses"]) and np.array_equal(z_bounds, g["z_bounds"])


def test_build_extrinsic_and_poses_avg(golden):
    g = golden("f10_llff_poses")
    probe = synth.build_extrinsic(np.array([0.1, -0.2, 0.9]), np.array([0.05, 1.0, 0.0]), np.array([1.0, 2.0, 3.0]))
    np.testing.assert_allclose(probe, g["extrinsic_probe"], rtol=0, atol=1e-12)
    rot = probe[:, :3]
    np.testing.assert_allclose(rot.T @ rot, np.eye(3), atol=1e-12)           # orthonormal frame
    np.testing.assert_allclose(synth.poses_avg(g["recentred"]), g["poses_avg"], rtol=0, atol=1e-12)


def test_recenter_poses(golden):
    g = golden("f10_llff_poses")
    got = synth.recenter_poses(g["poses"])
    np.testing.assert_allclose(got, g["recentred"], rtol=0, atol=1e-12)
    # the central pose of recentred poses is the identity frame at the origin
    np.testing.assert_allclose(synth.poses_avg(got), np.eye(4)[:3], atol=1e-9)


def test_spiral_paths(golden):
    g = golden("f10_llff_poses")
    for zflat, key in ((False, "spiral"), (True, "spiral_zflat")):
        got = synth.llff_spiral_poses(g["recentred"], g["z_bounds"], path_zflat=zflat)
        assert got.dtype == np.float32 and got.shape == g[key].shape == ((60 if zflat else 120), 3, 4)
        np.testing.assert_allclose(got, g[key], rtol=0, atol=1e-6)
    flat = synth.llff_spiral_poses(g["recentred"], g["z_bounds"], path_zflat=True)
    centre = synth.poses_avg(g["recentred"])
    # z-flat: positions stay in the plane through the (shifted) centre spanned by its x and y axes
    off = flat[:, :, 3] - flat[:, :, 3].mean(0)
    assert np.abs(off @ centre[:, 2]).max() < 1e-5
