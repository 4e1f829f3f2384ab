"""How fast is bench.py's CPU baseline (oracle/torch_port.py, kind "port") next to the reference it stands in for?
Build container only (the reference never travels): the imported reference's two render_scene calls vs the port's
render_batch on the same 1024 rays, 8 threads.  VERDICT r05 weak 4(b): the port skips the reference's per-call validation
and lambda lists, so it is FASTER -- the speed-up bench.py reports against it is conservative.
    python scripts/port_vs_reference.py   ->  profiles/r06_port_vs_reference.json"""
import importlib.util, json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("nerf_synth", os.path.join(ROOT, "torch-nerf_amd", "torch_nerf", "amd", "synth.py"))
synth = importlib.util.module_from_spec(spec); spec.loader.exec_module(synth)
sys.path.insert(0, ROOT)
from oracle import torch_port as TP                                         # noqa: E402
sys.path.insert(0, os.environ.get("NERF_REFERENCE", "/root/reference"))
import torch_nerf.src.renderer.cameras as cameras                           # noqa: E402  (the REFERENCE's modules)
import torch_nerf.src.renderer.integrators.quadrature_integrator as integ   # noqa: E402
import torch_nerf.src.renderer.ray_samplers as samplers                     # noqa: E402
import torch_nerf.src.renderer.volume_renderer as vr                        # noqa: E402
import torch_nerf.src.network.nerf as nerf                                  # noqa: E402
import torch_nerf.src.scene as scene                                        # noqa: E402
from torch_nerf.src.signal_encoder.positional_encoder import PositionalEncoder  # noqa: E402
assert vr.__file__.startswith("/root/reference")
torch.set_num_threads(8)
n, H, W = 1024, 800, 800
focal = float(synth.blender_focal(W)); pose = torch.from_numpy(synth.pose_spherical(37.0, -30.0, 4.0))
flats = [synth.nerf_flat_params(seed=s, sigma_bias=1.0, sigma_gain=30.0) for s in (3, 4)]
pix = torch.from_numpy(synth.pixel_batch(0, H, W, n))
cam = cameras.PerspectiveCamera({"f_x": focal, "f_y": focal, "img_width": W, "img_height": H}, pose, 2.0, 6.0)
renderer = vr.VolumeRenderer(integ.QuadratureIntegrator(), samplers.StratifiedSampler(), cam)
enc = {"coord_enc": PositionalEncoder(3, 10, True), "dir_enc": PositionalEncoder(3, 4, True)}
scenes = []
for f in flats:
    net = nerf.NeRF(63, 27)
    net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.split_flat_params(f).items()})
    scenes.append(scene.PrimitiveCube(net, enc))
params = [{k: torch.from_numpy(v.copy()) for k, v in synth.split_flat_params(f).items()} for f in flats]
draws = tuple(torch.rand(n, s) for s in (64, 64, 128, 128))
def ref():
    with torch.no_grad():
        c, idx, w = renderer.render_scene(scenes[0], n, 64, False, "cpu", pixel_indices=pix)
        return renderer.render_scene(scenes[1], n, (64, 128), False, "cpu", pixel_indices=idx, weights=w)[0]
def port():
    with torch.no_grad():
        return TP.render_batch(params[0], params[1], pix, H, W, focal, pose, 2.0, 6.0, 64, 128, draws)[2]
ref(); port()
tr, tp = [], []
for _ in range(7):                      # alternating, so that a drift of the shared host hits both alike
    t0 = time.perf_counter(); ref(); tr.append(time.perf_counter() - t0)
    t0 = time.perf_counter(); port(); tp.append(time.perf_counter() - t0)
t_ref, t_port = sorted(tr)[3], sorted(tp)[3]
out = {"rays": n, "threads": 8, "reference_rays_per_s": n / t_ref, "port_rays_per_s": n / t_port, "port_vs_reference": t_ref / t_port,
       "reference_rays_per_s_range": [n / max(tr), n / min(tr)], "port_rays_per_s_range": [n / max(tp), n / min(tp)],
       "where": "build container (8 CPUs), torch " + torch.__version__,
       "what": "coarse 64 + fine 64+128 forward, no_grad; 7 alternating passes each, medians"}
json.dump(out, open(os.path.join(ROOT, "profiles", "r06_port_vs_reference.json"), "w"), indent=1)
print(out)
