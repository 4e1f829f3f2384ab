"""MI355X-native drop-in for the `torch_nerf` package of DveloperY0115/torch-NeRF.

Put the directory that contains this package (``torch-nerf_amd/``) on ``sys.path``
instead of the reference checkout: the runners' imports (``torch_nerf.src.renderer...``,
``torch_nerf.src.scene``, ``torch_nerf.src.network``, ``torch_nerf.src.signal_encoder``)
resolve here, with the same class names, signatures, return conventions and error
behaviour, and the work runs in hand-written gfx950 kernels (``torch_nerf.amd``).
"""
