"""MI355X-native backend of the torch_nerf renderer: ctypes binding of libnerf_amd.so
(include/nerf_amd.h), torch.autograd.Function wrappers and the ray-shard helpers."""
