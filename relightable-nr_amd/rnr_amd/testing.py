"""Names the tests use for the seeded synthetic scene; the generators live in rnr_amd.scene / rnr_amd.rays (nothing in the
product package imports this module)."""
from .rays import ray_pivots  # noqa: F401
from .scene import synthetic_light_probe, synthetic_textures, tiny_scene, unet_state_dict  # noqa: F401
