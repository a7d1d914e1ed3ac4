"""`neural_renderer.cuda.load_textures` — never reached on the hot path (network.py:108 uses load_texture=False)."""


def load_textures(*a, **k):
    raise NotImplementedError('load_textures (OBJ-with-texture baking) is out of scope of the hot-path build')
