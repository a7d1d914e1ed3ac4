"""`neural_renderer.cuda.create_texture_image` — export utility of save_obj only; out of scope."""


def create_texture_image(*a, **k):
    raise NotImplementedError('create_texture_image (save_obj atlas baking) is out of scope of the hot-path build')
