"""nr.save_obj / create_texture_image (reference: neural_renderer/save_obj.py:1-82) on the HIP kernels behind
neural_renderer.cuda.create_texture_image.  The atlas PNG is written with PIL (the reference uses skimage.io.imsave)."""
import os

import numpy as np
import torch


def create_texture_image(textures, texture_size_out=16):
    """textures [nf, ts, ts, ts, 3] -> (atlas image [H, W, 3] float numpy, flipped like the reference; vt [nf, 3, 2])."""
    import neural_renderer.cuda.create_texture_image as create_texture_image_cuda
    num_faces = textures.shape[0]
    tile_width = int((num_faces - 1.) ** 0.5) + 1
    tile_height = int((num_faces - 1.) / tile_width) + 1
    image = torch.zeros(tile_height * texture_size_out, tile_width * texture_size_out, 3, dtype=torch.float32,
                        device='cuda')
    vertices = torch.zeros((num_faces, 3, 2), dtype=torch.float32)
    face_nums = torch.arange(num_faces)
    column = (face_nums % tile_width).float()
    row = (face_nums // tile_width).float()
    vertices[:, 0, 0] = column * texture_size_out
    vertices[:, 0, 1] = row * texture_size_out
    vertices[:, 1, 0] = column * texture_size_out
    vertices[:, 1, 1] = (row + 1) * texture_size_out - 1
    vertices[:, 2, 0] = (column + 1) * texture_size_out - 1
    vertices[:, 2, 1] = (row + 1) * texture_size_out - 1
    vertices = vertices.cuda()
    image = create_texture_image_cuda.create_texture_image(vertices, textures.detach().float().contiguous().cuda(),
                                                           image, 1e-5)
    vertices[:, :, 0] /= (image.shape[1] - 1)
    vertices[:, :, 1] /= (image.shape[0] - 1)
    return image.cpu().numpy()[::-1, ::1], vertices.cpu().numpy()


def save_obj(filename, vertices, faces, textures=None):
    assert vertices.ndimension() == 2
    assert faces.ndimension() == 2
    if textures is not None:
        from PIL import Image
        filename_mtl = filename[:-4] + '.mtl'
        filename_texture = filename[:-4] + '.png'
        material_name = 'material_1'
        texture_image, vertices_textures = create_texture_image(textures)
        Image.fromarray((np.clip(texture_image, 0, 1) * 255).round().astype(np.uint8)).save(filename_texture)
    faces = faces.detach().cpu().numpy()
    with open(filename, 'w') as f:
        f.write('# %s\n#\n\n' % os.path.basename(filename))
        if textures is not None:
            f.write('mtllib %s\n\n' % os.path.basename(filename_mtl))
        for vertex in vertices.detach().cpu().numpy():
            f.write('v %.8f %.8f %.8f\n' % (vertex[0], vertex[1], vertex[2]))
        f.write('\n')
        if textures is not None:
            for vertex in vertices_textures.reshape((-1, 2)):
                f.write('vt %.8f %.8f\n' % (vertex[0], vertex[1]))
            f.write('\nusemtl %s\n' % material_name)
            for i, face in enumerate(faces):
                f.write('f %d/%d %d/%d %d/%d\n' % (face[0] + 1, 3 * i + 1, face[1] + 1, 3 * i + 2, face[2] + 1, 3 * i + 3))
            f.write('\n')
        else:
            for face in faces:
                f.write('f %d %d %d\n' % (face[0] + 1, face[1] + 1, face[2] + 1))
    if textures is not None:
        with open(filename_mtl, 'w') as f:
            f.write('newmtl %s\n' % material_name)
            f.write('map_Kd %s\n' % os.path.basename(filename_texture))
