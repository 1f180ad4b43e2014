"""OBJ I/O and the Mesh container (neural_renderer/load_obj.py:95-141, save_obj.py:4-18, mesh.py:7-38)."""
import os

import numpy as np
import torch


def _parse_obj(filename_obj):
    vertices, faces = [], []
    with open(filename_obj) as f:
        for line in f:
            tok = line.split()
            if not tok:
                continue
            if tok[0] == 'v':
                vertices.append([float(v) for v in tok[1:4]])
            elif tok[0] == 'f':
                # polygons are fan-triangulated around their first vertex (load_obj.py:117-123)
                ids = [int(v.split('/')[0]) for v in tok[1:]]
                for i in range(len(ids) - 2):
                    faces.append((ids[0], ids[i + 1], ids[i + 2]))
    return np.asarray(vertices, dtype=np.float32).reshape(-1, 3), np.asarray(faces, dtype=np.int32).reshape(-1, 3) - 1


def load_textures(filename_obj, filename_texture, texture_size):
    """Bilinear-sample a UV image into per-face ts^3 textures (load_obj.py:11-91; "not well tested" upstream,
    unused by 3D-SDN).  Returns a float32 tensor [faces, ts, ts, ts, 3] on the current CUDA device."""
    import PIL.Image
    uv, tfaces = [], []
    with open(filename_obj) as f:
        for line in f:
            tok = line.split()
            if not tok:
                continue
            if tok[0] == 'vt':
                uv.append([float(v) for v in tok[1:3]])
            elif tok[0] == 'f':
                ids = [int(v.split('/')[1]) for v in tok[1:]]
                for i in range(len(ids) - 2):
                    tfaces.append((ids[0], ids[i + 1], ids[i + 2]))
    uv = np.asarray(uv, dtype=np.float32)
    tfaces = np.asarray(tfaces, dtype=np.int32) - 1
    dev = torch.device('cuda', torch.cuda.current_device())
    faces = torch.tensor(uv[tfaces], device=dev) % 1  # [nf, 3, 2]
    image = np.asarray(PIL.Image.open(filename_texture).convert('RGB')).astype(np.float32) / 255.
    image = torch.tensor(image[::-1].copy(), device=dev)
    ih, iw = image.shape[:2]
    ts = texture_size
    g = torch.arange(ts, device=dev, dtype=torch.float32) / (ts - 1.)
    d0, d1, d2 = torch.meshgrid(g, g, g, indexing='ij')
    s = d0 + d1 + d2
    over = s > 1
    d0, d1, d2 = [torch.where(over, d / s, d) for d in (d0, d1, d2)]
    dims = torch.stack([d0, d1, d2], dim=-1).reshape(-1, 3)  # [ts^3, 3]
    pos = torch.einsum('tk,fkc->ftc', dims, faces)  # [nf, ts^3, 2]
    px = pos[..., 0] * (iw - 1)
    py = pos[..., 1] * (ih - 1)
    x0, y0 = px.long(), py.long()
    wx1, wy1 = px - x0.float(), py - y0.float()
    wx0, wy0 = 1 - wx1, 1 - wy1
    x1, y1 = (x0 + 1).clamp(max=iw - 1), ((py + 1).long()).clamp(max=ih - 1)
    c = (image[y0, x0] * (wx0 * wy0)[..., None] + image[y1, x0] * (wx0 * wy1)[..., None] +
         image[y0, x1] * (wx1 * wy0)[..., None] + image[y1, x1] * (wx1 * wy1)[..., None])
    return c.reshape(faces.shape[0], ts, ts, ts, 3).contiguous()


def load_obj(filename_obj, normalization=True, filename_texture=None, texture_size=4):
    """
    Load Wavefront .obj file.
    This function only supports vertices (v x x x) and faces (f x x x).
    Returns numpy arrays (vertices float32 [nv,3], faces int32 [nf,3]) like the reference, plus the texture
    tensor when a texture image is given.
    """
    vertices, faces = _parse_obj(filename_obj)
    textures = None
    if filename_texture is not None:
        textures = load_textures(filename_obj, filename_texture, texture_size)
    if normalization:
        # unit cube centred at zero (load_obj.py:132-136)
        vertices -= vertices.min(0)[None, :]
        vertices /= np.abs(vertices).max()
        vertices *= 2
        vertices -= vertices.max(0)[None, :] / 2
    if textures is None:
        return vertices, faces
    return vertices, faces, textures


def save_obj(filename, vertices, faces):
    if isinstance(vertices, torch.Tensor):
        vertices = vertices.detach().cpu().numpy()
    if isinstance(faces, torch.Tensor):
        faces = faces.detach().cpu().numpy()
    assert vertices.ndim == 2
    assert faces.ndim == 2
    with open(filename, 'w') as f:
        f.write('# %s\n' % os.path.basename(filename))
        f.write('#\n')
        f.write('\n')
        f.write('g mesh\n')
        f.write('\n')
        for vertex in vertices:
            f.write('v  %.4f %.4f %.4f\n' % (vertex[0], vertex[1], vertex[2]))
        f.write('\n')
        for face in faces:
            f.write('f  %d %d %d\n' % (face[0] + 1, face[1] + 1, face[2] + 1))


class Mesh(torch.nn.Module):
    """Learnable mesh: vertices + per-face textures (mesh.py:7-38)."""

    def __init__(self, filename_obj, texture_size=4, normalization=True):
        super(Mesh, self).__init__()
        vertices, faces = load_obj(filename_obj, normalization)
        self.vertices = torch.nn.Parameter(torch.tensor(vertices))
        self.register_buffer('faces', torch.tensor(faces))
        self.num_vertices = self.vertices.shape[0]
        self.num_faces = self.faces.shape[0]
        shape = (self.num_faces, texture_size, texture_size, texture_size, 3)
        self.textures = torch.nn.Parameter(torch.randn(shape) * 0.05)  # chainer.initializers.Normal(): std 0.05
        self.texture_size = texture_size

    def get_batch(self, batch_size):
        vertices = self.vertices[None].expand(batch_size, -1, -1)
        faces = self.faces[None].expand(batch_size, -1, -1)
        textures = torch.sigmoid(self.textures[None].expand(batch_size, -1, -1, -1, -1, -1))
        return vertices, faces, textures

    def set_lr(self, lr_vertices, lr_textures):
        self.vertices.lr = lr_vertices
        self.textures.lr = lr_textures
