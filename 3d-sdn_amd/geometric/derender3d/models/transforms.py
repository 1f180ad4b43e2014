"""FFD (Bernstein free-form deformation) and PerspectiveTransform.

Reference: /root/reference/geometric/derender3d/models/transforms.py:10-158.  Same constructor arguments,
same forward signatures and return values.  Differences are operational only:
  * the Bernstein basis `B` and the control lattice `P0` are registered buffers that live on the module's
    device (the reference re-uploads both with `.cuda()` on every call, transforms.py:97 -- 3.8 MB per object);
  * the decode V = (P0 + dP) . B is one [V,64] x [64,3] contraction instead of a [V,3,4,4,4] temporary.
"""
import numpy as np
import scipy.special
import torch

from torch.nn.modules import Module


class FFD(Module):
    class Constraint:
        class Type:
            symmetry = 0
            homogeneity = 1

        class Axis:
            x = 0
            y = 1
            z = 2

        @staticmethod
        def symmetry(axis):
            c = FFD.Constraint(FFD.Constraint.Type.symmetry)
            c.axis = axis
            return c

        @staticmethod
        def homogeneity(axis, index):
            c = FFD.Constraint(FFD.Constraint.Type.homogeneity)
            c.axis = axis
            c.index = index
            return c

        def __init__(self, type):
            self.type = type

    @staticmethod
    def flip(x, dim):
        return torch.flip(x, dims=(dim,))

    def __init__(self, vertices, num_grids=4, constraints=None):
        super(FFD, self).__init__()

        assert num_grids % 2 == 0

        self.num_grids = num_grids
        self.constraints = constraints if constraints is not None else []

        vertices = torch.as_tensor(vertices, dtype=torch.float32)
        grids = np.arange(num_grids)
        binoms = torch.tensor(scipy.special.binom(num_grids - 1, grids), dtype=torch.float32)
        grid_1ds = torch.tensor(grids, dtype=torch.float32)
        grid_3ds = torch.tensor(np.stack(np.meshgrid(grids, grids, grids, indexing='ij')), dtype=torch.float32)

        # Bernstein polynomials of degree n-1 in (0.5 + v) per axis (transforms.py:58-62)
        coeff = (
            binoms *
            torch.pow(torch.unsqueeze(0.5 + vertices, dim=2), grid_1ds) *
            torch.pow(torch.unsqueeze(0.5 - vertices, dim=2), num_grids - 1 - grid_1ds)
        )
        B = torch.einsum('ni,nj,nk->nijk', torch.unbind(coeff, dim=1))
        self.register_buffer('B', torch.unsqueeze(B, dim=1), persistent=False)        # [V,1,n,n,n]
        self.register_buffer('P0', grid_3ds / (num_grids - 1) - 0.5, persistent=False)  # [3,n,n,n]

    def constrain(self, ffd_coeff):
        """Apply the symmetry / homogeneity constraints to the raw offsets (transforms.py:69-95)."""
        dP = ffd_coeff.view(3, self.num_grids, self.num_grids, self.num_grids)
        for constraint in self.constraints:
            if constraint.type == FFD.Constraint.Type.symmetry:
                _dP = FFD.flip(dP, dim=constraint.axis + 1)
                (_dPx, _dPy, _dPz) = torch.unbind(_dP, dim=0)
                _dP = torch.stack([_dPx, _dPy, -_dPz], dim=0)
                dP = (dP + _dP) / 2
            elif constraint.type == FFD.Constraint.Type.homogeneity:
                dPs = torch.unbind(dP, dim=constraint.axis + 1)
                _dPs = [dPs[index] for index in constraint.index]
                _dP_mean = sum(_dPs) / len(_dPs)
                _dPs = []
                for index in range(self.num_grids):
                    if index in constraint.index:
                        _dP = torch.cat([
                            _dP_mean[:constraint.axis], dPs[index][constraint.axis:constraint.axis + 1],
                            _dP_mean[constraint.axis + 1:]], dim=0)
                    else:
                        _dP = dPs[index]
                    _dPs.append(_dP)
                dP = torch.stack(_dPs, dim=constraint.axis + 1)
        return dP

    def forward(self, ffd_coeff):
        dP = self.constrain(ffd_coeff)
        n3 = self.num_grids ** 3
        P = (self.P0.to(dP.device) + dP).reshape(3, n3)
        return torch.matmul(self.B.to(dP.device).reshape(-1, n3), P.t())  # [V,3]


def constrain_batched(dP, constraints, num_grids):
    """FFD.constrain for a batch: dP [n, 3, g, g, g] (transforms.py:69-95 with one leading dimension)."""
    for constraint in constraints:
        if constraint.type == FFD.Constraint.Type.symmetry:
            _dP = torch.flip(dP, dims=(constraint.axis + 2,))
            _dP = torch.stack([_dP[:, 0], _dP[:, 1], -_dP[:, 2]], dim=1)
            dP = (dP + _dP) / 2
        elif constraint.type == FFD.Constraint.Type.homogeneity:
            dPs = torch.unbind(dP, dim=constraint.axis + 2)
            _dP_mean = sum(dPs[index] for index in constraint.index) / len(constraint.index)
            _dPs = []
            for index in range(num_grids):
                if index in constraint.index:
                    a = constraint.axis
                    _dPs.append(torch.cat([_dP_mean[:, :a], dPs[index][:, a:a + 1], _dP_mean[:, a + 1:]], dim=1))
                else:
                    _dPs.append(dPs[index])
            dP = torch.stack(_dPs, dim=constraint.axis + 2)
    return dP


class FFDBank(Module):
    """All mesh templates of a Derenderer3d on the device, padded to a common size, so that a whole frame's objects
    are decoded by ONE kernel launch (csrc/fast_ffd.hip) instead of one FFD.forward per object.

    Padding: extra vertices repeat vertex 0 of their template (they never change the zoom-to-fit minimum of
    PerspectiveTransform, transforms.py:149) and extra faces are (0, 0, 0), which the rasterizer drops.
    """

    def __init__(self, ffds, faces_list):
        super(FFDBank, self).__init__()
        from sdn_hip import ops  # noqa: F401  (fail early when the HIP library is missing)
        self.num_grids = ffds[0].num_grids
        self.constraints = ffds[0].constraints
        n3 = self.num_grids ** 3
        self.nverts = [int(f.B.shape[0]) for f in ffds]
        self.nfaces = [int(f.shape[0]) for f in faces_list]
        vmax, fmax = max(self.nverts), max(self.nfaces)
        Bt = torch.zeros(len(ffds), n3, vmax)
        faces = torch.zeros(len(ffds), fmax, 3, dtype=torch.int32)
        for c, (ffd, f) in enumerate(zip(ffds, faces_list)):
            B = ffd.B.reshape(-1, n3)
            Bt[c, :, :B.shape[0]] = B.t()
            Bt[c, :, B.shape[0]:] = B[0][:, None]
            faces[c, :f.shape[0]] = f
        self.register_buffer('Bt', Bt.contiguous(), persistent=False)
        self.register_buffer('faces', faces.contiguous(), persistent=False)
        self.register_buffer('P0', ffds[0].P0.reshape(3, n3).clone(), persistent=False)
        # FFD.constrain (symmetry / homogeneity averaging, transforms.py:69-95) is linear in the coefficients: apply it
        # once to the identity and keep the [3 g^3, 3 g^3] matrix, so that decode() is one small GEMM instead of ~40
        # flips / stacks / cats per call (and as many again in backward)
        eye = torch.eye(3 * n3).reshape(3 * n3, 3, self.num_grids, self.num_grids, self.num_grids)
        C = constrain_batched(eye, self.constraints, self.num_grids).reshape(3 * n3, 3 * n3)
        self.register_buffer('constraint_matrix', C.contiguous(), persistent=False)

    def decode(self, ffd_coeffs, classes):
        """ffd_coeffs [n, 3 * g^3], classes [n] (int tensor on the device) -> vertices [n, vmax, 3], faces [n, fmax, 3]."""
        from sdn_hip import ops
        n = ffd_coeffs.shape[0]
        g = self.num_grids
        dP = ffd_coeffs.reshape(n, 3 * g ** 3) @ self.constraint_matrix  # row i of the matrix = constrain(e_i)
        P = self.P0[None] + dP.reshape(n, 3, g ** 3)
        cls = classes.to(torch.int32)
        verts = ops.FFDDecode.apply(P.contiguous(), self.Bt, cls)
        return verts, self.faces.index_select(0, classes.long())


class PerspectiveTransform(Module):
    def forward(self,
                vertices,
                scales=None,
                rotations=None,
                translations=None,
                perspective_translations=None,
                zooms=None,
                zoom_tos=None):
        """transforms.py:102-158.  The complete test-time form (scale + rotation + translation + zoom-to-fit, what
        Derenderer3d.render and the optimisation loop of scripts/main.py:439-456 use) runs as one fused HIP op on the GPU;
        every other argument combination takes the element-wise path below, which is the reference's own arithmetic."""
        if (vertices.is_cuda and scales is not None and rotations is not None and translations is not None
                and zoom_tos is not None and zooms is None and vertices.dim() == 3):
            from sdn_hip import ops
            n = vertices.shape[0]
            persp = translations if perspective_translations is None else perspective_translations

            def full(x, k):
                return x.reshape(-1, k).expand(n, k) if x.shape[0] != n else x.reshape(n, k)
            return ops.PerspectiveTransformFn.apply(vertices, full(scales, 3), full(rotations, 4),
                                                    full(translations, 3), full(persp, 3), full(zoom_tos, 1))
        return self._forward_elementwise(vertices, scales, rotations, translations, perspective_translations, zooms,
                                         zoom_tos)

    def _forward_elementwise(self, vertices, scales=None, rotations=None, translations=None,
                             perspective_translations=None, zooms=None, zoom_tos=None):

        if scales is not None:
            scales = scales.unsqueeze(dim=1)
            vertices = vertices * scales

        if rotations is not None:
            (a, b, c, d) = torch.unbind(rotations, dim=1)

            # rotation matrix of the unit quaternion (a, b, c, d) (transforms.py:118-128)
            T = torch.stack([
                a * a + b * b - c * c - d * d,
                2 * b * c - 2 * a * d,
                2 * b * d + 2 * a * c,
                2 * b * c + 2 * a * d,
                a * a - b * b + c * c - d * d,
                2 * c * d - 2 * a * b,
                2 * b * d - 2 * a * c,
                2 * c * d + 2 * a * b,
                a * a - b * b - c * c + d * d,
            ], dim=1).view(-1, 3, 3)

            vertices = torch.matmul(vertices, torch.transpose(T, dim0=1, dim1=2))

        if translations is not None:
            translations = translations.unsqueeze(dim=1)
            vertices = vertices + translations

        if perspective_translations is not None:
            perspective_translations = perspective_translations.unsqueeze(dim=1)
        else:
            perspective_translations = translations

        (x, y, z) = torch.unbind(vertices, dim=2)
        (x0, y0, z0) = torch.unbind(perspective_translations, dim=2)

        # shear so that the ray (x0, y0, z0) becomes the optical axis (transforms.py:143-146)
        x = x - x0 / z0 * z
        y = y - y0 / z0 * z

        if zoom_tos is not None:
            zooms = torch.min(torch.abs(z) / torch.max(torch.abs(x), torch.abs(y)), dim=1, keepdim=True)[0] * zoom_tos

        z = z / zooms

        vertices = torch.stack([x, y, z], dim=2)

        if zoom_tos is None:
            return vertices
        else:
            return (vertices, zooms)
