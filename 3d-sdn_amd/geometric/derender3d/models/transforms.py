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


class _Constraint(object):
    """A linear constraint on the lattice offsets dP [3, g, g, g] (reference: transforms.py:11-36, 69-95).
      symmetry(axis):           dP <- (dP + M flip_axis(dP)) / 2 with M = diag(1, 1, -1) (the z component is negated
                                whatever the axis -- kept as the reference has it);
      homogeneity(axis, index): on the lattice planes `index` along `axis`, every component except `axis` is replaced
                                by its mean over those planes."""

    class Type:
        symmetry = 0
        homogeneity = 1

    class Axis:
        x = 0
        y = 1
        z = 2

    def __init__(self, type, axis=None, index=None):
        self.type = type
        self.axis = axis
        self.index = index

    @classmethod
    def symmetry(cls, axis):
        return cls(cls.Type.symmetry, axis=axis)

    @classmethod
    def homogeneity(cls, axis, index):
        return cls(cls.Type.homogeneity, axis=axis, index=list(index))


def apply_constraints(dP, constraints):
    """dP [..., 3, g, g, g] (any number of leading batch dimensions) -> constrained offsets, same shape."""
    comp = dP.dim() - 4                       # dimension that holds the (x, y, z) components
    for c in constraints:
        ax = comp + 1 + c.axis                # lattice dimension the constraint acts along
        if c.type == _Constraint.Type.symmetry:
            shape = [1] * dP.dim()
            shape[comp] = 3
            sign = torch.tensor([1.0, 1.0, -1.0], dtype=dP.dtype, device=dP.device).reshape(shape)
            dP = (dP + torch.flip(dP, dims=(ax,)) * sign) / 2
        else:
            idx = torch.as_tensor(c.index, device=dP.device)
            planes = dP.index_select(ax, idx)
            total = planes.select(ax, 0)
            for k in range(1, len(c.index)):  # summed in index order, like the reference's python sum()
                total = total + planes.select(ax, k)
            mean = (total / len(c.index)).unsqueeze(ax).expand_as(planes)
            keep = torch.zeros(3, dtype=torch.bool, device=dP.device)
            keep[c.axis] = True               # the component along the axis keeps its own value
            shape = [1] * dP.dim()
            shape[comp] = 3
            dP = dP.index_copy(ax, idx, torch.where(keep.reshape(shape), planes, mean))
    return dP


def bernstein_basis(t, one_minus_t, n):
    """B[v, i] = C(n-1, i) t_v^i (1 - t_v)^(n-1-i): the degree n-1 Bernstein polynomials at t [V] (1 - t is passed in so
    that the caller decides how it is rounded)."""
    i = torch.arange(n, dtype=torch.float32)
    binom = torch.tensor(scipy.special.binom(n - 1, np.arange(n)), dtype=torch.float32)
    return binom * torch.pow(t[:, None], i) * torch.pow(one_minus_t[:, None], n - 1 - i)


class FFD(Module):
    """Free-form deformation of a template: V = sum_ijk (P0 + dP)_ijk B_i(x) B_j(y) B_k(z) over a g^3 control lattice
    on [-0.5, 0.5]^3 (reference: transforms.py:10-99; same constructor, `forward(ffd_coeff [3 g^3]) -> [V, 3]`)."""

    Constraint = _Constraint

    @staticmethod
    def flip(x, dim):
        return torch.flip(x, dims=(dim,))

    def __init__(self, vertices, num_grids=4, constraints=None):
        super(FFD, self).__init__()
        if num_grids % 2:
            raise AssertionError('num_grids must be even')
        self.num_grids = num_grids
        self.constraints = list(constraints) if constraints is not None else []
        v = torch.as_tensor(vertices, dtype=torch.float32)
        # lattice coordinate t = 0.5 + v; 1 - t is formed as 0.5 - v like the reference (:61): bit-compatible basis
        per_axis = [bernstein_basis(0.5 + v[:, d], 0.5 - v[:, d], num_grids) for d in range(3)]
        B = torch.einsum('ni,nj,nk->nijk', per_axis[0], per_axis[1], per_axis[2])
        lattice = torch.linspace(0, num_grids - 1, num_grids)
        P0 = torch.stack(torch.meshgrid(lattice, lattice, lattice, indexing='ij')) / (num_grids - 1) - 0.5
        self.register_buffer('B', B[:, None], persistent=False)   # [V, 1, g, g, g]
        self.register_buffer('P0', P0, persistent=False)          # [3, g, g, g]

    def constrain(self, ffd_coeff):
        g = self.num_grids
        return apply_constraints(ffd_coeff.reshape(3, g, g, g), self.constraints)

    def forward(self, ffd_coeff):
        dP = self.constrain(ffd_coeff)
        n3 = self.num_grids ** 3
        P = (self.P0.to(dP.device) + dP).reshape(3, n3)
        return torch.matmul(self.B.to(dP.device).reshape(-1, n3), P.t())  # [V, 3]


def constrain_batched(dP, constraints, num_grids):
    """apply_constraints for dP [n, 3, g, g, g] (kept as a name for FFDBank and the tests)."""
    return apply_constraints(dP, constraints)


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
        cls, faces = self._class_rows(classes)
        # P = P0 + coeffs . C (row i of the constraint matrix = constrain(e_i)) is formed inside the op
        verts = ops.FFDDecode.apply(ffd_coeffs.reshape(n, 3 * g ** 3), self.Bt, cls, self.constraint_matrix, self.P0)
        return verts, faces

    def _class_rows(self, classes):
        """(int32 classes, faces [n, fmax, 3] of those classes).  The optimisation loop decodes the same objects every
        iteration (scripts/main.py:439-456): the 8 MB face gather and the two index conversions are kept for as long as
        the caller passes the very same, unmodified tensor.  "Unmodified" is judged by identity and the tensor's autograd
        version counter: writes through `classes.data` or by kernels outside autograd do not advance it -- after such a write
        call `invalidate_class_cache()` (the same class of staleness as sdn_hip.conv.invalidate_weight_caches documents for
        packed weights).  The cache pins one [n, fmax, 3] int32 face tensor (~8 MB for 16 objects) per bank."""
        hit = self.__dict__.get('_rows_cache')
        if hit is not None and hit[0] is classes and hit[1] == classes._version:
            return hit[2], hit[3]
        cls = classes.to(torch.int32)
        faces = self.faces.index_select(0, classes.long())
        self.__dict__['_rows_cache'] = (classes, classes._version, cls, faces)
        return cls, faces


    def invalidate_class_cache(self):
        """forget the cached (classes -> int32 classes, faces) gather: see _class_rows"""
        self.__dict__.pop('_rows_cache', None)


class PerspectiveTransform(Module):
    def forward(self,
                vertices,
                scales=None,
                rotations=None,
                translations=None,
                perspective_translations=None,
                zooms=None,
                zoom_tos=None):
        """transforms.py:102-158.  The two complete forms Derenderer3d.render uses -- test time (scale + rotation +
        translation + zoom-to-fit, `zoom_tos`) and training / the optimisation loop of scripts/main.py:433-456 (the same
        with given `zooms`) -- run as one fused HIP op on the GPU; every other argument combination takes the element-wise
        path below, which is the reference's own arithmetic."""
        if (vertices.is_cuda and scales is not None and rotations is not None and translations is not None
                and (zoom_tos is None) != (zooms is None) and vertices.dim() == 3):
            from sdn_hip import ops
            n = vertices.shape[0]
            persp = translations if perspective_translations is None else perspective_translations

            def full(x, k):
                return x.reshape(-1, k).expand(n, k) if x.shape[0] != n else x.reshape(n, k)
            # (one tensor for both translations -- what Derenderer3d.render passes -- stays ONE object: the fused op then returns
            # a single, already summed gradient instead of two that autograd adds with a launch of its own)
            t3 = full(translations, 3)
            p3 = t3 if persp is translations else full(persp, 3)
            if zooms is None:
                return ops.PerspectiveTransformFn.apply(vertices, full(scales, 3), full(rotations, 4), t3, p3, full(zoom_tos, 1))
            out, _ = ops.PerspectiveTransformFn.apply(vertices, full(scales, 3), full(rotations, 4), t3, p3, None, full(zooms, 1))
            return out
        return self._forward_elementwise(vertices, scales, rotations, translations, perspective_translations, zooms,
                                         zoom_tos)

    def _forward_elementwise(self, vertices, scales=None, rotations=None, translations=None,
                             perspective_translations=None, zooms=None, zoom_tos=None):
        """vertices [n, V, 3]; per-object scales [n, 3], unit quaternions [n, 4], translations [n, 3]."""
        v = vertices
        if scales is not None:
            v = v * scales[:, None, :]
        if rotations is not None:
            v = torch.matmul(v, quaternion_matrix(rotations).transpose(1, 2))
        if translations is not None:
            v = v + translations[:, None, :]
        centre = translations if perspective_translations is None else perspective_translations
        centre = centre[:, None, :]
        # shear that turns the ray through `centre` into the optical axis (transforms.py:143-146)
        depth = v[..., 2]
        sheared = [v[..., d] - centre[..., d] / centre[..., 2] * depth for d in (0, 1)]
        if zoom_tos is not None:
            # zoom-to-fit: the tightest |z| / max(|x|, |y|) over the vertices (:149)
            spread = torch.max(sheared[0].abs(), sheared[1].abs())
            zooms = (depth.abs() / spread).min(dim=1, keepdim=True)[0] * zoom_tos
        out = torch.stack((sheared[0], sheared[1], depth / zooms), dim=2)
        return out if zoom_tos is None else (out, zooms)


def quaternion_matrix(q):
    """Rotation matrices [n, 3, 3] of quaternions q = (a, b, c, d) [n, 4], in the un-normalised form the reference
    uses (transforms.py:118-128: the diagonal is a^2 + b^2 - c^2 - d^2, ...)."""
    a, b, c, d = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    rows = (
        (a * a + b * b - c * c - d * d, 2 * b * c - 2 * a * d, 2 * b * d + 2 * a * c),
        (2 * b * c + 2 * a * d, a * a - b * b + c * c - d * d, 2 * c * d - 2 * a * b),
        (2 * b * d - 2 * a * c, 2 * c * d + 2 * a * b, a * a - b * b - c * c + d * d),
    )
    return torch.stack([torch.stack(r, dim=1) for r in rows], dim=1)
