"""nr.vertices_to_faces: gather the three corner positions of every face.

Reference: neural_renderer/vertices_to_faces.py:4-21 (a fancy-index gather on the flattened vertex array; Chainer derives
the scatter-add backward).  Here both directions are one HIP launch each (`sdn_gather_faces[_bwd]`, csrc/geometry.hip).
"""
from sdn_hip import ops


def vertices_to_faces(vertices, faces):
    """vertices [batch, nv, 3] float32, faces [batch, nf, 3] int32  ->  [batch, nf, 3 (corner), 3 (xyz)]."""
    if vertices.dim() != 3 or vertices.shape[2] != 3:
        raise ValueError('vertices must be [batch, nv, 3], got %s' % (tuple(vertices.shape),))
    if faces.dim() != 3 or faces.shape[2] != 3:
        raise ValueError('faces must be [batch, nf, 3], got %s' % (tuple(faces.shape),))
    if vertices.shape[0] != faces.shape[0]:
        raise ValueError('batch sizes differ: %d vertices sets, %d face sets' % (vertices.shape[0], faces.shape[0]))
    return ops.GatherFaces.apply(vertices, faces, False)
