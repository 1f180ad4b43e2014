from sdn_hip import ops


def vertices_to_faces(vertices, faces):
    """
    :param vertices: [batch size, number of vertices, 3]
    :param faces: [batch size, number of faces, 3)
    :return: [batch size, number of faces, 3, 3]

    Reference: neural_renderer/vertices_to_faces.py:4-21 (fancy-index gather; scatter-add backward).
    """
    assert (vertices.dim() == 3)
    assert (faces.dim() == 3)
    assert (vertices.shape[0] == faces.shape[0])
    assert (vertices.shape[2] == 3)
    assert (faces.shape[2] == 3)
    return ops.GatherFaces.apply(vertices, faces, False)
