// Camera transform + perspective of ONE vertex and the face normal of one triangle: the float operations of k_project /
// k_face_normals (geometry.hip), shared with the face set-up of the rasterizer (raster_fwd.hip, r06: the fused three-map forward
// gathers, projects and sets a face up in one kernel).  Both files are built with -ffp-contract=off: the same operations in
// the same order wherever they are evaluated, so fusing cannot change a bit.
// Reference: /root/reference/geometric/neural_renderer/look.py:7-45, look_at.py:7-46 (chainer F.normalize: x / (|x| + 1e-5)),
// perspective.py:5-19, derender3d/models/renderer.py:66-76 (face normals), :243 (x flip).
#pragma once
#include <hip/hip_runtime.h>

namespace sdn {

struct Basis {
    float xa[3], ya[3], za[3], e[3];
};

__device__ __forceinline__ void normalize3(float v[3])
{
    const float n = sqrtf((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]) + 1e-5f;
    v[0] = v[0] / n;
    v[1] = v[1] / n;
    v[2] = v[2] / n;
}

__device__ __forceinline__ void cross3(const float a[3], const float b[3], float c[3])
{
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}

__device__ __forceinline__ Basis camera_basis(int mode, const float* eye, const float* dir, const float* up, int b)
{
    Basis B;
    float u[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        B.e[k] = eye[3 * b + k];
        u[k] = up[3 * b + k];
        const float d = dir[3 * b + k];
        B.za[k] = (mode == 2) ? (d - B.e[k]) : d;  // look_at: at - eye (look_at.py:30)
    }
    normalize3(B.za);
    cross3(u, B.za, B.xa);
    normalize3(B.xa);
    cross3(B.za, B.xa, B.ya);
    normalize3(B.ya);
    return B;
}

// v: the (already x-flipped) vertex; B valid when mode != 0; has_w: divide by depth and the screen half-width w
__device__ __forceinline__ void project_vertex(const float v[3], int mode, const Basis& B, bool has_w, float w, float o[3])
{
    o[0] = v[0];
    o[1] = v[1];
    o[2] = v[2];
    if (mode != 0) {
        const float d[3] = {v[0] - B.e[0], v[1] - B.e[1], v[2] - B.e[2]};
        o[0] = (d[0] * B.xa[0] + d[1] * B.xa[1]) + d[2] * B.xa[2];
        o[1] = (d[0] * B.ya[0] + d[1] * B.ya[1]) + d[2] * B.ya[2];
        o[2] = (d[0] * B.za[0] + d[1] * B.za[1]) + d[2] * B.za[2];
    }
    if (has_w) {
        o[0] = o[0] / o[2] / w;
        o[1] = o[1] / o[2] / w;
    }
}

// normalize(cross(v0 - v1, v2 - v1)) with the x sign sx folded in (renderer.py:66-76, :268-270)
__device__ __forceinline__ void face_normal(const float v0[3], const float v1[3], const float v2[3], float sx, float c[3])
{
    const float v10[3] = {v0[0] - v1[0], v0[1] - v1[1], v0[2] - v1[2]};
    const float v12[3] = {v2[0] - v1[0], v2[1] - v1[1], v2[2] - v1[2]};
    cross3(v10, v12, c);
    normalize3(c);
    c[0] = c[0] * sx;
}

// r06: where the GATHER build of k_face_setup takes its faces from (sdn_render_maps_fwd): vertices + indices + camera instead
// of a finished [bs, nf, 3, 3] tensor.  It writes that tensor (faces_out) and the face normals on its way.
struct FaceSource {
    const float* verts;        // [bs, nv, 3]
    const int32_t* faces_idx;  // [bs or 1, nf0, 3]
    long fstride;              // elements between the index lists of two batch items (0: shared)
    int nv, nf0, fill_back, flip_x, mode;
    const float *eye, *dir, *up, *width;
    float* faces_out;          // [bs, nf, 9] post-projection faces, nf = nf0 or 2 nf0
    float* normals_out;        // optional [bs, nf, 3]: normals of the pre-camera (x-flipped) faces, x sign sx
    float sx;
};

}  // namespace sdn
