// Vertex-side kernels of the renderer for gfx950: camera transform + perspective, face gather with
// fill_back, face normals -- and their backward passes.
//
// Reference (paths under /root/reference/geometric/):
//   neural_renderer/look.py:7-45, look_at.py:7-46   camera basis (chainer F.normalize: x / (|x| + 1e-5))
//   neural_renderer/perspective.py:5-19             x / z / tan(angle), y / z / tan(angle)
//   neural_renderer/vertices_to_faces.py:4-21       gather; renderer.py:41 fill_back = faces ++ faces[::-1]
//   derender3d/models/renderer.py:66-76             face normals = normalize(cross(v0 - v1, v2 - v1))
//   derender3d/models/renderer.py:243               x flip (folded in as flip_x)
// The reference runs each of these as a handful of CuPy array kernels plus Chainer autograd; here each
// direction is one coalesced pass.  Arithmetic order matches oracle/nr_oracle.py (left-to-right sums,
// no FMA: this file is built with -ffp-contract=off).
#include "camera_math.h"
#include "raster_math.h"
#include "sdn_common.h"

namespace sdn {

char* error_slot()
{
    static thread_local char buf[512] = {0};
    return buf;
}

__global__ __launch_bounds__(256) void k_project(const float* __restrict__ verts, int bs, int nv, int mode,
                                                  const float* eye, const float* dir, const float* up,
                                                  const float* width, int flip_x, float* __restrict__ out)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)bs * nv) return;
    const int b = (int)(i / nv);
    float v[3] = {verts[i * 3 + 0], verts[i * 3 + 1], verts[i * 3 + 2]};
    if (flip_x) v[0] = v[0] * -1.0f;
    float o[3] = {v[0], v[1], v[2]};
    if (mode != 0) {
        const Basis B = camera_basis(mode, eye, dir, up, b);
        const float d[3] = {v[0] - B.e[0], v[1] - B.e[1], v[2] - B.e[2]};
        o[0] = (d[0] * B.xa[0] + d[1] * B.xa[1]) + d[2] * B.xa[2];
        o[1] = (d[0] * B.ya[0] + d[1] * B.ya[1]) + d[2] * B.ya[2];
        o[2] = (d[0] * B.za[0] + d[1] * B.za[1]) + d[2] * B.za[2];
    }
    if (width) {
        const float w = width[b];
        o[0] = o[0] / o[2] / w;
        o[1] = o[1] / o[2] / w;
    }
    out[i * 3 + 0] = o[0];
    out[i * 3 + 1] = o[1];
    out[i * 3 + 2] = o[2];
}

__global__ __launch_bounds__(256) void k_project_bwd(const float* __restrict__ verts, int bs, int nv, int mode,
                                                      const float* eye, const float* dir, const float* up,
                                                      const float* width, int flip_x,
                                                      const float* __restrict__ grad_out,
                                                      float* __restrict__ grad_verts)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)bs * nv) return;
    const int b = (int)(i / nv);
    float v[3] = {verts[i * 3 + 0], verts[i * 3 + 1], verts[i * 3 + 2]};
    if (flip_x) v[0] = v[0] * -1.0f;
    float o[3] = {v[0], v[1], v[2]};
    Basis B;
    if (mode != 0) {
        B = camera_basis(mode, eye, dir, up, b);
        const float d[3] = {v[0] - B.e[0], v[1] - B.e[1], v[2] - B.e[2]};
        o[0] = (d[0] * B.xa[0] + d[1] * B.xa[1]) + d[2] * B.xa[2];
        o[1] = (d[0] * B.ya[0] + d[1] * B.ya[1]) + d[2] * B.ya[2];
        o[2] = (d[0] * B.za[0] + d[1] * B.za[1]) + d[2] * B.za[2];
    }
    float g[3] = {grad_out[i * 3 + 0], grad_out[i * 3 + 1], grad_out[i * 3 + 2]};
    if (width) {
        // x' = (ox / oz) / w : d/dox = 1 / oz / w ; d/doz = -(ox / oz) / oz / w
        const float w = width[b];
        const float gx = g[0] / w, gy = g[1] / w;
        const float gz = g[2] + (-(gx * (o[0] / o[2])) / o[2]) + (-(gy * (o[1] / o[2])) / o[2]);
        g[0] = gx / o[2];
        g[1] = gy / o[2];
        g[2] = gz;
    }
    if (mode != 0) {
        const float t0 = (g[0] * B.xa[0] + g[1] * B.ya[0]) + g[2] * B.za[0];
        const float t1 = (g[0] * B.xa[1] + g[1] * B.ya[1]) + g[2] * B.za[1];
        const float t2 = (g[0] * B.xa[2] + g[1] * B.ya[2]) + g[2] * B.za[2];
        g[0] = t0;
        g[1] = t1;
        g[2] = t2;
    }
    if (flip_x) g[0] = g[0] * -1.0f;
    grad_verts[i * 3 + 0] = g[0];
    grad_verts[i * 3 + 1] = g[1];
    grad_verts[i * 3 + 2] = g[2];
}

__global__ __launch_bounds__(256) void k_gather_faces(const float* __restrict__ verts,
                                                       const int32_t* __restrict__ faces_idx, int bs, int nv, int nf0,
                                                       long fstride, int fill_back, int flip_x, float* __restrict__ out)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)bs * nf0) return;
    const int b = (int)(i / nf0), f = (int)(i % nf0);
    const int32_t* idx = faces_idx + (size_t)b * fstride + (size_t)f * 3;
    const int nf = fill_back ? 2 * nf0 : nf0;
    float v[3][3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float* p = verts + ((size_t)b * nv + idx[k]) * 3;
        v[k][0] = flip_x ? p[0] * -1.0f : p[0];   // derender3d/models/renderer.py:243 (sdn_render_maps_fwd)
        v[k][1] = p[1];
        v[k][2] = p[2];
    }
    float* o = out + ((size_t)b * nf + f) * 9;
#pragma unroll
    for (int k = 0; k < 3; k++)
#pragma unroll
        for (int d = 0; d < 3; d++) o[3 * k + d] = v[k][d];
    if (fill_back) {
        float* o2 = out + ((size_t)b * nf + nf0 + f) * 9;
#pragma unroll
        for (int k = 0; k < 3; k++)
#pragma unroll
            for (int d = 0; d < 3; d++) o2[3 * k + d] = v[2 - k][d];
    }
}

__global__ __launch_bounds__(256) void k_gather_faces_bwd(const float* __restrict__ grad_faces,
                                                           const int32_t* __restrict__ faces_idx, int bs, int nv,
                                                           int nf0, long fstride, int fill_back, int flip_x,
                                                           float* __restrict__ grad_verts,
                                                           const uint32_t* __restrict__ visible)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)bs * nf0) return;
    const int b = (int)(i / nf0), f = (int)(i % nf0);
    const int nf = fill_back ? 2 * nf0 : nf0;
    // (sparse rows: a face without a pixel has no gradient and its row may be unwritten -- never read it)
    bool use1 = true, use2 = fill_back != 0;
    if (visible) {
        use1 = visible[(size_t)b * nf + f] != 0u;
        use2 = use2 && visible[(size_t)b * nf + nf0 + f] != 0u;
        if (!use1 && !use2) return;
    }
    const int32_t* idx = faces_idx + (size_t)b * fstride + (size_t)f * 3;
    const float* g = grad_faces + ((size_t)b * nf + f) * 9;
    const float* g2 = grad_faces + ((size_t)b * nf + nf0 + f) * 9;
    float t[9];
    bool any = false;
#pragma unroll
    for (int k = 0; k < 3; k++)
#pragma unroll
        for (int d = 0; d < 3; d++) {
            float x = use1 ? g[3 * k + d] : 0.0f;
            if (use2) x = x + g2[3 * (2 - k) + d];
            t[3 * k + d] = x;
            any = any || x != 0.0f;
        }
    // most faces of a mesh are hidden or back-facing and carry an all-zero gradient: no atomics for those
    if (!any) return;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        float* dst = grad_verts + ((size_t)b * nv + idx[k]) * 3;
        if (flip_x) t[3 * k] = t[3 * k] * -1.0f;
#pragma unroll
        for (int d = 0; d < 3; d++) unsafeAtomicAdd(&dst[d], t[3 * k + d]);
    }
}

__global__ __launch_bounds__(256) void k_face_normals(const float* __restrict__ faces, long total, float sx,
                                                       float* __restrict__ normals)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const float* f = faces + i * 9;
    const float v10[3] = {f[0] - f[3], f[1] - f[4], f[2] - f[5]};
    const float v12[3] = {f[6] - f[3], f[7] - f[4], f[8] - f[5]};
    float c[3];
    cross3(v10, v12, c);
    normalize3(c);
    normals[i * 3 + 0] = c[0] * sx;   // sx = -1: the x sign fix of derender3d/models/renderer.py:268-270, applied to the colours
    normals[i * 3 + 1] = c[1];
    normals[i * 3 + 2] = c[2];
}

// k_gather_faces + k_face_normals in one pass (sdn_render_maps_fwd): the normals of face f and of its fill_back twin straight
// from the vertices, without the [bs, nf, 3, 3] face tensor in between (49 MB written and read back per frame).  The same
// float operations on the same values as the two kernels, so the colours are bit-identical.
__global__ __launch_bounds__(256) void k_face_normals_gather(const float* __restrict__ verts,
                                                              const int32_t* __restrict__ faces_idx, int bs, int nv, int nf0,
                                                              long fstride, int fill_back, int flip_x, float sx,
                                                              float* __restrict__ normals)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)bs * nf0) return;
    const int b = (int)(i / nf0), f = (int)(i % nf0);
    const int32_t* idx = faces_idx + (size_t)b * fstride + (size_t)f * 3;
    const int nf = fill_back ? 2 * nf0 : nf0;
    float v[3][3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float* p = verts + ((size_t)b * nv + idx[k]) * 3;
        v[k][0] = flip_x ? p[0] * -1.0f : p[0];
        v[k][1] = p[1];
        v[k][2] = p[2];
    }
#pragma unroll
    for (int twin = 0; twin < 2; twin++) {
        if (twin && !fill_back) break;
        const float* a = twin ? v[2] : v[0];   // the twin holds the vertices in reverse order (vertices_to_faces + fill_back)
        const float* c2 = twin ? v[0] : v[2];
        const float v10[3] = {a[0] - v[1][0], a[1] - v[1][1], a[2] - v[1][2]};
        const float v12[3] = {c2[0] - v[1][0], c2[1] - v[1][1], c2[2] - v[1][2]};
        float c[3];
        cross3(v10, v12, c);
        normalize3(c);
        float* o = normals + ((size_t)b * nf + (twin ? nf0 : 0) + f) * 3;
        o[0] = c[0] * sx;
        o[1] = c[1];
        o[2] = c[2];
    }
}

__global__ __launch_bounds__(256) void k_face_normals_bwd(const float* __restrict__ faces,
                                                           const float* __restrict__ grad_normals, long total, float sx,
                                                           float* __restrict__ grad_faces)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const float* f = faces + i * 9;
    const float v10[3] = {f[0] - f[3], f[1] - f[4], f[2] - f[5]};
    const float v12[3] = {f[6] - f[3], f[7] - f[4], f[8] - f[5]};
    float c[3];
    cross3(v10, v12, c);
    const float nrm = sqrtf((c[0] * c[0] + c[1] * c[1]) + c[2] * c[2]);
    const float s = nrm + 1e-5f;
    const float gn[3] = {grad_normals[i * 3 + 0] * sx, grad_normals[i * 3 + 1], grad_normals[i * 3 + 2]};
    // y = c / s, s = |c| + eps  =>  g_c = g_n / s - c * (g_n . c) / (s^2 |c|)
    const float dot = (gn[0] * c[0] + gn[1] * c[1]) + gn[2] * c[2];
    const float coef = (nrm > 0.0f) ? dot / (s * s * nrm) : 0.0f;
    const float gc[3] = {gn[0] / s - c[0] * coef, gn[1] / s - c[1] * coef, gn[2] / s - c[2] * coef};
    float g10[3], g12[3];
    cross3(v12, gc, g10);  // cross.py:47-53: ga = cross(b, gc), gb = cross(gc, a)
    cross3(gc, v10, g12);
    float* o = grad_faces + i * 9;
#pragma unroll
    for (int d = 0; d < 3; d++) {
        o[d] = g10[d];
        o[3 + d] = -(g10[d] + g12[d]);
        o[6 + d] = g12[d];
    }
}

}  // namespace sdn

namespace sdn {

int launch_gather_faces(const float* verts, const int32_t* faces_idx, int bs, int nv, int nf0, long fstride, int fill_back,
                        int flip_x, float* faces_out, hipStream_t st)
{
    hipLaunchKernelGGL(k_gather_faces, dim3(cdiv((long)bs * nf0, 256)), dim3(256), 0, st, verts, faces_idx, bs, nv, nf0,
                       fstride, fill_back, flip_x, faces_out);
    return check_launch("k_gather_faces");
}

int launch_gather_faces_bwd(const float* grad_faces, const int32_t* faces_idx, int bs, int nv, int nf0, long fstride,
                            int fill_back, int flip_x, int zero_first, float* grad_verts, hipStream_t st,
                            const uint32_t* visible)
{
    if (zero_first) {
        hipError_t e = hipMemsetAsync(grad_verts, 0, (size_t)bs * nv * 3 * sizeof(float), st);
        if (e != hipSuccess) return fail(SDN_ELAUNCH, "hipMemsetAsync(grad_verts): %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(k_gather_faces_bwd, dim3(cdiv((long)bs * nf0, 256)), dim3(256), 0, st, grad_faces, faces_idx, bs, nv,
                       nf0, fstride, fill_back, flip_x, grad_verts, visible);
    return check_launch("k_gather_faces_bwd");
}

int launch_face_normals(const float* faces, long total, float sx, float* normals, hipStream_t st)
{
    hipLaunchKernelGGL(k_face_normals, dim3(cdiv(total, 256)), dim3(256), 0, st, faces, total, sx, normals);
    return check_launch("k_face_normals");
}

int launch_face_normals_gather(const float* verts, const int32_t* faces_idx, int bs, int nv, int nf0, long fstride,
                               int fill_back, int flip_x, float sx, float* normals, hipStream_t st)
{
    hipLaunchKernelGGL(k_face_normals_gather, dim3(cdiv((long)bs * nf0, 256)), dim3(256), 0, st, verts, faces_idx, bs, nv, nf0,
                       fstride, fill_back, flip_x, sx, normals);
    return check_launch("k_face_normals_gather");
}

int launch_face_normals_bwd(const float* faces, const float* grad_normals, long total, float sx, float* grad_faces,
                            hipStream_t st)
{
    hipLaunchKernelGGL(k_face_normals_bwd, dim3(cdiv(total, 256)), dim3(256), 0, st, faces, grad_normals, total, sx,
                       grad_faces);
    return check_launch("k_face_normals_bwd");
}

}  // namespace sdn

using namespace sdn;

SDN_API const char* sdn_last_error(void) { return error_slot(); }
SDN_API int sdn_version(void) { return SDN_ABI_VERSION; }

static int check_camera(const char* who, int mode, const float* eye, const float* dir, const float* up)
{
    if (mode < 0 || mode > 2) return fail(SDN_EINVAL, "%s: camera_mode %d not in {0,1,2}", who, mode);
    if (mode != 0 && (!eye || !dir || !up)) return fail(SDN_EINVAL, "%s: camera needs eye, dir, up", who);
    return SDN_OK;
}

SDN_API int sdn_project_vertices(const float* verts, int bs, int nv, int camera_mode, const float* eye,
                                 const float* dir, const float* up, const float* width, int flip_x, float* out,
                                 sdnStream stream)
{
    if (!verts || !out || bs <= 0 || nv <= 0) return fail(SDN_EINVAL, "sdn_project_vertices: bad arguments");
    int rc = check_camera("sdn_project_vertices", camera_mode, eye, dir, up);
    if (rc) return rc;
    hipLaunchKernelGGL(k_project, dim3(cdiv((long)bs * nv, 256)), dim3(256), 0, (hipStream_t)stream, verts, bs, nv,
                       camera_mode, eye, dir, up, width, flip_x, out);
    return check_launch("k_project");
}

SDN_API int sdn_project_vertices_bwd(const float* verts, int bs, int nv, int camera_mode, const float* eye,
                                     const float* dir, const float* up, const float* width, int flip_x,
                                     const float* grad_out, float* grad_verts, sdnStream stream)
{
    if (!verts || !grad_out || !grad_verts || bs <= 0 || nv <= 0)
        return fail(SDN_EINVAL, "sdn_project_vertices_bwd: bad arguments");
    int rc = check_camera("sdn_project_vertices_bwd", camera_mode, eye, dir, up);
    if (rc) return rc;
    hipLaunchKernelGGL(k_project_bwd, dim3(cdiv((long)bs * nv, 256)), dim3(256), 0, (hipStream_t)stream, verts, bs,
                       nv, camera_mode, eye, dir, up, width, flip_x, grad_out, grad_verts);
    return check_launch("k_project_bwd");
}

SDN_API int sdn_gather_faces(const float* verts, const int32_t* faces_idx, int bs, int nv, int nf0,
                             long faces_batch_stride, int fill_back, float* faces_out, sdnStream stream)
{
    if (!verts || !faces_idx || !faces_out || bs <= 0 || nv <= 0 || nf0 <= 0)
        return fail(SDN_EINVAL, "sdn_gather_faces: bad arguments");
    return launch_gather_faces(verts, faces_idx, bs, nv, nf0, faces_batch_stride, fill_back, 0, faces_out, (hipStream_t)stream);
}

SDN_API int sdn_gather_faces_bwd(const float* grad_faces, const int32_t* faces_idx, int bs, int nv, int nf0,
                                 long faces_batch_stride, int fill_back, float* grad_verts, sdnStream stream)
{
    if (!grad_faces || !faces_idx || !grad_verts || bs <= 0 || nv <= 0 || nf0 <= 0)
        return fail(SDN_EINVAL, "sdn_gather_faces_bwd: bad arguments");
    return launch_gather_faces_bwd(grad_faces, faces_idx, bs, nv, nf0, faces_batch_stride, fill_back, 0, 1, grad_verts,
                                   (hipStream_t)stream);
}

SDN_API int sdn_face_normals(const float* faces, long n_faces_total, float* normals, sdnStream stream)
{
    if (!faces || !normals || n_faces_total <= 0) return fail(SDN_EINVAL, "sdn_face_normals: bad arguments");
    return launch_face_normals(faces, n_faces_total, 1.0f, normals, (hipStream_t)stream);
}

SDN_API int sdn_face_normals_bwd(const float* faces, const float* grad_normals, long n_faces_total, float* grad_faces,
                                 sdnStream stream)
{
    if (!faces || !grad_normals || !grad_faces || n_faces_total <= 0)
        return fail(SDN_EINVAL, "sdn_face_normals_bwd: bad arguments");
    return launch_face_normals_bwd(faces, grad_normals, n_faces_total, 1.0f, grad_faces, (hipStream_t)stream);
}
