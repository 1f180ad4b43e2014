// sdn_render_maps_fwd / _bwd: silhouette + normal + depth maps of a frame's objects from ONE host call each way.
//
// Reference: Derenderer3d.render calls Renderer.forward three times per object
// (/root/reference/geometric/derender3d/models/__init__.py:203-224 -> renderer.py:216-272): every call flips x (:243), builds
// a `_Renderer`, runs look + perspective, vertices_to_faces with fill_back, (for the normal map) face normals as a constant
// texture (:66-93) and one Rasterize; Chainer's autograd walks the same graph back.  The product already renders the three
// maps from one rasterisation, but as five torch.autograd Functions + two element-wise ops -- ~20 launches issued from Python,
// which had become the frame step's bound (host issue 1.34 of 1.38 ms per 16-object frame).  These two entry points issue the
// same launchers of this library in the same order from C; nothing about the arithmetic changes:
//   fwd:  [gather(verts, flip) -> face normals (x sign folded into the colours)]  project(verts, flip) -> gather -> rasterize
//   bwd:  rasterize_bwd (silhouette term with eps_alpha, colour + depth terms with eps, as two Rasterize calls would) ->
//         gather_bwd -> project_bwd [-> normals_bwd -> gather_bwd, added]
//         (r06, only the silhouette differentiated: the edge pass adds to the vertices itself -> project_bwd)
// The x sign of the normal map (renderer.py:268-270) is applied to the face colours instead of to the finished map:
// negation commutes exactly with the rasterizer's products, sums and 2x2 pooling.
#include <cstdlib>

#include "camera_math.h"
#include "sdn_common.h"

using namespace sdn;

namespace {

size_t a256(size_t n) { return (n + 255) & ~(size_t)255; }

struct MapsLayout {
    size_t pv, faces9, face_inv, faces_n, colors, fim, wmap, dmap, rgbmap, bgcopy, raster_ws, raster_ws_bytes, total;
    // backward workspace
    size_t b_raster, b_raster_bytes, g_faces9, g_colors, g_pv, g_faces_n, g_v2, b_total;
    int nf, S;
};

int maps_layout(int bs, int nv, int nf0, int fill_back, int image_size, int flags, MapsLayout& L)
{
    if (bs < 1 || nv < 1 || nf0 < 1 || image_size < 1) return fail(SDN_EINVAL, "sdn_render_maps: bad sizes");
    L.nf = fill_back ? 2 * nf0 : nf0;
    L.S = (flags & SDN_AA) ? 2 * image_size : image_size;
    const size_t n = (size_t)bs * L.nf, px = (size_t)bs * L.S * L.S, v = (size_t)bs * nv;
    const bool normal = (flags & SDN_RGB) != 0;
    size_t o = 0;
    auto take = [&](size_t bytes) {
        const size_t at = o;
        o += a256(bytes);
        return at;
    };
    L.pv = take(v * 12);
    L.faces9 = take(n * 36);
    L.face_inv = take(n * 36);
    L.colors = take(normal ? n * 12 : 0);
    L.fim = take(px * 4);
    L.wmap = take(px * 12);
    L.dmap = take(px * 4);
    L.rgbmap = take(normal ? px * 12 : 0);
    L.bgcopy = take(0);   // (r04 kept a copy of the forward call's background colour here; since ABI 6 _bwd is handed it again)
    int rc = sdn_raster_workspace_bytes(bs, L.nf, L.S, &L.raster_ws_bytes);
    if (rc) return rc;
    L.raster_ws = 0;     // the rasterizer's forward workspace (tile lists: ~130 MB of a 16-object frame) is a separate,
    L.total = o;         // forward-only allocation: it used to sit in the state and stayed pinned by every live autograd graph
    o = 0;
    rc = sdn_raster_bwd_workspace_bytes(bs, L.nf, L.S, &L.b_raster_bytes);
    if (rc) return rc;
    L.b_raster = take(L.b_raster_bytes);
    L.g_faces9 = take(n * 36);
    L.g_colors = take(normal ? n * 12 : 0);
    L.g_pv = take(v * 12);
    L.g_faces_n = take(normal ? n * 36 : 0);
    L.faces_n = take(normal ? n * 36 : 0);   // (the pre-camera faces, gathered again only when the normal map takes a gradient)
    L.g_v2 = take(normal ? v * 12 : 0);
    L.b_total = o;
    return SDN_OK;
}

__global__ __launch_bounds__(256) void k_add_inplace(float* __restrict__ a, const float* __restrict__ b, long n)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) a[i] = a[i] + b[i];
}

// SDN_VERTEX_SINK=0: the r03-r05 backward chain (face-gradient tensor, memset, k_gather_faces_bwd) for A/B runs
bool sink_enabled()
{
    static const bool on = [] { const char* e = getenv("SDN_VERTEX_SINK"); return !(e && e[0] == '0'); }();
    return on;
}

}  // namespace

SDN_API int sdn_render_maps_bytes(int bs, int nv, int nf0, int fill_back, int image_size, int flags, size_t* state_bytes,
                                  size_t* bwd_bytes, size_t* fwd_scratch_bytes)
{
    MapsLayout L;
    int rc = maps_layout(bs, nv, nf0, fill_back, image_size, flags, L);
    if (rc) return rc;
    if (state_bytes) *state_bytes = L.total;
    if (bwd_bytes) *bwd_bytes = L.b_total;
    if (fwd_scratch_bytes) *fwd_scratch_bytes = L.raster_ws_bytes;
    return SDN_OK;
}

SDN_API int sdn_render_maps_fwd(const float* verts, int bs, int nv, const int32_t* faces_idx, int nf0,
                                long faces_batch_stride, int fill_back, int camera_mode, const float* eye, const float* dir,
                                const float* up, const float* width, int flip_x, int image_size, int flags, double near,
                                double far, double eps, const float* bg, float* alpha_out, float* normal_out,
                                float* depth_out, void* state, size_t state_bytes, void* scratch, size_t scratch_bytes,
                                sdnStream stream)
{
    if (!verts || !faces_idx || !alpha_out || !state || !scratch) return fail(SDN_EINVAL, "sdn_render_maps_fwd: null pointer");
    MapsLayout L;
    int rc = maps_layout(bs, nv, nf0, fill_back, image_size, flags, L);
    if (rc) return rc;
    if (state_bytes < L.total) return fail(SDN_ENOMEM, "sdn_render_maps_fwd: state %zu < %zu bytes", state_bytes, L.total);
    if (scratch_bytes < L.raster_ws_bytes)
        return fail(SDN_ENOMEM, "sdn_render_maps_fwd: scratch %zu < %zu bytes", scratch_bytes, L.raster_ws_bytes);
    const bool normal = (flags & SDN_RGB) != 0;
    if (normal && (!normal_out || !bg)) return fail(SDN_EINVAL, "sdn_render_maps_fwd: the normal map needs normal_out and bg");
    if ((flags & SDN_DEPTH) && !depth_out) return fail(SDN_EINVAL, "sdn_render_maps_fwd: depth_out is NULL");
    hipStream_t st = (hipStream_t)stream;
    char* s = (char*)state;
    float* colors = normal ? (float*)(s + L.colors) : nullptr;
    // r06: ONE launch builds the faces: k_face_setup's GATHER build gathers, x-flips, projects, writes the face tensor and the
    // pre-camera normals (the colours of the normal map) and sets the face up.  SDN_MAPS_FUSED_SETUP=0 keeps the four r03 launches
    // (k_face_normals_gather, k_project, k_gather_faces, k_face_setup) for A/B runs; bit-identical either way.
    static const bool fused_setup = [] { const char* e = getenv("SDN_MAPS_FUSED_SETUP"); return !(e && e[0] == '0'); }();
    FaceSource G = FaceSource();
    if (fused_setup) {
        G.verts = verts; G.faces_idx = faces_idx; G.fstride = faces_batch_stride; G.nv = nv; G.nf0 = nf0;
        G.fill_back = fill_back; G.flip_x = flip_x; G.mode = camera_mode; G.eye = eye; G.dir = dir; G.up = up; G.width = width;
        G.faces_out = (float*)(s + L.faces9); G.normals_out = colors; G.sx = flip_x ? -1.0f : 1.0f;
        if (camera_mode < 0 || camera_mode > 2 || (camera_mode != 0 && (!eye || !dir || !up)))
            return fail(SDN_EINVAL, "sdn_render_maps_fwd: camera mode %d needs eye / direction / up", camera_mode);
    } else {
        if (normal) {
            // normals of the fill_back'ed faces BEFORE the camera transform (renderer.py:66-76), on the x-flipped vertices
            if ((rc = launch_face_normals_gather(verts, faces_idx, bs, nv, nf0, faces_batch_stride, fill_back, flip_x,
                                                 flip_x ? -1.0f : 1.0f, colors, st)))
                return rc;
        }
        if ((rc = sdn_project_vertices(verts, bs, nv, camera_mode, eye, dir, up, width, flip_x, (float*)(s + L.pv), stream)))
            return rc;
        if ((rc = launch_gather_faces((const float*)(s + L.pv), faces_idx, bs, nv, nf0, faces_batch_stride, fill_back, 0,
                                      (float*)(s + L.faces9), st)))
            return rc;
    }
    // SDN_LAZY_MAPS: the forward stores the face-index and depth maps only; the weight and colour maps (24 of 32 bytes per
    // internal pixel, 226 MB of a 16-object frame) are re-derived by sdn_render_maps_bwd when the normal or the depth map takes a
    // gradient -- the silhouette gradient, which is what training and the optimisation loop differentiate, reads neither
    const int rflags = (flags & (SDN_RGB | SDN_DEPTH | SDN_AA | SDN_SAVE_MAPS | SDN_STREAM_FACES | SDN_COUNT_WORK | SDN_K1_COVERAGE)) | SDN_ALPHA |
                       (normal ? SDN_FACE_COLOR : 0) | SDN_LAZY_MAPS;
    // (r05: the background colour is no longer copied into the state -- a 12-byte device-to-device copy was a 4.6 us launch
    // of every frame step; sdn_render_maps_bwd is handed the forward call's `bg` again)
    return rasterize_fwd_core(fused_setup ? &G : nullptr, (const float*)(s + L.faces9), colors, normal ? 2 : 0, bs, L.nf, L.S, near,
                              far, eps, bg, 0, rflags, (float*)(s + L.face_inv), (int32_t*)(s + L.fim), (float*)(s + L.wmap),
                              (float*)(s + L.dmap), normal ? (float*)(s + L.rgbmap) : nullptr, normal_out, alpha_out, depth_out,
                              scratch, L.raster_ws_bytes, stream);
}

SDN_API int sdn_render_maps_bwd(const float* verts, int bs, int nv, const int32_t* faces_idx, int nf0,
                                long faces_batch_stride, int fill_back, int camera_mode, const float* eye, const float* dir,
                                const float* up, const float* width, int flip_x, int image_size, int flags, double eps,
                                double eps_alpha, const float* bg, const float* g_alpha, const float* g_normal, const float* g_depth,
                                float* grad_verts, const void* state, size_t state_bytes, void* workspace,
                                size_t workspace_bytes, sdnStream stream)
{
    if (!verts || !faces_idx || !grad_verts || !state || !workspace) return fail(SDN_EINVAL, "sdn_render_maps_bwd: null pointer");
    if (!(flags & SDN_SAVE_MAPS)) return fail(SDN_EINVAL, "sdn_render_maps_bwd: the forward call did not save its maps");
    MapsLayout L;
    int rc = maps_layout(bs, nv, nf0, fill_back, image_size, flags, L);
    if (rc) return rc;
    if (state_bytes < L.total || workspace_bytes < L.b_total)
        return fail(SDN_ENOMEM, "sdn_render_maps_bwd: state %zu / workspace %zu bytes, need %zu / %zu", state_bytes,
                    workspace_bytes, L.total, L.b_total);
    const bool normal = (flags & SDN_RGB) != 0;
    if (!normal) g_normal = nullptr;
    if (!(flags & SDN_DEPTH)) g_depth = nullptr;
    if (normal && (g_normal || g_depth) && !bg)
        return fail(SDN_EINVAL, "sdn_render_maps_bwd: the colour map is re-derived for this gradient: pass the forward call's bg");
    hipStream_t st = (hipStream_t)stream;
    const char* s = (const char*)state;
    char* w = (char*)workspace;
    const float* faces9 = (const float*)(s + L.faces9);
    const float* colors = normal ? (const float*)(s + L.colors) : nullptr;
    float* g_faces9 = (float*)(w + L.g_faces9);
    float* g_colors = g_normal ? (float*)(w + L.g_colors) : nullptr;
    // SDN_SPARSE_GRAD: faces without a pixel keep unwritten gradient rows; the gathers below skip them by the same flags
    const int base = (flags & (SDN_AA | SDN_SERIAL_EDGES)) | (normal ? SDN_FACE_COLOR : 0) | SDN_SPARSE_GRAD;
    // the visible-face flags exist only when a pass below runs an edge term (silhouette or colour gradient): a depth-only
    // backward never builds them, and the gathers must then visit every (dense, zero-filled) row
    const uint32_t* visible = (g_alpha || g_normal) ? raster_bwd_visible_flags(w + L.b_raster) : nullptr;
    auto raster_bwd = [&](int fl, double e, const float* gr, const float* ga, const float* gd, float* gt) {
        return sdn_rasterize_bwd(faces9, colors, normal ? 2 : 0, bs, L.nf, L.S, e, fl, (const float*)(s + L.face_inv),
                                 (const int32_t*)(s + L.fim), (const float*)(s + L.wmap), (const float*)(s + L.dmap),
                                 normal ? (const float*)(s + L.rgbmap) : nullptr, gr, ga, gd, g_faces9, gt, w + L.b_raster,
                                 L.b_raster_bytes, stream);
    };
    if ((normal && (g_normal || g_depth)) || (!normal && g_depth)) {
        // the forward call was lazy: weights (and the colour map) of every pixel, by the forward's own shading routine
        char* sw = const_cast<char*>(s);
        if ((rc = launch_reshade_maps(faces9, colors, normal ? 2 : 0, bs, L.nf, L.S, 0.0, eps,
                                      normal ? bg : nullptr, 0,
                                      (flags & (SDN_AA | SDN_K1_COVERAGE)) | (normal ? (SDN_RGB | SDN_FACE_COLOR) : 0), (const float*)(s + L.face_inv),
                                      (const int32_t*)(s + L.fim), (const float*)(s + L.dmap), (float*)(sw + L.wmap),
                                      normal ? (float*)(sw + L.rgbmap) : nullptr, st)))
            return rc;
    }
    // the silhouette term with rasterize_silhouettes' eps (module default), then colour + depth with the Renderer's: what
    // the separate Rasterize calls of the reference produce (derender3d/models/renderer.py:37,57,90-92)
    float* g_pv = (float*)(w + L.g_pv);
    if (g_alpha && !g_depth && !g_colors && sink_enabled()) {
        // only the silhouette is differentiated (the test-time optimisation and the training loss, scripts/main.py:142-151, 447): the
        // edge pass adds each face's gradient straight to its vertices -- no face-gradient tensor, no gather launch, no memset
        const VertexSink sink = {faces_idx, faces_batch_stride, nv, nf0, fill_back, g_pv};
        if ((rc = rasterize_bwd_core(&sink, faces9, colors, normal ? 2 : 0, bs, L.nf, L.S, eps_alpha, base | SDN_ALPHA,
                                     (const float*)(s + L.face_inv), (const int32_t*)(s + L.fim), (const float*)(s + L.wmap),
                                     (const float*)(s + L.dmap), nullptr, nullptr, g_alpha, nullptr, nullptr, nullptr,
                                     w + L.b_raster, L.b_raster_bytes, stream)))
            return rc;
        return sdn_project_vertices_bwd(verts, bs, nv, camera_mode, eye, dir, up, width, flip_x, g_pv, grad_verts, stream);
    }
    if (!normal) {
        // silhouette (+ depth) only: one pass, as a single rasterize_rgbad call without colours (K7 does not use eps)
        if ((rc = raster_bwd(base | SDN_ALPHA | (g_depth ? SDN_DEPTH : 0), eps_alpha, nullptr, g_alpha, g_depth, nullptr))) return rc;
    } else if ((rc = raster_bwd(base | SDN_ALPHA, eps_alpha, nullptr, g_alpha, nullptr, nullptr))) {
        return rc;
    }
    if (normal && (g_normal || g_depth)) {
        if (g_colors) {
            hipError_t e = hipMemsetAsync(g_colors, 0, (size_t)bs * L.nf * 12, st);
            if (e != hipSuccess) return fail(SDN_ELAUNCH, "sdn_render_maps_bwd: memset: %s", hipGetErrorString(e));
        }
        if ((rc = raster_bwd(base | (g_normal ? SDN_RGB : 0) | (g_depth ? SDN_DEPTH : 0) | SDN_ACCUMULATE, eps, g_normal, nullptr,
                             g_depth, g_colors)))
            return rc;
    }
    if ((rc = launch_gather_faces_bwd(g_faces9, faces_idx, bs, nv, nf0, faces_batch_stride, fill_back, 0, 1, g_pv, st, visible)))
        return rc;
    if ((rc = sdn_project_vertices_bwd(verts, bs, nv, camera_mode, eye, dir, up, width, flip_x, g_pv, grad_verts, stream)))
        return rc;
    if (g_colors) {
        // the normal map's colours depend on the (flipped) vertices through the face normals
        float* g_faces_n = (float*)(w + L.g_faces_n);
        float* g_v2 = (float*)(w + L.g_v2);
        float* faces_n = (float*)(w + L.faces_n);
        if ((rc = launch_gather_faces(verts, faces_idx, bs, nv, nf0, faces_batch_stride, fill_back, flip_x, faces_n, st))) return rc;
        if ((rc = launch_face_normals_bwd(faces_n, g_colors, (long)bs * L.nf, flip_x ? -1.0f : 1.0f,
                                          g_faces_n, st)))
            return rc;
        if ((rc = launch_gather_faces_bwd(g_faces_n, faces_idx, bs, nv, nf0, faces_batch_stride, fill_back, flip_x, 1, g_v2, st, visible)))
            return rc;
        const long n = (long)bs * nv * 3;
        hipLaunchKernelGGL(k_add_inplace, dim3(cdiv(n, 256)), dim3(256), 0, st, grad_verts, (const float*)g_v2, n);
        if ((rc = check_launch("k_add_inplace"))) return rc;
    }
    return SDN_OK;
}
