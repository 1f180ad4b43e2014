// Shared host-side helpers of libsdn_hip.so (error slot, launch checks).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "../../include/sdn_hip.h"

#define SDN_API extern "C" __attribute__((visibility("default")))

namespace sdn {

char* error_slot();  // thread-local buffer, 512 bytes

inline int fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(error_slot(), 512, fmt, ap);
    va_end(ap);
    return code;
}

inline int check_launch(const char* what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(SDN_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
    return SDN_OK;
}

inline unsigned cdiv(long a, long b) { return (unsigned)((a + b - 1) / b); }

// Opt-in launch timing for bench.py's roofline figures (timing.hip).  A TimedLaunch brackets the kernel launches issued
// during its lifetime with a hipEvent pair on the launch stream and files the pair, with the algorithmic work of the
// launch (bytes or flops, the caller's unit), under a slot; sdn_timing_read_slot sums them.  Costs nothing when off.
// TIME_CONV_NARROW (r05): the exact-fp32 head kernels (conv_narrow.hip) used to file under the MFMA slots.
enum TimingSlot { TIME_RASTER_TILES = 0, TIME_EDGE_SCAN = 1, TIME_CONV_GEMM = 2, TIME_CONV_WGRAD = 3, TIME_CONV_NARROW = 4,
                  TIME_RASTER_TILES_K1 = 5, TIME_SLOTS = 6 };
// The launchers only see PADDED channel counts (D's 18-channel input arrives as 32): a caller that knows the layer's true
// shape declares the algorithmic work of the NEXT timed launch issued by this thread; consumed (reset) by that launch.
void timing_declare_work(double work);
struct TimedLaunch {
    TimedLaunch(int slot, hipStream_t st, double work);
    ~TimedLaunch();
    int slot_;
    hipStream_t st_;
    double work_;
    hipEvent_t e0_ = nullptr, e1_ = nullptr;
};

// internal launchers of geometry.hip with the extras only the fused entry points use (x flip, colour sign, accumulate)
int launch_gather_faces(const float* verts, const int32_t* faces_idx, int bs, int nv, int nf0, long fstride, int fill_back,
                        int flip_x, float* faces_out, hipStream_t st);
// `visible` (optional, u32 [bs, nf]): rows of grad_faces whose flag is 0 are not read and count as zero (SDN_SPARSE_GRAD)
int launch_gather_faces_bwd(const float* grad_faces, const int32_t* faces_idx, int bs, int nv, int nf0, long fstride,
                            int fill_back, int flip_x, int zero_first, float* grad_verts, hipStream_t st,
                            const uint32_t* visible = nullptr);
const uint32_t* raster_bwd_visible_flags(const void* raster_bwd_workspace);
int launch_face_normals_gather(const float* verts, const int32_t* faces_idx, int bs, int nv, int nf0, long fstride,
                               int fill_back, int flip_x, float sx, float* normals, hipStream_t st);
int launch_face_normals(const float* faces, long total, float sx, float* normals, hipStream_t st);
// weight / colour maps of an SDN_LAZY_MAPS forward, re-derived from (face index, depth) by the forward's own shading routine
int launch_reshade_maps(const float* faces, const float* textures, int ts, int bs, int nf, int S, double far, double eps,
                        const float* bg, int bg_per_batch, int flags, const float* face_inv, const int32_t* face_index_map,
                        const float* depth_map, float* weight_map, float* rgb_map, hipStream_t st);
int launch_face_normals_bwd(const float* faces, const float* grad_normals, long total, float sx, float* grad_faces,
                            hipStream_t st);
// sdn_rasterize_fwd with the faces built from vertices inside the face set-up kernel (camera_math.h: FaceSource; r06)
struct FaceSource;
int rasterize_fwd_core(const FaceSource* src, const float* faces, const float* textures, int ts, int bs, int nf, int S, double near,
                       double far, double eps, const float* bg, int bg_per_batch, int flags, float* face_inv,
                       int32_t* face_index_map, float* weight_map, float* depth_map, float* rgb_map, float* rgb_out,
                       float* alpha_out, float* depth_out, void* workspace, size_t workspace_bytes, sdnStream stream);

// sdn_rasterize_bwd whose edge pass adds each face's gradient straight to its vertices (r06, sdn_render_maps_bwd's silhouette-only
// case): k_edge_plan clears grad_verts [bs, nv, 3] on its way, k_edge_reduce scatters with float atomics -- the [bs, nf, 3, 3] face
// gradient, its gather launch and the clearing memset are gone.  Only for passes without colour / depth terms.
struct VertexSink {
    const int32_t* faces_idx;   // [1 | bs, nf0, 3]
    long fstride;               // 0: one index list for the batch
    int nv, nf0, fill_back;     // fill_back: face nf0 + f is face f with its vertices reversed
    float* grad_verts;          // [bs, nv, 3]
};
int rasterize_bwd_core(const VertexSink* sink, const float* faces, const float* textures, int ts, int bs, int nf, int S, double eps,
                       int flags, const float* face_inv, const int32_t* face_index_map, const float* weight_map,
                       const float* depth_map, const float* rgb_map, const float* g_rgb_out, const float* g_alpha_out,
                       const float* g_depth_out, float* grad_faces, float* grad_textures, void* workspace, size_t workspace_bytes,
                       sdnStream stream);

}  // namespace sdn
