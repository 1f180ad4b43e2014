/*
 * sdn_hip.h -- C ABI of libsdn_hip.so, the MI355X (gfx950) implementation of 3D-SDN's hot path.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch's allocator on the Python side),
 *     except `size_t* out`/`int* out` result slots and `const char*` returns, which are host memory;
 *   - every launcher takes a `sdnStream` (a hipStream_t passed as void*), enqueues asynchronously
 *     and returns 0 on success or a negative SDN_E* code; sdn_last_error() holds the text
 *     (thread-local);  there is no global mutable state, so the library is re-entrant across the
 *     one-Python-thread-per-GPU callers of nn.DataParallel
 *     (reference: geometric/scripts/main.py:182);
 *   - tensors are dense, row-major, float32 / int32, shapes as documented per argument.
 *
 * Each entry point cites the reference interface it replaces (paths under
 * /root/reference/geometric/ or /root/reference/textural/).  INTEGRATION.md shows the
 * reference-side binding (ctypes) a maintainer would add.
 */
#ifndef SDN_HIP_H
#define SDN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* sdnStream;

#define SDN_OK 0
#define SDN_EINVAL (-1)   /* bad argument (shape, flag combination, null pointer) */
#define SDN_ELAUNCH (-2)  /* HIP launch / runtime error */
#define SDN_ENOMEM (-3)   /* workspace too small */

/* flags of sdn_rasterize_fwd / _bwd */
#define SDN_RGB 1          /* return_rgb   (rasterize.py:21) */
#define SDN_ALPHA 2        /* return_alpha */
#define SDN_DEPTH 4        /* return_depth */
#define SDN_AA 8           /* outputs are 2x2 average-pooled (rasterize.py:942-966) */
#define SDN_FACE_COLOR 16  /* `textures` is [bs,nf,3]: one colour per face, sampled through the same
                              trilinear arithmetic as a constant ts=2 texture (render_normal,
                              derender3d/models/renderer.py:78-79) */
#define SDN_SAVE_MAPS 32   /* keep the S x S maps needed by sdn_rasterize_bwd */
#define SDN_ACCUMULATE 64  /* sdn_rasterize_bwd: add into grad_faces / grad_textures instead of overwriting */
#define SDN_STREAM_FACES 256 /* sdn_rasterize_fwd: skip the per-tile face lists; every tile streams all faces (the path
                               taken automatically when the lists overflow their budget; for verification) */
#define SDN_COUNT_WORK 512 /* sdn_rasterize_fwd: also tally candidate pixel tests / tests passed / depth keys of k_raster_tiles
                             into the workspace (read with sdn_raster_work_counters; a measurement build of the kernel,
                             never timed) */
#define SDN_SPARSE_GRAD 1024 /* sdn_rasterize_bwd: grad_faces rows of faces that own no pixel of the face-index map (their
                                gradient is zero) are left UNWRITTEN; the caller must skip them -- their flags are
                                u32[bs * nf] at byte 256 of the workspace (used by sdn_render_maps_bwd: most faces of a mesh are
                                hidden, writing and re-reading 36 zero bytes for each was a tenth of the frame step) */
#define SDN_LAZY_MAPS 2048 /* sdn_rasterize_fwd with SDN_SAVE_MAPS: store only the face-index and depth maps (8 of the 32 bytes
                              per internal pixel); the barycentric-weight and colour maps are re-derived -- bit-identically, by
                              the forward's own shading routine -- when a backward pass needs them (depth / colour
                              gradients), never for the silhouette gradient.  Used by sdn_render_maps_fwd / _bwd. */
#define SDN_K1_COVERAGE 4096 /* sdn_rasterize_fwd / sdn_render_maps_fwd: the coverage rule and barycentric arithmetic of the
                              reference's DEFAULT forward kernel K1 (rasterize.py:102-236, selected by scripts/env.sh:11 through
                              NEURAL_RENDERER_UNSAFE=1): pixel-space scanlines over the x-sorted vertices instead of the safe
                              kernels' NDC edge tests.  Exact depth ties, which K1 leaves to thread scheduling, go to the lowest
                              face index.  A plain (untuned) tile kernel serves it; the backward entry points need no flag (they
                              read the maps and face_inv the forward call left, as the reference's do). */
#define SDN_SERIAL_EDGES 128 /* sdn_rasterize_bwd: walk every edge serially in the reference's summation order
                               (bit-comparable with rasterize.py:523-745; slow, for verification) */

const char* sdn_last_error(void);
/* The ABI revision this header describes.  It is raised whenever an entry point gains / loses an argument OR a caller-owned
 * buffer changes its required size behind an unchanged signature (r04: the `key` / `acc` scratch of
 * sdn_perspective_transform*, new arguments of sdn_in_apply / sdn_in_bwd / sdn_act_bwd / sdn_render_maps_*; r05: struct sdn_op
 * with 40 ints, sdn_render_maps_bwd takes bg).  A binding
 * must compare sdn_version() with the SDN_ABI_VERSION it was written against and refuse a library that answers otherwise
 * (sdn_hip/__init__.py: lib()): a stale lib/libsdn_hip.so would otherwise be handed buffers of the wrong size. */
#define SDN_ABI_VERSION 8
int sdn_version(void);

/* ---- camera: neural_renderer/look.py:7-45, look_at.py:7-46, perspective.py:5-19 ------------------
 * out[b,v] = perspective(look*(verts[b,v] * (flip_x ? (-1,1,1) : 1))).
 * camera_mode: 0 none, 1 'look' (dir = viewing direction), 2 'look_at' (dir = `at` point).
 * eye/dir/up: [bs,3].  width: [bs] = tan(angle/180*3.1416) or NULL for no perspective division.
 * flip_x folds derender3d/models/renderer.py:243. */
int sdn_project_vertices(const float* verts, int bs, int nv, int camera_mode, const float* eye,
                         const float* dir, const float* up, const float* width, int flip_x,
                         float* out, sdnStream stream);
/* grad_verts[b,v] = d loss / d verts, given grad_out = d loss / d out.  (Chainer autograd of the ops
 * above; the reference only propagates to vertices, derender3d/models/renderer.py:205-213.) */
int sdn_project_vertices_bwd(const float* verts, int bs, int nv, int camera_mode, const float* eye,
                             const float* dir, const float* up, const float* width, int flip_x,
                             const float* grad_out, float* grad_verts, sdnStream stream);

/* ---- vertices_to_faces + fill_back: neural_renderer/vertices_to_faces.py:4-21, renderer.py:41 -----
 * faces_out[b, f]      = verts[b, faces[b,f,{0,1,2}]]            f <  nf0
 * faces_out[b, nf0+f]  = verts[b, faces[b,f,{2,1,0}]]            when fill_back
 * faces_idx may be shared by the whole batch (faces_batch_stride = 0) or per batch (= nf0*3). */
int sdn_gather_faces(const float* verts, const int32_t* faces_idx, int bs, int nv, int nf0,
                     long faces_batch_stride, int fill_back, float* faces_out, sdnStream stream);
/* scatter-add of grad_faces [bs,nf,3,3] into grad_verts [bs,nv,3] (zeroed by the callee). */
int sdn_gather_faces_bwd(const float* grad_faces, const int32_t* faces_idx, int bs, int nv, int nf0,
                         long faces_batch_stride, int fill_back, float* grad_verts, sdnStream stream);

/* ---- face normals: derender3d/models/renderer.py:66-76 (cross.py:25-38 + chainer normalize) -------
 * normals[b,f] = normalize(cross(v0 - v1, v2 - v1)), eps 1e-5 added to the norm. faces [bs,nf,3,3]. */
int sdn_face_normals(const float* faces, long n_faces_total, float* normals, sdnStream stream);
int sdn_face_normals_bwd(const float* faces, const float* grad_normals, long n_faces_total,
                         float* grad_faces, sdnStream stream);

/* ---- Rasterize: neural_renderer/rasterize.py:19-894 + rasterize_rgbad :897-974 --------------------
 * S = internal image size (2*image_size with SDN_AA).  Workspace: query first. */
int sdn_raster_workspace_bytes(int bs, int nf, int S, size_t* out);

/* Forward.  faces [bs,nf,3,3] post-projection (x,y NDC, z depth).
 * textures: NULL | [bs,nf,ts,ts,ts,3] | [bs,nf,3] with SDN_FACE_COLOR.   bg: [3] or [bs,3].
 * Saved state (required iff SDN_SAVE_MAPS; else may be NULL):
 *   face_index_map [bs,S,S] i32, weight_map [bs,S,S,3], depth_map [bs,S,S], rgb_map [bs,S,S,3]
 *   (only with SDN_RGB).   face_inv [bs,nf,3,3] is always written (workspace-like, caller-owned).
 * Outputs (R = S/2 with SDN_AA else S; vertically flipped like rasterize.py:953-957):
 *   rgb_out [bs,3,R,R], alpha_out [bs,R,R], depth_out [bs,R,R]; each may be NULL if its flag is off. */
int sdn_rasterize_fwd(const float* faces, const float* textures, int ts, int bs, int nf, int S,
                      double near, double far, double eps, const float* bg, int bg_per_batch, int flags,
                      float* face_inv, int32_t* face_index_map, float* weight_map, float* depth_map,
                      float* rgb_map, float* rgb_out, float* alpha_out, float* depth_out,
                      void* workspace, size_t workspace_bytes, sdnStream stream);

/* After a forward call with SDN_COUNT_WORK: out3 (HOST memory) = {candidate pixel tests, tests passed
 * (rasterize.py:311-313), depth keys submitted (:332)} of k_raster_tiles, summed over the launch.  Synchronises the stream.
 * Measurement aid for bench.py's ALU roofline; the counting build of the kernel is never the timed one. */
int sdn_raster_work_counters(const void* workspace, int bs, int nf, int S, unsigned long long* out3, sdnStream stream);

/* Same counting build: out8 (HOST memory) = shader-clock ticks summed over the launch's WAVES for {batch fetch + waiting
 * for the tile's other waves, lane-private boxes, wave-shared boxes, thin faces, epilogue, whole kernel}, then the
 * longest wave's ticks and the number of waves (a 1/16 sample of the tiles; ticks of the 100 MHz constant clock).  Tells an unbalanced launch (max >> mean) from a uniformly slow one.
 * Measurement aid (tools/tile_stats.py); synchronises the stream. */
int sdn_raster_phase_clocks(const void* workspace, int bs, int nf, int S, unsigned long long* out8, sdnStream stream);

/* Backward (rasterize.py:846-886): K5 silhouette/colour edge gradient, K6 texture scatter, K7 depth.
 * g_* are gradients wrt the (pooled, flipped) outputs of the forward call, NULL = zero.
 * grad_faces [bs,nf,3,3] and grad_textures (same shape as textures) are fully written by the callee
 * (added to with SDN_ACCUMULATE).  Workspace: query sdn_raster_bwd_workspace_bytes first. */
int sdn_raster_bwd_workspace_bytes(int bs, int nf, int S, size_t* out);
int sdn_rasterize_bwd(const float* faces, const float* textures, int ts, int bs, int nf, int S,
                      double eps, int flags, const float* face_inv, const int32_t* face_index_map,
                      const float* weight_map, const float* depth_map, const float* rgb_map,
                      const float* g_rgb_out, const float* g_alpha_out, const float* g_depth_out,
                      float* grad_faces, float* grad_textures, void* workspace, size_t workspace_bytes,
                      sdnStream stream);

/* ---- the three maps of a frame's objects in one call each way ------------------------------------------------------------
 * Derenderer3d.render calls Renderer.forward three times per object (derender3d/models/__init__.py:203-224 ->
 * derender3d/models/renderer.py:216-272): x flip (:243), look + perspective, vertices_to_faces with fill_back, face normals
 * as a constant texture (:66-93), Rasterize, x sign of the normal map (:268-270); Chainer's autograd walks it back.
 * sdn_render_maps_fwd issues this library's launchers for all objects of a frame from C, in that order:
 *   [sdn_gather_faces(verts, x flipped) -> sdn_face_normals]  sdn_project_vertices -> sdn_gather_faces -> sdn_rasterize_fwd
 * and sdn_render_maps_bwd the matching sdn_rasterize_bwd (silhouette term with eps_alpha, colour + depth terms with eps:
 * what the reference's separate Rasterize calls use, renderer.py:37,57,90-92) -> sdn_gather_faces_bwd ->
 * sdn_project_vertices_bwd [-> sdn_face_normals_bwd -> sdn_gather_faces_bwd, added].
 * flags: SDN_RGB = the normal map is wanted, SDN_DEPTH, SDN_AA, SDN_SAVE_MAPS (required for _bwd), SDN_SERIAL_EDGES (_bwd).
 * verts [bs,nv,3] as handed to Renderer.forward (NOT flipped); faces_idx / camera arguments as sdn_gather_faces /
 * sdn_project_vertices; bg [3] device (normal map only).  Outputs alpha [bs,R,R], normal [bs,3,R,R], depth [bs,R,R]
 * (R = image_size; NULL when not requested).  state: caller-owned, sdn_render_maps_bytes; it carries the projected vertices,
 * both face arrays, the colours and the S x S maps to the backward call; scratch (fwd_scratch_bytes): the rasterizer's tile
 * lists, dead when the forward call returns to the stream (r04: no longer part of the state a live graph pins).
 * g_* NULL = no gradient for that map.  _bwd's bg (ABI 6): the forward call's background colour again (needed when g_normal or
 * g_depth is given with the normal map on: the lazily stored colour map is re-derived from it; the forward call no longer copies
 * it into the state). */
int sdn_render_maps_bytes(int bs, int nv, int nf0, int fill_back, int image_size, int flags, size_t* state_bytes,
                          size_t* bwd_workspace_bytes, size_t* fwd_scratch_bytes);
int sdn_render_maps_fwd(const float* verts, int bs, int nv, const int32_t* faces_idx, int nf0, long faces_batch_stride,
                        int fill_back, int camera_mode, const float* eye, const float* dir, const float* up,
                        const float* width, int flip_x, int image_size, int flags, double near, double far, double eps,
                        const float* bg, float* alpha_out, float* normal_out, float* depth_out, void* state,
                        size_t state_bytes, void* scratch, size_t scratch_bytes, sdnStream stream);
int sdn_render_maps_bwd(const float* verts, int bs, int nv, const int32_t* faces_idx, int nf0, long faces_batch_stride,
                        int fill_back, int camera_mode, const float* eye, const float* dir, const float* up,
                        const float* width, int flip_x, int image_size, int flags, double eps, double eps_alpha,
                        const float* bg, const float* g_alpha, const float* g_normal, const float* g_depth, float* grad_verts,
                        const void* state, size_t state_bytes, void* workspace, size_t workspace_bytes, sdnStream stream);

/* ---- FFD decode: derender3d/models/transforms.py:68-99 (FFD.forward), batched over objects of different templates --
 * Bt  [n_classes, ncoef, vmax]  Bernstein basis of every template, coefficient-major (padded vertices repeat vertex 0)
 * P   [n, 3, ncoef]             control points P0 + dP of each object (constraints already applied)
 * cls [n] int32                 template of each object
 * out [n, vmax, 3]              deformed vertices.     grad_P [n, 3, ncoef] = d loss / d P given grad_out [n, vmax, 3]. */
int sdn_ffd_decode(const float* Bt, const float* P, const int32_t* cls, int n, int vmax, int ncoef,
                   float* out, sdnStream stream);
int sdn_ffd_decode_bwd(const float* Bt, const int32_t* cls, const float* grad_out, int n, int vmax,
                       int ncoef, float* grad_P, sdnStream stream);
/* The symmetry / homogeneity constraints of FFD.constrain (derender3d/models/transforms.py:69-95) are linear in the
 * coefficients: M [m, m] (row i = constrain(e_i), built once on the host side), m = 3 * grids^3.
 * out [n, m] = base [m] (optional) + x [n, m] . M   (transpose != 0: x . M^T, the gradient direction). */
int sdn_ffd_coefficients(const float* x, const float* M, const float* base, int n, int m, int transpose, float* out,
                         sdnStream stream);

/* ==== textural branch: pix2pixHD-style generator / discriminator / encoder conv stacks ==================================
 * Reference operator surface: textural/models/networks.py (GlobalGenerator :211-239, ResnetBlock :244-283,
 * Encoder :286-308, NLayerDiscriminator :412-461), i.e. nn.Conv2d / nn.ConvTranspose2d / nn.ReflectionPad2d /
 * nn.InstanceNorm2d(affine=False, track_running_stats=True) / ReLU / LeakyReLU(0.2) / Tanh, run by cuDNN there.
 * Activations are channels-last fp32 [N, H, W, Cp], Cp = channel count padded to a multiple of 16 (pad channels hold
 * zeros).  Weights are pre-packed by sdn_conv_pack_weights into one fragment-ordered bf16 (hi, lo) buffer.  `precision` is 3
 * (bf16x3 split products, fp32-class results; the default everywhere) or 1 (plain bf16).
 * dy / dx tap tables are HOST arrays (int8, at most 64 taps). */

/* out[n, qy*ostride+py, qx*ostride+px, co] (=|+=) act(bias[co] + sum_t sum_ci f(in[n, qy*istride+dy[t], qx*istride+dx[t], ci]) * W[co, t*Cip+ci])
 *   Conv2d forward (networks.py:218,224,261,291,297,420-437): ostride 1, istride = stride, dy = ky - pad;
 *   ConvTranspose2d forward (:233,303) and the data gradient of strided Conv2d: one call per output phase (py, px);
 *   pad_mode 0: outside = 0;  1: reflected (ReflectionPad2d folded in, :218,236,251,265).   in_relu: f = ReLU.
 *   act 0 none, 1 LeakyReLU(0.2), 2 tanh.   stats [N, SDN_STAT_SLOTS, Cop, 2] fp64 (zeroed by the caller): += sum, sum of
 *   squares of the pre-activation per (n, co), spread over SDN_STAT_SLOTS partial copies -- the InstanceNorm statistics.   w_packed: 2 * w_rows * Kp bf16 from sdn_conv_pack_weights. */
#define SDN_STAT_SLOTS 8
/*   Layers whose output grid cannot fill the chip are split over K.  workspace NULL: the K slices meet in `out` through float
 *   atomics (fastest; sums differ in the last bits run to run).  workspace (>= sdn_conv_gemm_workspace_bytes): every slice
 *   stores its partial tile and one pass adds them in slice order -- bit-reproducible results (torch's deterministic mode). */
int sdn_conv_gemm_workspace_bytes(int N, int OH, int OW, int Cop, size_t* out);
int sdn_conv_gemm(const float* in, int N, int IH, int IW, int Cip, float* out, int OH, int OW, int Cop, int QH, int QW,
                  int istride, int ostride, int py, int px, int ntaps, const int8_t* dy, const int8_t* dx, int pad_mode,
                  int in_relu, const void* w_packed, int Kp, int w_rows, const float* bias, int act,
                  double* stats, int accumulate, int precision, void* workspace, size_t workspace_bytes,
                  sdnStream stream);

/* The s*s PHASE launches of a ConvTranspose2d forward (networks.py:233,303) or of a strided Conv2d's data gradient -- one
 * sdn_conv_gemm call per output phase (py, px) so far -- as ONE launch (r05).  Common arguments as sdn_conv_gemm; per phase k <
 * nphase (<= 4): its output sub-grid QH[k] x QW[k] at (py[k], px[k]), ntaps[k] (<= 16) taps, its packed weights w_packed[k]
 * (sdn_conv_pack_weights of that phase's tap list, Kp[k] columns).  `taps`: HOST int8, per phase dy[ntaps[k]] then
 * dx[ntaps[k]], phases concatenated; QH .. ntaps, Kp, w_packed are HOST arrays.  No split K (a phase never owns the whole
 * output tensor); bias / act / stats / accumulate apply to every phase's outputs as in the single-phase call. */
int sdn_conv_gemm_phases(const float* in, int N, int IH, int IW, int Cip, float* out, int OH, int OW, int Cop, int istride,
                         int ostride, int nphase, const int32_t* QH, const int32_t* QW, const int32_t* py, const int32_t* px,
                         const int32_t* ntaps, const int8_t* taps, int pad_mode, int in_relu, const void* const* w_packed,
                         const int32_t* Kp, int w_rows, const float* bias, int act, double* stats, int accumulate,
                         int precision, sdnStream stream);

/* dw[r, t*Cc + c] += sum_{n,q} a(rows[n, q, r]) * b(gath[n, q*istride + d_t, c])   (autograd of the layers above wrt their
 * weights).  rows [N, QH, QW, Cr], gath [N, GH, GW, Cc], dw [Cr, ntaps*Cc] fp32 (zeroed by the caller).  splits: K slices
 * over the positions; workspace NULL: combined with float atomics; workspace of splits * Cr * ntaps*Cc floats: combined in
 * slice order (bit-reproducible). */
int sdn_conv_wgrad(const float* rows, const float* gath, float* dw, int N, int QH, int QW, int Cr, int GH, int GW, int Cc,
                   int istride, int ntaps, const int8_t* dy, const int8_t* dx, int pad_mode, int relu_rows,
                   int relu_gath, int splits, int precision, void* workspace, size_t workspace_bytes, sdnStream stream);
/* The same sum for stride-1 layers whose `rows` operand has only rows_used <= 8 meaningful channels (the heads:
 * networks.py:236 c7s1-3, :306 c7s1-5, :437 the discriminators' last 4x4 conv): exact fp32 on the vector ALUs with
 * the gathered operand's tile + halo resident in LDS, instead of a 32-row MFMA tile that re-gathers the input per tap. */
int sdn_conv_wgrad_narrow(const float* rows, const float* gath, float* dw, int N, int QH, int QW, int Cr, int rows_used,
                          int GH, int GW, int Cc, int ntaps, const int8_t* dy, const int8_t* dx, int pad_mode,
                          int relu_rows, int relu_gath, sdnStream stream);

/* The same weight gradient for the 7 x 7 layers with Cr == 16 (rows_used <= 16) on the matrix cores (conv_whead.hip, r06):
 * the generator head 64 -> 3, the encoder head 16 -> 5 and the encoder stem 3 -> 16 (networks.py:236, 306, 291).  bf16 x 3
 * split products with fp32 accumulation, the precision class of sdn_conv_wgrad.  The taps must be a dense 7 x 7 window (any
 * order), Cc 16 or 64; arguments, dw layout and the caller's zero fill as sdn_conv_wgrad_narrow.  Its workgroups meet in dw
 * through float atomics: not for the deterministic mode. */
int sdn_conv_wgrad_head_mfma(const float* rows, const float* gath, float* dw, int N, int QH, int QW, int Cr, int rows_used,
                             int GH, int GW, int Cc, int ntaps, const int8_t* dy, const int8_t* dx, int pad_mode,
                             int relu_rows, int relu_gath, sdnStream stream);

/* Stride-1 convolutions with rows_used <= 8 output channels (the heads above, forward; and a data gradient restricted to
 * a few input channels): exact fp32 on the vector ALUs, the input tile + halo kept in LDS as channel planes.
 *   out[n, q, r] = act(bias[r] + sum_{dy,dx,c} f(in[n, q + (dy_min + dy, dx_min + dx), c]) * w_dense[dy, dx, c, r])
 * w_dense [KH, KW, Cip, RP] fp32 with RP = 1 / 4 / 8 for rows_used 1 / 2-4 / 5-8 (absent taps, channels, rows = 0);
 * KH == KW in {3, 4, 7}.  out [N, QH, QW, Cop]: channels >= rows_used are written as zeros.  pad_mode / in_relu / act as
 * sdn_conv_gemm. */
int sdn_conv_narrow_fwd(const float* in, int N, int IH, int IW, int Cip, float* out, int QH, int QW, int Cop,
                        int rows_used, const float* w_dense, int KH, int KW, int dy_min, int dx_min, int pad_mode,
                        int in_relu, const float* bias, int act, sdnStream stream);

/* The same head layers on the matrix cores (r05, csrc/conv_head.hip): 7 x 7 windows over Cip = 16 or 64 input channels,
 * rows_used <= 16 output channels in a 16-channel (padded) output tensor -- networks.py:236 (64 -> 3), :306 (16 -> 5) and the
 * stem's data gradient towards the encoder features.  v_mfma_f32_16x16x32_bf16, bf16 x 3 split products (fp32-class, not
 * bit-exact fp32 like sdn_conv_narrow_fwd), the input patch of an 8 x 32 output block staged once in LDS.
 * w_frag: [steps][2 (hi, lo)][64][8] bf16, steps = sdn_conv_head_steps: element (s, part, lane, j) = weight of output channel
 * lane % 16 at 8-channel slot u = 4 s + lane / 16 (tap = u / (Cip / 8) in window order ky * KW + kx, channels 8 (u % (Cip / 8)) + j),
 * zero for channels >= rows_used and slots behind the last tap.  Other arguments as sdn_conv_narrow_fwd.
 * r06 (ABI 7): `stats` (optional, [N, SDN_STAT_SLOTS, Cop, 2] fp64, zeroed by the caller) receives the InstanceNorm statistics of
 * bias + sum, as sdn_conv_gemm's epilogue files them -- the encoder's stem (networks.py:291-293: 3 -> 16 channels under
 * InstanceNorm) runs here; and Cop may be 64 over a 16-channel input (rows_used <= 64: the data gradient of the generator head
 * towards its 64 input channels, networks.py:236 backwards), w_frag then holds four row groups [4][steps][2][64][8]. */
int sdn_conv_head_steps(int Cip, int KH, int KW, int* steps);
int sdn_conv_head_mfma(const float* in, int N, int IH, int IW, int Cip, float* out, int QH, int QW, int Cop, int rows_used,
                       const void* w_frag, int KH, int KW, int dy_min, int dx_min, int pad_mode, int in_relu,
                       const float* bias, int act, double* stats, sdnStream stream);

/* InstanceNorm2d forward from the statistics the conv epilogue gathered (networks.py:27): first mr[n, c] = (mean, rstd)
 * ([N, Cp, 2] fp32, written here and kept for the backward pass) and the running_mean / running_var update torch does
 * in training mode (pointers may be NULL), then z <- (z - mean) * rstd in place (act 1: LeakyReLU(0.2) materialised);
 * out2 (optional) = z + f(res) (ResnetBlock, :281-283; f = ReLU when res_relu).  planes (optional, r04): bf16 (hi, lo)
 * operand planes (see sdn_split_planes) of what the tensor's consumers multiply -- out2 when there is one, else z --
 * through ReLU when planes_relu. */
int sdn_in_apply(float* z, const double* stats, float* mr, const float* res, float* out2, int N, int HW, int C, int Cp,
                 float eps, int act, int res_relu, float momentum, float* running_mean, float* running_var, void* planes,
                 long plane_stride, int planes_relu, sdnStream stream);
/* InstanceNorm2d (+ deferred ReLU / materialised LeakyReLU) backward, in place on g.  mode 0: stored = xhat; 1: stored =
 * xhat and consumers applied ReLU; 2: stored = LeakyReLU(xhat); | SDN_IN_BWD_SUMS_ZEROED (8): `sums` arrives zeroed (else the call
 * clears it with one fill launch).  mr from sdn_in_apply; sums: [N, Cp, 2] fp64 scratch.
 * planes (optional): the result also as bf16 operand planes (for sdn_conv_tile / sdn_conv_wgrad_tile). */
#define SDN_IN_BWD_SUMS_ZEROED 8
int sdn_in_bwd(float* g, const float* stored, const float* mr, double* sums, int N, int HW, int Cp, int mode, void* planes,
               long plane_stride, sdnStream stream);
/* layers without a norm: g <- g * act'(y) in place (act 0 none, 1 LeakyReLU, 2 tanh, 3 deferred ReLU) and
 * bias_grad[c] += sum g (optional, [Cp]).  planes (optional): the result also as bf16 operand planes. */
int sdn_act_bwd(float* g, const float* y, float* bias_grad, long npos, int Cp, int act, void* planes, long plane_stride,
                sdnStream stream);
/* adjoint of nn.ReflectionPad2d(pad): gp [N, H+2pad, W+2pad, Cp] -> out [N, H, W, Cp] (+= with accumulate). */
int sdn_reflect_fold(const float* gp, float* out, int N, int H, int W, int Cp, int pad, int accumulate, sdnStream stream);
/* Logical matrix Wm[r, k] = w[r*sr + c*sc + tapidx[t]] with k = t*Ccp + c -- or, when Ccp % 32 == 0 and Kp == ntaps*Ccp,
 * k = ((c/32)*ntaps + t)*32 + c%32 (channel-block-major: the taps of a 32-channel block are adjacent K steps of
 * sdn_conv_gemm, which applies the same rule to its gather) --, zero padded to [rows, Kp] (rows % 32 == 0, Kp % 32 == 0),
 * split into bf16 hi / lo and stored in MFMA fragment order: 2 * rows * Kp bf16 at
 *   packed[(((r/32) * (Kp/16) + k/16) * 2 + part) * 512 + (r%32 + 32*((k%16)/8)) * 8 + k%8],  part 0 = hi, 1 = lo.
 * tapidx is a DEVICE int32 array.  (sr, sc) select Conv2d [O,I,kh,kw] / ConvTranspose2d [I,O,kh,kw], forward /
 * data-gradient orientation.  Ccp % 8 == 0 (a thread packs eight columns of one row / tap; SDN_EINVAL otherwise). */
int sdn_conv_pack_weights(const float* w, int R, int C, long sr, long sc, const int32_t* tapidx, int ntaps, int Ccp,
                          int Kp, int rows, void* packed, sdnStream stream);
/* ---- r04: tiled MFMA kernels on bf16 operand PLANES (csrc/conv_tile.hip, conv_wtile.hip, conv_planes.hip).  Same layers,
 * same arithmetic (three bf16 products of split operands, fp32 accumulation) as sdn_conv_gemm / sdn_conv_wgrad, i.e. the
 * Conv2d / ConvTranspose2d forward, data gradient and weight gradient that /root/reference/textural/models/networks.py:211-283,
 * 412-461 leaves to cuDNN; the operands arrive pre-split and are copied to LDS by LDS-DMA.
 * A plane pair is [2][n] bf16: plane 0 = hi = bf16(x), plane 1 = lo = bf16(x - hi), `plane_stride` elements apart
 * (>= n, a multiple of 8).  relu != 0 splits max(x, 0) -- the deferred ReLU of the conv chains. */
int sdn_split_planes(const float* x, long n, int relu, void* planes, long plane_stride, sdnStream stream);
/* A chain's input: the NCHW fp32 tensors the caller would torch.cat along the channels (pix2pixHD_model.py:155-166, 199-210;
 * networks.py:238-239 then runs the first Conv2d on it) written side by side into ONE channels-last buffer out [N, H, W, Cp],
 * Cp >= sum(channels), pad channels zero.  parts / channels: HOST arrays of nparts (<= 8) device pointers / channel counts;
 * every part dense [N, channels[k], H, W].  Cp <= 128. */
int sdn_assemble_nhwc(const float* const* parts, const int32_t* channels, int nparts, int N, int H, int W, int Cp, float* out,
                      sdnStream stream);
/* K-major weights for sdn_conv_tile: packed[r][step][part][32] bf16 (part 0 = hi, 1 = lo), step = cb * ntaps + t over
 * 32-channel blocks cb and taps t, element = W[r, cb*32 + k%32, tap t]; rows >= R a multiple of 64, Ccp % 32 == 0;
 * 2 * rows * ntaps * Ccp bf16.  (sr, sc, tapidx) as sdn_conv_pack_weights. */
int sdn_conv_pack_weights_kmajor(const float* w, int R, int C, long sr, long sc, const int32_t* tapidx, int ntaps, int Ccp,
                                 int rows, void* packed, sdnStream stream);
/* sdn_conv_gemm's contract on planes: in_planes [2][N,IH,IW,Cip] (Cip % 32 == 0), out fp32 [N,OH,OW,Cop]; geometry, taps,
 * pad_mode, bias, act, stats, accumulate as sdn_conv_gemm.  out_planes (optional): the stored value, (ReLU'd when
 * planes_relu,) split, for the next layer.  w_rows >= Cop rounded up to the N tile (128 for Cop > 64, else 64).
 * ksplit > 1 (dense launches only: ostride 1, QH x QW = OH x OW, no act / stats / planes / accumulate): the LAST 256-position
 * tile of every image is computed as ksplit K slices by ksplit workgroups and summed with float atomics (summation order
 * not fixed) into rows the launcher zeroes -- for grids like the 26 x 80 data gradient of the residual layers, whose 32
 * leftover rows per image would otherwise cost a second round of the 256 CUs. */
int sdn_conv_tile(const void* in_planes, long plane_stride, int N, int IH, int IW, int Cip, float* out, void* out_planes,
                  long out_plane_stride, int planes_relu, int OH, int OW, int Cop, int QH, int QW, int istride, int ostride,
                  int py, int px, int ntaps, const int8_t* dy, const int8_t* dx, int pad_mode, const void* w_kmajor,
                  int w_rows, const float* bias, int act, double* stats, int accumulate, int ksplit, sdnStream stream);
/* Stride-1 convolutions with the input patch staged in LDS (csrc/conv_halo.hip): sdn_conv_tile's contract for launches with
 * istride = ostride = 1, py = px = 0, QH x QW = OH x OW whose taps fill a kh x kw window (kh * kw = ntaps >= 9) -- the 3x3
 * residual-block layers and the 4x4 stride-1 discriminator layers of textural/models/networks.py:244-283, 431-442, forward and
 * data gradient.  A workgroup owns a TH x TW block of output positions and copies the block's (TH + kh - 1) x (TW + kw - 1)
 * input patch once per 32-channel block instead of one activation tile per tap.  Needs Cop > 64 (128-channel N tiles).
 * sdn_conv_halo_blocks: the block shape the launcher picks and the grid size (blocks = 0: no block fits the window). */
int sdn_conv_halo_blocks(int N, int OH, int OW, int Cop, int kh, int kw, int* TH, int* TW, long* blocks);
int sdn_conv_halo(const void* in_planes, long plane_stride, int N, int IH, int IW, int Cip, float* out, int OH, int OW, int Cop,
                  int ntaps, const int8_t* dy, const int8_t* dx, int pad_mode, const void* w_kmajor, int w_rows,
                  const float* bias, int act, double* stats, int accumulate, sdnStream stream);
/* sdn_conv_wgrad's contract on planes: rows_planes [2][N*QH*QW, Cr], gath_planes [2][N,GH,GW,Cc] (ReLU already applied
 * where the fp32 entry point took relu_* flags); dw [Cr, ntaps*Cc] fp32 is ADDED to (zeroed by the caller): the work is
 * split stream-K style over one workgroup per CU and the parts of a tile meet in float atomics. */
int sdn_conv_wgrad_tile(const void* rows_planes, long rows_stride, const void* gath_planes, long gath_stride, float* dw,
                        int N, int QH, int QW, int Cr, int GH, int GW, int Cc, int istride, int ntaps, const int8_t* dy,
                        const int8_t* dx, int pad_mode, sdnStream stream);
/* grad_w[r*sr + c*sc + tapidx[t]] (+)= dw[r, t*Ccp + c]  (inverse of the packing map, for sdn_conv_wgrad's output).
 * accumulate 0: plain stores (a tap list covering the whole window defines every element: grad_w needs no zero fill).
 * Ccp % 4 == 0 and dw 16-byte aligned (16-byte loads of four columns of one row / tap; SDN_EINVAL otherwise). */
int sdn_conv_unpack_grad(const float* dw, int R, int C, long sr, long sc, const int32_t* tapidx, int ntaps, int Ccp,
                         float* grad_w, int accumulate, sdnStream stream);

/* ---- derender3d encoder: torchvision ResNet-18 behind geometric/derender3d/models/derenderer.py:25-27,48 ----------------------
 * Its convolutions are sdn_conv_gemm / sdn_conv_wgrad launches (bias-free 7x7 s2, 3x3 s1/s2, 1x1 s2); the entries below are
 * the rest of the network on channels-last fp32 tensors [rows = N*H*W, C] (C % 4 == 0; C <= 64 a power of two, else C % 64 == 0).
 *
 * nn.BatchNorm2d (+ the BasicBlock tail `relu(bn(x) + identity)`):  out = f(x * scale + shift + res), f = ReLU when relu,
 * scale = gamma * rstd, shift = beta - mean * scale.  training: batch mean / biased variance over the rows (sums: [C,2] fp64
 * scratch), running_mean / running_var (may be NULL) updated with momentum and the unbiased variance; eval: the running
 * statistics.  mr [C,2] = (mean, rstd) and ss [C,2] = (scale, shift) are written for the backward pass.  res may be NULL. */
int sdn_bn_forward(const float* x, long rows, int C, const float* gamma, const float* beta, float* running_mean,
                   float* running_var, float momentum, float eps, int training, const float* res, int relu, float* out,
                   float* mr, float* ss, double* sums, sdnStream stream);
/* Backward of the above.  gm [rows,C] <- g masked by the ReLU (out > 0): also the gradient of `res`.  dx <- gradient wrt x.
 * sums [C,2] fp64 <- (sum gm, sum gm * xhat) = (d beta, d gamma). */
int sdn_bn_backward(const float* g, const float* out, const float* x, const float* mr, const float* gamma, long rows, int C,
                    int training, int relu, float* gm, float* dx, double* sums, sdnStream stream);
/* nn.MaxPool2d(kernel_size=3, stride=2, padding=1) of torchvision's ResNet stem on [N,H,W,C]; idx [N,OH,OW,C] int8 = window
 * tap of the first maximum, OH = (H - 1) / 2 + 1.  The backward pass gathers (deterministic, no atomics). */
int sdn_maxpool3x3s2_fwd(const float* x, int N, int H, int W, int C, float* out, int8_t* idx, sdnStream stream);
int sdn_maxpool3x3s2_bwd(const float* g, const int8_t* idx, int N, int H, int W, int C, float* gin, sdnStream stream);
/* nn.AdaptiveAvgPool2d(1) (derenderer.py:26): forward x [N,HW,C] -> out [N,C]; backward (flag) x = g [N,C] -> out [N,HW,C]. */
int sdn_avgpool_global(const float* x, int N, int HW, int C, float* out, int backward, sdnStream stream);

/* Instance-wise average pooling of the Encoder (networks.py:310-325): x [N, C, HW] fp32 (NCHW), seg [N, HW] int32 dense
 * segment ids in [0, K) (ids are unique across the batch, as after networks.py:313-316).  sums [C, K] and counts [K] are
 * caller-owned scratch that is also the result table (sums[c, k] / counts[k] = mean feature of segment k, what
 * generate_feat_dict reports, :327-346); out [N, C, HW] = the mean of each pixel's segment.  The backward pass is the
 * same call on the incoming gradient. */
int sdn_segment_mean(const float* x, const int32_t* seg, int N, int C, int HW, int K, float* sums, float* counts,
                     float* out, sdnStream stream);

/* nn.AvgPool2d(3, stride=2, padding=[1, 1], count_include_pad=False): the input pyramid of MultiscaleDiscriminator
 * (networks.py:392, 406) and LocalEnhancer (:190).  in [N,C,H,W] / out [N,C,OH,OW] (OH = (H-1)/2+1) addressed through HOST
 * arrays of 4 element strides (n, c, h, w), so NCHW tensors and channels-last views need no copy; inner_c: threads run channel
 * fastest (channels-last storage).  _bwd: g [N,C,OH,OW] -> gin [N,C,H,W] (H, W = the INPUT size), gather form, no atomics. */
int sdn_avgpool3x3s2_fwd(const float* in, int N, int C, int H, int W, const long* in_strides, float* out,
                         const long* out_strides, int inner_c, sdnStream stream);
int sdn_avgpool3x3s2_bwd(const float* g, int N, int C, int H, int W, const long* g_strides, float* gin,
                         const long* gin_strides, int inner_c, sdnStream stream);

/* torch.nn.L1Loss() between two fp32 tensors of the same dense memory layout, flattened to n elements: `criterionFeat`
 * of the textural model (textural/models/pix2pixHD_model.py:86; discriminator feature matching :213-221, image
 * reconstruction).  fwd: out[0] = mean |a - b| (sum: one fp64 of device scratch).  bwd: grad_a = sgn(a - b) *
 * grad_out[0] / n, grad_b = -grad_a; either may be null.  All pointers are device pointers, 16-byte aligned. */
int sdn_l1_loss_fwd(const float* a, const float* b, long n, double* sum, float* out, sdnStream stream);
int sdn_l1_loss_bwd(const float* a, const float* b, long n, const float* grad_out, float* grad_a, float* grad_b,
                    sdnStream stream);
/* The loss of the test-time optimisation loop, geometric/scripts/main.py:445-451:
 *     loss = mean( mse_loss(masks, target, reduce=False) [* (1 - ignore)] + 100 * mean(ffd ** 2) )
 * fused: forward = one partial sum per block + one finishing wave that adds them in block order and writes out[0] (no
 * atomics: the same bits every run); backward = d loss / d masks and d loss / d ffd (either may be NULL) in one launch,
 * scaled by grad_out[0].  sums: SDN_SIL_LOSS_SUMS doubles of device scratch, no initialisation needed, kept for the
 * backward call (which reads the first three).  n = elements of masks / target / ignore (ignore may be NULL), nffd =
 * elements of ffd. */
#define SDN_SIL_LOSS_BLOCKS 512
#define SDN_SIL_LOSS_SUMS (3 + 3 * SDN_SIL_LOSS_BLOCKS)
int sdn_silhouette_loss_fwd(const float* masks, const float* target, const float* ignore, long n, const float* ffd, long nffd,
                            double* sums, float* out, sdnStream stream);
int sdn_silhouette_loss_bwd(const float* masks, const float* target, const float* ignore, long n, const float* ffd, long nffd,
                            const double* sums, const float* grad_out, float* grad_masks, float* grad_ffd, sdnStream stream);
/* Pose parameters of a frame's objects, derender3d/models/__init__.py:106-116: quat[n,4] = (cos(theta/2), 0, sin(theta/2), 0),
 * scales[n,3] = exp(log_scales); and the adjoint (g_quat / g_scales may be NULL = no gradient arrived). */
int sdn_pose_params(const float* theta, const float* log_scales, int n, float* quat, float* scales, sdnStream stream);
/* The whole pose algebra of Derenderer3d.render, derender3d/models/__init__.py:95-158, for a frame's n objects in one launch
 * each way (the reference: ~45 element-wise torch ops forward, as many autograd nodes backward):
 *   thetas = atan2(delta_1, delta_0);  rotations = (cos t/2, 0, sin t/2, 0);  scales = exp(log_scales);
 *   depths = sqrt(exp(log_depths) / (extent_0 extent_1));  center2ds = centre + translation2ds * extent;
 *   translations = depths * ray(center2ds), ray(row, col) = (col, -row, -1) / |.|;
 *   alphas = remainder(-(thetas - atan(t_x / t_z)) + pi, 2 pi) - pi;
 *   training != 0 (crop-centred camera, :139-150): persp = depths * ray(centre), zooms = (image_size / focals) / max(extent);
 *   else (object-centred, :152-153): persp = translations, zooms = render_size / (2 focals)  (the `zoom_tos`).
 * centre / extent / theta_deltas / translation2ds / center2ds [n,2], focals / log_depths / thetas / alphas / depths / zooms [n],
 * log_scales / scales / translations / persp [n,3], rotations [n,4]; all fp32 device arrays.  _bwd: any g_* input may be NULL
 * (no gradient arrived), any g_* output may be NULL (not wanted); zooms depend on no differentiable input. */
int sdn_pose_algebra(const float* centre, const float* extent, const float* focals, const float* theta_deltas,
                     const float* log_scales, const float* log_depths, const float* translation2ds, int n, int training,
                     float image_size, float render_size, float* thetas, float* alphas, float* rotations, float* scales,
                     float* depths, float* center2ds, float* translations, float* persp, float* zooms, sdnStream stream);
int sdn_pose_algebra_bwd(const float* centre, const float* extent, const float* theta_deltas, const float* thetas,
                         const float* scales, const float* depths, const float* center2ds, const float* translations, int n,
                         int training, const float* g_thetas, const float* g_alphas, const float* g_rotations,
                         const float* g_scales, const float* g_depths, const float* g_center2ds, const float* g_translations,
                         const float* g_persp, float* g_theta_deltas, float* g_log_scales, float* g_log_depths,
                         float* g_translation2ds, sdnStream stream);
int sdn_pose_params_bwd(const float* theta, const float* scales, const float* g_quat, const float* g_scales, int n,
                        float* g_theta, float* g_log_scales, sdnStream stream);

/* ---- per-frame compositing of the rendered objects: geometric/scripts/main.py:541-602 ---------------------------------------
 * masks [n,R,R], normals [n,3,R,R], depth_maps [n,R,R], zooms [n] (device).  objs: DEVICE int32 [m,7] rows
 * (object index, paste size, left, top, first row of its `bounds` table, first element of its coefficient tables, ksize)
 * in painter's order (far first); bounds [.,2] = (first source index, count) per output index, kk8 = Pillow's 22-bit
 * fixed-point bilinear weights, kkf = the same weights as doubles (Resample.c precompute_coeffs / normalize_coeffs_8bpc,
 * prepared by derender3d/compositing.py).  inst [H,W], nrm [3,H,W], dep [H,W] are updated where an object covers the
 * pixel (the caller initialises them to 0 / 0.5 / 1 as main.py:545-547 does).  Bit-identical to the PIL path. */
int sdn_composite_frame(const float* masks, const float* normals, const float* depth_maps, const float* zooms, int n,
                        int R, const int32_t* objs, int m, const int32_t* bounds, const int32_t* kk8, const double* kkf,
                        int H, int W, float* inst, float* nrm, float* dep, sdnStream stream);

/* ---- PerspectiveTransform: derender3d/models/transforms.py:102-158, all objects of a frame at once -----------------------
 * out[b,v] = zoom( shear( R(quat[b]) (verts[b,v] * scales[b]) + trans[b] ) ),  shear: x -= x0/z0 * z, y -= y0/z0 * z with
 * (x0,y0,z0) = persp[b].  Test-time form (zoom_fixed NULL, :147-158): zooms[b] = min_v |z| / max(|x|,|y|) * zoom_to[b];
 * training form (zoom_fixed [n], :139-150, also what the optimisation loop of scripts/main.py:433-456 runs because it puts
 * the model in train mode): zooms[b] = zoom_fixed[b], zoom_to unused (may be NULL).  z /= zooms[b] either way.
 * key: n * (1 + ceil(V / 256)) uint64 (caller-owned, no initialisation needed, kept for the backward pass): key[b] = bits
 * of zooms[b] / zoom_to[b] << 32 | the argmin vertex (0xffffffff in the training form); the rest is scratch (one minimum
 * per block of 256 vertices). */
/* bytes of the two caller-owned scratch buffers above / below for n objects of V vertices (so that a binding never restates
 * the formulas; the reference's PerspectiveTransform, derender3d/models/transforms.py:102-158, has no scratch -- its minimum and
 * its sums are torch reductions): *key_bytes for sdn_perspective_transform's `key`, *acc_bytes for sdn_perspective_transform_bwd's `acc`. */
int sdn_perspective_transform_scratch(int n, int V, size_t* key_bytes, size_t* acc_bytes);
int sdn_perspective_transform(const float* verts, const float* scales, const float* quat, const float* trans,
                              const float* persp, const float* zoom_to, const float* zoom_fixed, int n, int V, float* out,
                              float* zooms, void* key, sdnStream stream);
/* gradients of the above given g_out [n,V,3] and (optional) g_zooms [n]; acc: 36 n floats of scratch (no initialisation
 * needed).  After the training
 * form pass zoom_to = ones [n]: g_zoom_to[b] * zoom_to / zoom_fixed[b] ... i.e. g_zoom_to[b] / zoom_fixed[b] is then
 * d loss / d zoom_fixed[b].   g_persp may equal g_trans (one tensor passed as both
 * translations): the two gradients are then added into it (r06). */
int sdn_perspective_transform_bwd(const float* verts, const float* scales, const float* quat, const float* trans,
                                  const float* persp, const float* zoom_to, int n, int V, const float* out,
                                  const void* key, const float* g_out, const float* g_zooms, float* g_verts,
                                  float* g_scales, float* g_quat, float* g_trans, float* g_persp, float* g_zoom_to,
                                  float* acc, sdnStream stream);

/* ---- Mask R-CNN custom ops (only `--source maskrcnn` of geometric/scripts/main.py:642-661 needs them) -------
 * Greedy NMS: geometric/maskrcnn/nms/src/nms.c:4-69 (cpu_nms) / nms_cuda.c:17-67 + cuda/nms_kernel.cu:26-82.
 * boxes_sorted [n,4] and areas_sorted [n] (= (x2-x1+1)*(y2-y1+1), as pth_nms.py computes them) are already in descending
 * score order.  keep [n] int64 receives the kept positions (indices into the sorted arrays, ascending), *count their
 * number -- both DEVICE memory: the greedy pass runs on the device too (the reference copies the mask to the host).
 * strict 0: suppress when IoU >= thresh (cpu_nms, nms.c:59); 1: IoU > thresh (nms_kernel.cu:66).  Workspace: query first. */
int sdn_nms_workspace_bytes(int n, size_t* out);
int sdn_nms(const float* boxes_sorted, const float* areas_sorted, int n, float thresh, int strict, long long* keep,
            long long* count, void* workspace, size_t workspace_bytes, sdnStream stream);
/* crop_and_resize: geometric/maskrcnn/roialign/roi_align/src/crop_and_resize.c:7-158 (forward), :160-251 (backward);
 * CUDA twins cuda/crop_and_resize_kernel.cu:10-185.  image [B,C,H,W]; boxes [n,4] = (y1,x1,y2,x2) normalised to [0,1];
 * box_index [n] int32 in [0,B); crops [n,C,crop_h,crop_w]: bilinear samples, `extrapolation` outside the image.  The
 * backward pass zeroes grads_image [B,C,H,W] and scatter-adds (float atomics, as the reference's CUDA path). */
int sdn_crop_and_resize_fwd(const float* image, int B, int C, int H, int W, const float* boxes, const int32_t* box_index,
                            int n, int crop_h, int crop_w, float extrapolation, float* crops, sdnStream stream);
int sdn_crop_and_resize_bwd(const float* grads, const float* boxes, const int32_t* box_index, int n, int crop_h, int crop_w,
                            float* grads_image, int B, int C, int H, int W, sdnStream stream);

/* ==== launch lists: one host call per conv-chain pass ===========================================================================
 * In the reference one pass over a network is `self.model(input)` (textural/models/networks.py:238-239, 306-308, 395-407)
 * and `loss.backward()` (textural/train.py:88-95): PyTorch walks the module list / the autograd graph and launches
 * kernel after kernel.  Here a pass of a conv chain is 50-400 launches of the entry points above whose scalar arguments
 * depend only on the chain and the input shape.  The host plans them ONCE into an array of sdn_op records and replays the
 * array with one call per pass: sdn_program_run resolves every record's pointer arguments from a caller-built table of
 * DEVICE pointers (`slots`), and calls the launcher the record names -- the very entry points of this header, in array
 * order, each on the main or the side stream.  Nothing is fused or re-ordered: a program is the launch sequence, stored.
 *
 * Record layout: code = SDN_OP_*; stream 0 = main, 1 = side; buf[k] = slot index of the k-th pointer argument of that
 * entry point (in declaration order, -1 = NULL); i[] / f[] / l[] = its int / float / long-or-size_t arguments in declaration
 * order; taps = byte offset of the op's `int8 dy[ntaps], dx[ntaps]` pair in the program's tap blob (or -1).  Conv records
 * may carry their algorithmic GFLOP (true channel counts) in f[3] for the timing slots (sdn_timing_declare_work). */
typedef struct sdn_op {
    int32_t code, stream;
    int32_t buf[8];
    int32_t i[40];
    float f[4];
    int64_t l[2];
    int32_t taps, reserved;
} sdn_op;

enum {
    SDN_OP_CONV_GEMM = 1,     /* sdn_conv_gemm: buf in,out,w_packed,bias,stats,workspace; i N,IH,IW,Cip,OH,OW,Cop,QH,QW,istride,
                                 ostride,py,px,ntaps,pad_mode,in_relu,Kp,w_rows,act,accumulate,precision; l workspace_bytes */
    SDN_OP_CONV_NARROW_FWD,   /* sdn_conv_narrow_fwd: buf in,out,w_dense,bias; i N,IH,IW,Cip,QH,QW,Cop,rows_used,KH,KW,dy_min,
                                 dx_min,pad_mode,in_relu,act */
    SDN_OP_IN_APPLY,          /* sdn_in_apply: buf z,stats,mr,res,out2,running_mean,running_var,planes; i N,HW,C,Cp,act,res_relu,
                                 planes_relu; f eps,momentum; l plane_stride */
    SDN_OP_IN_BWD,            /* sdn_in_bwd: buf g,stored,mr,sums,planes; i N,HW,Cp,mode; l plane_stride */
    SDN_OP_ACT_BWD,           /* sdn_act_bwd: buf g,y,bias_grad,planes; l npos,plane_stride; i Cp,act */
    SDN_OP_REFLECT_FOLD,      /* sdn_reflect_fold: buf gp,out; i N,H,W,Cp,pad,accumulate */
    SDN_OP_CONV_WGRAD,        /* sdn_conv_wgrad: buf rows,gath,dw,workspace; i N,QH,QW,Cr,GH,GW,Cc,istride,ntaps,pad_mode,
                                 relu_rows,relu_gath,splits,precision; l workspace_bytes */
    SDN_OP_CONV_WGRAD_NARROW, /* sdn_conv_wgrad_narrow: buf rows,gath,dw; i N,QH,QW,Cr,rows_used,GH,GW,Cc,ntaps,pad_mode,
                                 relu_rows,relu_gath */
    SDN_OP_PACK_WEIGHTS,      /* sdn_conv_pack_weights: buf w,tapidx,packed; i R,C,ntaps,Ccp,Kp,rows; l sr,sc */
    SDN_OP_UNPACK_GRAD,       /* sdn_conv_unpack_grad: buf dw,tapidx,grad_w; i R,C,ntaps,Ccp,accumulate; l sr,sc */
    SDN_OP_MEMSET,            /* hipMemsetAsync(buf[0], 0, l[0] bytes) */
    SDN_OP_COPY,              /* hipMemcpyAsync(buf[0] <- buf[1], l[0] bytes, device to device) */
    SDN_OP_ADD,               /* buf[0][k] = buf[1][k] + buf[2][k], k < l[0] floats (l[0] % 4 == 0; gradient of a tensor read twice) */
    SDN_OP_COLSUM,            /* buf[1][c] = sum over l[0] rows of buf[0][row, c], c < i[1], row pitch i[0] floats, in a fixed
                                 order (bias gradients in deterministic mode) */
    SDN_OP_FORK,              /* the side stream waits for everything enqueued on the main stream so far */
    SDN_OP_JOIN,              /* the main stream waits for everything enqueued on the side stream so far */
    SDN_OP_SPLIT_PLANES,      /* sdn_split_planes: buf x,planes; l n,plane_stride; i relu */
    SDN_OP_PACK_WEIGHTS_KMAJOR, /* sdn_conv_pack_weights_kmajor: buf w,tapidx,packed; i R,C,ntaps,Ccp,rows; l sr,sc */
    SDN_OP_CONV_TILE,         /* sdn_conv_tile: buf in_planes,out,out_planes,w_kmajor,bias,stats; l plane_stride,out_plane_stride;
                                 i N,IH,IW,Cip,planes_relu,OH,OW,Cop,QH,QW,istride,ostride,py,px,ntaps,pad_mode,w_rows,act,accumulate,ksplit */
    SDN_OP_CONV_HALO,         /* sdn_conv_halo: buf in_planes,out,w_kmajor,bias,stats; l plane_stride;
                                 i N,IH,IW,Cip,OH,OW,Cop,ntaps,pad_mode,w_rows,act,accumulate */
    SDN_OP_CONV_WGRAD_TILE,   /* sdn_conv_wgrad_tile: buf rows_planes,gath_planes,dw; l rows_stride,gath_stride;
                                 i N,QH,QW,Cr,GH,GW,Cc,istride,ntaps,pad_mode */
    SDN_OP_CONV_GEMM_PHASES,  /* sdn_conv_gemm_phases: buf in,out,w_packed[0..3],bias,stats; i N,IH,IW,Cip,OH,OW,Cop,istride,ostride,
                                 nphase,pad_mode,in_relu,w_rows,act,accumulate,precision, then per phase k: i[16+6k ..] = QH,QW,py,px,
                                 ntaps,Kp; taps = offset of the phases' concatenated (dy, dx) lists */
    SDN_OP_CONV_HEAD_MFMA,    /* sdn_conv_head_mfma: buf in,out,w_frag,bias,stats; i N,IH,IW,Cip,QH,QW,Cop,rows_used,KH,KW,dy_min,dx_min,
                                 pad_mode,in_relu,act */
    SDN_OP_CONV_WGRAD_HEAD,   /* sdn_conv_wgrad_head_mfma: the record of SDN_OP_CONV_WGRAD_NARROW */
    SDN_OP_CODES
};

typedef struct sdn_program sdn_program;
/* Copies the records and the tap blob (HOST memory) and validates codes, slot indices and tap offsets. */
int sdn_program_create(const sdn_op* ops, int n_ops, const int8_t* taps, size_t tap_bytes, int n_slots, sdn_program** out);
/* Replays the program.  slots: HOST array of n_slots DEVICE pointers.  side may equal main (one stream; FORK / JOIN are then
 * no-ops).  op_ms: NULL, or a HOST array of n_ops floats: every record is then bracketed by events on its stream, the call
 * synchronises both streams and reports each record's duration in milliseconds (a measurement mode, never the timed path).
 * On failure *failed_op (may be NULL) is the index of the record whose launcher returned the error. */
int sdn_program_run(const sdn_program* prog, void* const* slots, int n_slots, sdnStream main, sdnStream side, float* op_ms,
                    int* failed_op);
int sdn_program_destroy(sdn_program* prog);

/* ---- measurement aid (bench.py): when enabled, every sdn_rasterize_fwd brackets its k_raster_tiles launch with a
 * hipEvent pair on the launch stream; sdn_timing_read synchronises them, returns the summed kernel time and the
 * number of launches since the last read, and clears the list.  Off by default; process-wide. */
int sdn_timing_enable(int enable);
int sdn_timing_read(double* ms_total, long* launches);
/* the same for any timed kernel family: slot 0 k_raster_tiles, 1 the silhouette edge-gradient kernels, 2 the MFMA forward /
 * data-gradient kernels (k_conv_gemm, k_conv_tile, k_conv_halo, k_conv_s2), 3 the MFMA weight-gradient kernels (k_conv_wgrad,
 * k_wgrad_tile), 4 the exact-fp32 head kernels (k_conv_narrow_fwd, k_wgrad_narrow), 5 k_raster_tiles_k1; *work (may be NULL)
 * receives the summed algorithmic work of the launches (flops for the conv slots, 0 for the raster slots). */
int sdn_timing_read_slot(int slot, double* ms_total, long* launches, double* work);
/* The conv launchers compute their work from the PADDED channel counts they are handed.  A caller that knows the layer's
 * true channel counts declares the work (flops) of the next timed launch this thread issues; sdn_program_run does so for
 * every record whose f[3] is non-zero (f[3] = the record's algorithmic GFLOP). */
int sdn_timing_declare_work(double work);

#ifdef __cplusplus
}
#endif
#endif /* SDN_HIP_H */
