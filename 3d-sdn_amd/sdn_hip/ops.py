"""torch.autograd wrappers over the C ABI (device tensors in, device tensors out, no host hops).

Each Function mirrors one piece of the reference's Chainer graph (paths under
/root/reference/geometric/):
  ProjectVertices  look / look_at / perspective      neural_renderer/look.py, look_at.py, perspective.py
  GatherFaces      vertices_to_faces (+ fill_back)   neural_renderer/vertices_to_faces.py, renderer.py:41
  FaceNormals      normalize(cross(v10, v12))        derender3d/models/renderer.py:66-76
  RasterizeMaps    Rasterize + flip + 2x2 pool       neural_renderer/rasterize.py:19-974
"""
import numpy as np
import torch

from . import (AA, ACCUMULATE, ALPHA, COUNT_WORK, DEPTH, FACE_COLOR, K1_COVERAGE, RGB, SAVE_MAPS, SERIAL_EDGES, STREAM_FACES, check, lib, ptr, raster_bwd_workspace,
               raster_workspace, stream, want)

CAMERA_NONE, CAMERA_LOOK, CAMERA_LOOK_AT = 0, 1, 2

import threading

_tls = threading.local()


class verification:
    """Thread-local verification / measurement switches for the rasterizer (no process-wide state: two threads -- e.g.
    nn.DataParallel replicas -- never see each other's settings; the backward pass uses what its forward call saw):
        serial_edges  backward walks every edge serially in the reference's exact summation order (SDN_SERIAL_EDGES)
        stream_faces  forward skips the per-tile face lists, i.e. forces the list-overflow path (SDN_STREAM_FACES)
        count_work    forward runs the counting build of k_raster_tiles; `last_work` then holds (candidate pixel tests,
                      tests passed, depth keys) of the most recent call on this thread (SDN_COUNT_WORK)
    Usage: `with ops.verification(serial_edges=True): ...`."""

    def __init__(self, serial_edges=None, stream_faces=None, count_work=None):
        new = dict(serial_edges=serial_edges, stream_faces=stream_faces, count_work=count_work)
        self.new = {k: v for k, v in new.items() if v is not None}   # unspecified switches keep their value (nesting)

    def __enter__(self):
        self.old = {k: getattr(_tls, k, False) for k in self.new}
        for k, v in self.new.items():
            setattr(_tls, k, bool(v))
        return self

    def __exit__(self, *a):
        for k, v in self.old.items():
            setattr(_tls, k, v)
        return False


def _switch(name):
    return getattr(_tls, name, False)


_K1 = [False]


def set_k1_coverage(flag):
    """Process-wide: forward rasterizations use the coverage rule of the reference's DEFAULT kernel K1 (rasterize.py:102-236,
    what `scripts/env.sh:11` selects through NEURAL_RENDERER_UNSAFE=1) instead of the safe kernels' -- SDN_K1_COVERAGE, with
    exact depth ties settled by face order.  neural_renderer.use_unsafe_rasterizer / the environment variable set it."""
    _K1[0] = bool(flag)


def k1_coverage():
    return _K1[0]


def last_clocks():
    """shader-clock ticks per phase of the same counting launch (sdn_raster_phase_clocks), summed over its waves."""
    return getattr(_tls, 'last_clocks', None)


def last_work():
    """(candidates, passed, keys) of this thread's latest RasterizeMaps forward under verification(count_work=True)."""
    return getattr(_tls, 'last_work', None)


def perspective_width(angle):
    """tan(angle / 180. * 3.1416) in float32 (neural_renderer/perspective.py:10-13)."""
    a = np.float32(angle)
    a = np.float32(a / np.float32(180.))
    a = np.float32(a * np.float32(3.1416))
    return np.float32(np.tan(a, dtype=np.float32))


def _f32(t, name):
    return want(t, torch.float32, name)


_ptf_sizes = {}


def _ptf_scratch(n, V):
    """(key bytes, acc bytes) of sdn_perspective_transform / _bwd: the library's own answer, remembered per shape."""
    r = _ptf_sizes.get((n, V))
    if r is None:
        from . import perspective_transform_scratch
        if len(_ptf_sizes) > 256:
            _ptf_sizes.clear()
        r = _ptf_sizes[(n, V)] = perspective_transform_scratch(n, V)
    return r


class ProjectVertices(torch.autograd.Function):
    @staticmethod
    def forward(ctx, vertices, camera_mode, eye, direction, up, width, flip_x):
        v = _f32(vertices, 'vertices')
        if v.dim() != 3 or v.shape[2] != 3:
            raise ValueError('vertices must be [batch, n, 3]')
        bs, nv = v.shape[:2]
        eye = _f32(eye, 'eye')
        direction = _f32(direction, 'direction')
        up = _f32(up, 'up')
        width = _f32(width, 'width')
        for t, n in ((eye, 'eye'), (direction, 'direction'), (up, 'up')):
            if t is not None and tuple(t.shape) != (bs, 3):
                raise ValueError('%s must be [batch, 3]' % n)
        if width is not None and width.numel() != bs:
            raise ValueError('width must have one entry per batch element')
        out = torch.empty_like(v)
        check(lib().sdn_project_vertices(ptr(v), bs, nv, int(camera_mode), ptr(eye), ptr(direction), ptr(up),
                                         ptr(width), int(flip_x), ptr(out), stream()))
        ctx.save_for_backward(v, eye, direction, up, width)
        ctx.cfg = (int(camera_mode), int(flip_x))
        return out

    @staticmethod
    def backward(ctx, g):
        v, eye, direction, up, width = ctx.saved_tensors
        mode, flip_x = ctx.cfg
        g = g.contiguous()
        gv = torch.empty_like(v)
        bs, nv = v.shape[:2]
        check(lib().sdn_project_vertices_bwd(ptr(v), bs, nv, mode, ptr(eye), ptr(direction), ptr(up), ptr(width),
                                             flip_x, ptr(g), ptr(gv), stream()))
        return gv, None, None, None, None, None, None


class GatherFaces(torch.autograd.Function):
    @staticmethod
    def forward(ctx, vertices, faces_idx, fill_back):
        v = _f32(vertices, 'vertices')
        f = want(faces_idx, torch.int32, 'faces')
        if v.dim() != 3 or v.shape[2] != 3:
            raise ValueError('vertices must be [batch, n, 3]')
        if f.dim() != 3 or f.shape[2] != 3:
            raise ValueError('faces must be [batch, n, 3]')
        bs, nv = v.shape[:2]
        if f.shape[0] not in (1, bs):
            raise ValueError('faces batch %d does not match vertices batch %d' % (f.shape[0], bs))
        nf0 = f.shape[1]
        stride = 0 if (f.shape[0] == 1 and bs > 1) else nf0 * 3
        nf = 2 * nf0 if fill_back else nf0
        out = torch.empty((bs, nf, 3, 3), dtype=torch.float32, device=v.device)
        check(lib().sdn_gather_faces(ptr(v), ptr(f), bs, nv, nf0, stride, int(bool(fill_back)), ptr(out), stream()))
        ctx.save_for_backward(f)
        ctx.cfg = (bs, nv, nf0, stride, int(bool(fill_back)))
        ctx.set_materialize_grads(False)
        return out

    @staticmethod
    def backward(ctx, g):
        if g is None:   # no gradient reached the faces (e.g. the normal map of a silhouette-only loss): no launches
            return None, None, None
        (f,) = ctx.saved_tensors
        bs, nv, nf0, stride, fill_back = ctx.cfg
        g = g.contiguous()
        gv = torch.empty((bs, nv, 3), dtype=torch.float32, device=g.device)
        check(lib().sdn_gather_faces_bwd(ptr(g), ptr(f), bs, nv, nf0, stride, fill_back, ptr(gv), stream()))
        return gv, None, None


class FaceNormals(torch.autograd.Function):
    @staticmethod
    def forward(ctx, faces):
        f = _f32(faces, 'faces')
        if f.dim() != 4 or f.shape[2:] != (3, 3):
            raise ValueError('faces must be [batch, n, 3, 3]')
        out = torch.empty(f.shape[:2] + (3,), dtype=torch.float32, device=f.device)
        check(lib().sdn_face_normals(ptr(f), f.shape[0] * f.shape[1], ptr(out), stream()))
        ctx.save_for_backward(f)
        ctx.set_materialize_grads(False)
        return out

    @staticmethod
    def backward(ctx, g):
        if g is None:   # the colours took no gradient (autograd would otherwise hand over zeros and run the kernels)
            return None
        (f,) = ctx.saved_tensors
        g = g.contiguous()
        gf = torch.empty_like(f)
        check(lib().sdn_face_normals_bwd(ptr(f), ptr(g), f.shape[0] * f.shape[1], ptr(gf), stream()))
        return gf


class RasterizeMaps(torch.autograd.Function):
    """faces [bs,nf,3,3] (+ textures) -> (rgb [bs,3,R,R] | None, alpha [bs,R,R] | None, depth [bs,R,R] | None).

    `eps_alpha`: when both rgb and alpha are requested and eps_alpha is not None, the backward runs the
    silhouette term and the colour term as two separate passes with their own eps -- what two separate
    Rasterize calls (rasterize_silhouettes with the module default 1e-4 and rasterize with the
    Renderer's 1e-3, derender3d/models/renderer.py:37,90-92) would have produced.  With eps_alpha None the
    two terms share one pass and one eps like a single rasterize_rgbad call (rasterize.py:627-655).
    """

    @staticmethod
    def forward(ctx, faces, textures, image_size, anti_aliasing, near, far, eps, background_color, return_rgb,
                return_alpha, return_depth, eps_alpha, face_color):
        if not any((return_rgb, return_alpha, return_depth)):
            raise Exception('nothing to draw')  # neural_renderer/rasterize.py:25-27
        f = _f32(faces, 'faces')
        if f.dim() != 4 or f.shape[2:] != (3, 3):
            raise ValueError('faces must be [batch size, number of faces, 3, 3]')
        bs, nf = f.shape[:2]
        dev = f.device
        S = int(image_size) * 2 if anti_aliasing else int(image_size)
        R = int(image_size)
        flags = (RGB if return_rgb else 0) | (ALPHA if return_alpha else 0) | (DEPTH if return_depth else 0)
        flags |= AA if anti_aliasing else 0
        tex = None
        ts = 0
        if return_rgb:
            tex = _f32(textures, 'textures')
            if face_color:
                if tuple(tex.shape) != (bs, nf, 3):
                    raise ValueError('face colours must be [batch, faces, 3]')
                flags |= FACE_COLOR
                ts = 2
            else:
                if tex.dim() != 6 or tex.shape[0] != bs or tex.shape[1] != nf or tex.shape[5] != 3 or \
                        tex.shape[2] < 2 or tex.shape[2] != tex.shape[3] or tex.shape[3] != tex.shape[4]:
                    raise ValueError('textures must be [batch, faces, ts, ts, ts, 3] with ts >= 2')
                ts = tex.shape[2]
        need_grad = any(ctx.needs_input_grad[:2])
        if need_grad:
            flags |= SAVE_MAPS
        if _switch('stream_faces'):
            flags |= STREAM_FACES
        if _switch('count_work'):
            flags |= COUNT_WORK
        if k1_coverage():
            flags |= K1_COVERAGE
        bg = None
        bg_per_batch = 0
        if return_rgb:
            if isinstance(background_color, torch.Tensor):
                bg = background_color.to(device=dev, dtype=torch.float32).contiguous()
            else:
                from . import const_f32
                bg = const_f32(background_color if background_color is not None else (0, 0, 0), dev)
            bg_per_batch = 1 if bg.dim() == 2 else 0
        face_inv = torch.empty((bs, nf, 3, 3), dtype=torch.float32, device=dev)
        fim = wmap = dmap = rgbmap = None
        if need_grad:
            fim = torch.empty((bs, S, S), dtype=torch.int32, device=dev)
            wmap = torch.empty((bs, S, S, 3), dtype=torch.float32, device=dev)
            dmap = torch.empty((bs, S, S), dtype=torch.float32, device=dev)
            if return_rgb:
                rgbmap = torch.empty((bs, S, S, 3), dtype=torch.float32, device=dev)
        rgb = torch.empty((bs, 3, R, R), dtype=torch.float32, device=dev) if return_rgb else None
        alpha = torch.empty((bs, R, R), dtype=torch.float32, device=dev) if return_alpha else None
        depth = torch.empty((bs, R, R), dtype=torch.float32, device=dev) if return_depth else None
        ws = raster_workspace(bs, nf, S, dev)
        check(lib().sdn_rasterize_fwd(ptr(f), ptr(tex), ts, bs, nf, S, float(near), float(far), float(eps), ptr(bg),
                                      bg_per_batch, flags, ptr(face_inv), ptr(fim), ptr(wmap), ptr(dmap),
                                      ptr(rgbmap), ptr(rgb), ptr(alpha), ptr(depth), ptr(ws), ws.numel(), stream()))
        if flags & COUNT_WORK:
            import ctypes
            c = (ctypes.c_ulonglong * 3)()
            check(lib().sdn_raster_work_counters(ptr(ws), bs, nf, S, c, stream()))
            _tls.last_work = (int(c[0]), int(c[1]), int(c[2]))
            c8 = (ctypes.c_ulonglong * 8)()
            check(lib().sdn_raster_phase_clocks(ptr(ws), bs, nf, S, c8, stream()))
            _tls.last_clocks = dict(zip(('fetch_and_wait', 'lane_boxes', 'wave_boxes', 'thin', 'epilogue', 'total', 'longest_wave',
                                         'waves'), (int(v) for v in c8)))
            flags &= ~COUNT_WORK
        if need_grad:
            ctx.save_for_backward(f, tex, face_inv, fim, wmap, dmap, rgbmap)
        ctx.cfg = (ts, bs, nf, S, float(eps), None if eps_alpha is None else float(eps_alpha), flags,
                   SERIAL_EDGES if _switch('serial_edges') else 0)
        ctx.set_materialize_grads(False)
        return rgb, alpha, depth

    @staticmethod
    def backward(ctx, g_rgb, g_alpha, g_depth):
        f, tex, face_inv, fim, wmap, dmap, rgbmap = ctx.saved_tensors
        ts, bs, nf, S, eps, eps_alpha, flags, serial = ctx.cfg   # `serial`: what the forward call's thread had set
        base = (flags & (AA | FACE_COLOR)) | serial
        g_rgb = None if g_rgb is None else g_rgb.contiguous()
        g_alpha = None if g_alpha is None else g_alpha.contiguous()
        g_depth = None if g_depth is None else g_depth.contiguous()
        grad_faces = torch.empty_like(f)
        want_tex_grad = (flags & RGB) and ctx.needs_input_grad[1] and g_rgb is not None
        grad_tex = torch.empty_like(tex) if want_tex_grad else None
        L = lib()
        ws = raster_bwd_workspace(bs, nf, S, f.device)

        def run(fl, e, gr, ga, gd, gt):
            check(L.sdn_rasterize_bwd(ptr(f), ptr(tex), ts, bs, nf, S, e, fl, ptr(face_inv), ptr(fim), ptr(wmap),
                                      ptr(dmap), ptr(rgbmap), ptr(gr), ptr(ga), ptr(gd), ptr(grad_faces), ptr(gt),
                                      ptr(ws), ws.numel(), stream()))

        split = (eps_alpha is not None) and (flags & RGB) and (flags & ALPHA)
        if split:
            # pass 1 stores the silhouette term, pass 2 adds colour (+ depth) and the texture scatter
            run(base | ALPHA, eps_alpha, None, g_alpha, None, None)
            if g_rgb is not None or g_depth is not None:
                if grad_tex is not None:
                    grad_tex.zero_()
                run(base | RGB | (flags & DEPTH) | ACCUMULATE, eps, g_rgb, None, g_depth, grad_tex)
        else:
            e = eps if (flags & RGB) or eps_alpha is None else eps_alpha
            run((flags & ~SAVE_MAPS & ~STREAM_FACES) | serial, e, g_rgb, g_alpha, g_depth, grad_tex)
        gf = grad_faces if ctx.needs_input_grad[0] else None
        return (gf, grad_tex) + (None,) * 11


class RenderMapsFn(torch.autograd.Function):
    """(alpha [bs,R,R], normal [bs,3,R,R] | None, depth [bs,R,R] | None) of a frame's objects: sdn_render_maps_fwd / _bwd, ONE C
    call each way instead of ProjectVertices + 2 x GatherFaces + FaceNormals + RasterizeMaps and two element-wise ops
    (derender3d/models/renderer.py:216-272 three times per object in the reference).  `vertices` are the caller's (not
    x-flipped); the result equals Renderer.render_maps' composition of the separate Functions bit for bit."""

    @staticmethod
    def forward(ctx, vertices, faces_idx, fill_back, camera_mode, eye, direction, up, width, flip_x, image_size,
                anti_aliasing, near, far, eps, eps_alpha, background_color, want_normal, want_depth):
        import ctypes
        from . import const_f32
        v = _f32(vertices, 'vertices')
        f = want(faces_idx, torch.int32, 'faces')
        if v.dim() != 3 or v.shape[2] != 3:
            raise ValueError('vertices must be [batch, n, 3]')
        if f.dim() != 3 or f.shape[2] != 3:
            raise ValueError('faces must be [batch, n, 3]')
        bs, nv = v.shape[:2]
        if f.shape[0] not in (1, bs):
            raise ValueError('faces batch %d does not match vertices batch %d' % (f.shape[0], bs))
        nf0 = f.shape[1]
        stride = 0 if (f.shape[0] == 1 and bs > 1) else nf0 * 3
        dev = v.device
        R = int(image_size)
        need_grad = ctx.needs_input_grad[0]
        flags = (RGB if want_normal else 0) | (DEPTH if want_depth else 0) | (AA if anti_aliasing else 0)
        flags |= SAVE_MAPS if need_grad else 0
        if _switch('stream_faces'):
            flags |= STREAM_FACES
        if k1_coverage():
            flags |= K1_COVERAGE
        nstate, nbwd, nscr = ctypes.c_size_t(0), ctypes.c_size_t(0), ctypes.c_size_t(0)
        check(lib().sdn_render_maps_bytes(bs, nv, nf0, int(bool(fill_back)), R, flags, ctypes.byref(nstate), ctypes.byref(nbwd),
                                          ctypes.byref(nscr)))
        state = torch.empty(nstate.value, dtype=torch.uint8, device=dev)
        scratch = torch.empty(nscr.value, dtype=torch.uint8, device=dev)   # forward-only (tile lists): not saved for backward
        alpha = torch.empty((bs, R, R), dtype=torch.float32, device=dev)
        normal = torch.empty((bs, 3, R, R), dtype=torch.float32, device=dev) if want_normal else None
        depth = torch.empty((bs, R, R), dtype=torch.float32, device=dev) if want_depth else None
        bg = None
        bg_is_callers = isinstance(background_color, torch.Tensor)
        if want_normal:
            bg = background_color if bg_is_callers else \
                const_f32(background_color if background_color is not None else (0, 0, 0), dev)
            bg = bg.to(device=dev, dtype=torch.float32).contiguous()
            if bg.numel() != 3:
                raise ValueError('render_maps takes one background colour')
        cam = (int(camera_mode), _f32(eye, 'eye'), _f32(direction, 'direction'), _f32(up, 'up'), _f32(width, 'width'),
               int(bool(flip_x)))
        check(lib().sdn_render_maps_fwd(ptr(v), bs, nv, ptr(f), nf0, stride, int(bool(fill_back)), cam[0], ptr(cam[1]),
                                        ptr(cam[2]), ptr(cam[3]), ptr(cam[4]), cam[5], R, flags, float(near), float(far),
                                        float(eps), ptr(bg), ptr(alpha), ptr(normal), ptr(depth), ptr(state), state.numel(),
                                        ptr(scratch), scratch.numel(), stream()))
        if need_grad:
            # the background colour is handed to the backward call again (it re-derives the colour map).  A caller's own tensor
            # goes through save_for_backward so that an in-place change between forward and backward trips autograd's version
            # check instead of silently changing the normal / depth gradients; the const_f32 cache is never written
            if bg is not None and bg_is_callers:
                ctx.save_for_backward(v, f, state, cam[1], cam[2], cam[3], cam[4], bg)
                ctx.bg = None
            else:
                ctx.save_for_backward(v, f, state, cam[1], cam[2], cam[3], cam[4])
                ctx.bg = bg
            ctx.cfg = (bs, nv, nf0, stride, int(bool(fill_back)), cam[0], cam[5], R, flags & ~STREAM_FACES, float(eps),
                       float(eps_alpha), nbwd.value, SERIAL_EDGES if _switch('serial_edges') else 0)
        ctx.set_materialize_grads(False)
        return alpha, normal, depth

    @staticmethod
    def backward(ctx, g_alpha, g_normal, g_depth):
        saved = ctx.saved_tensors
        v, f, state, eye, direction, up, width = saved[:7]
        bg = saved[7] if len(saved) > 7 else ctx.bg
        bs, nv, nf0, stride, fill_back, mode, flip_x, R, flags, eps, eps_alpha, nbwd, serial = ctx.cfg
        if g_alpha is None and g_normal is None and g_depth is None:
            return (None,) * 18
        g_alpha = None if g_alpha is None else g_alpha.contiguous()
        g_normal = None if g_normal is None else g_normal.contiguous()
        g_depth = None if g_depth is None else g_depth.contiguous()
        ws = torch.empty(nbwd, dtype=torch.uint8, device=v.device)
        gv = torch.empty_like(v)
        check(lib().sdn_render_maps_bwd(ptr(v), bs, nv, ptr(f), nf0, stride, fill_back, mode, ptr(eye), ptr(direction), ptr(up),
                                        ptr(width), flip_x, R, flags | serial, eps, eps_alpha, ptr(bg), ptr(g_alpha),
                                        ptr(g_normal), ptr(g_depth), ptr(gv), ptr(state), state.numel(), ptr(ws), ws.numel(),
                                        stream()))
        return (gv,) + (None,) * 17


class FFDDecode(torch.autograd.Function):
    """verts [n, vmax, 3] = P [n, 3, ncoef] . Bt[cls] [ncoef, vmax]  (derender3d/models/transforms.py:97-99).
    With `constraint` [3 ncoef, 3 ncoef] (and `base` [3, ncoef]) the first argument holds raw coefficient rows [n, 3 ncoef] and
    P = base + coeffs . constraint is formed by the library as well (FFD.constrain, transforms.py:69-95, is linear): the
    whole decode is two launches each way and no torch op."""

    @staticmethod
    def forward(ctx, P, Bt, cls, constraint=None, base=None):
        P = _f32(P, 'P')
        Bt = _f32(Bt, 'Bt')
        cls = want(cls, torch.int32, 'cls')
        n = P.shape[0]
        ncoef = Bt.shape[1] if Bt.dim() == 3 else -1
        if constraint is not None:
            Cm = _f32(constraint, 'constraint')
            m = 3 * ncoef
            if P.dim() != 2 or P.shape[1] != m or tuple(Cm.shape) != (m, m) or (base is not None and base.numel() != m):
                raise ValueError('FFDDecode: coefficients [n, 3 ncoef], constraint [3 ncoef, 3 ncoef], base [3, ncoef]')
            base = None if base is None else _f32(base, 'base')
            coeffs = P
            P = torch.empty((n, 3, ncoef), dtype=torch.float32, device=coeffs.device)
            check(lib().sdn_ffd_coefficients(ptr(coeffs), ptr(Cm), ptr(base), n, m, 0, ptr(P), stream()))
        else:
            Cm = None
        if P.dim() != 3 or P.shape[1] != 3 or Bt.dim() != 3 or P.shape[2] != ncoef or cls.numel() != n:
            raise ValueError('FFDDecode: P [n,3,ncoef], Bt [classes,ncoef,vmax], cls [n]')
        vmax = Bt.shape[2]
        out = torch.empty((n, vmax, 3), dtype=torch.float32, device=P.device)
        check(lib().sdn_ffd_decode(ptr(Bt), ptr(P), ptr(cls), n, vmax, ncoef, ptr(out), stream()))
        if Cm is None:
            ctx.save_for_backward(Bt, cls)
        else:
            ctx.save_for_backward(Bt, cls, Cm)
        ctx.cfg = (n, vmax, ncoef)
        return out

    @staticmethod
    def backward(ctx, g):
        Bt, cls = ctx.saved_tensors[:2]
        n, vmax, ncoef = ctx.cfg
        g = g.contiguous()
        gP = torch.empty((n, 3, ncoef), dtype=torch.float32, device=g.device)
        check(lib().sdn_ffd_decode_bwd(ptr(Bt), ptr(cls), ptr(g), n, vmax, ncoef, ptr(gP), stream()))
        if len(ctx.saved_tensors) == 3:
            gc = torch.empty((n, 3 * ncoef), dtype=torch.float32, device=g.device)
            check(lib().sdn_ffd_coefficients(ptr(gP), ptr(ctx.saved_tensors[2]), None, n, 3 * ncoef, 1, ptr(gc), stream()))
            return gc, None, None, None, None
        return gP, None, None, None, None


class PoseParamsFn(torch.autograd.Function):
    """(rotations [n,4], scales [n,3]) = ((cos(theta/2), 0, sin(theta/2), 0), exp(log_scales)) -- Derenderer3d.render,
    derender3d/models/__init__.py:106-116: one launch each way instead of ten element-wise ones."""

    @staticmethod
    def forward(ctx, theta, log_scales):
        th = _f32(theta, 'theta').reshape(-1)
        n = th.shape[0]
        ls = _f32(log_scales, 'log_scales').reshape(n, 3)
        quat = torch.empty(n, 4, dtype=torch.float32, device=th.device)
        scales = torch.empty(n, 3, dtype=torch.float32, device=th.device)
        check(lib().sdn_pose_params(ptr(th), ptr(ls), n, ptr(quat), ptr(scales), stream()))
        ctx.save_for_backward(th, scales)
        ctx.shapes = (theta.shape, log_scales.shape)
        ctx.set_materialize_grads(False)
        return quat, scales

    @staticmethod
    def backward(ctx, g_quat, g_scales):
        th, scales = ctx.saved_tensors
        n = th.shape[0]
        want_t, want_s = ctx.needs_input_grad
        if g_quat is None and g_scales is None:
            return None, None
        gq = g_quat.contiguous() if g_quat is not None else None
        gs = g_scales.contiguous() if g_scales is not None else None
        gt = torch.empty(n, dtype=torch.float32, device=th.device) if want_t else None
        gl = torch.empty(n, 3, dtype=torch.float32, device=th.device) if want_s else None
        if gt is None and gl is None:
            return None, None
        check(lib().sdn_pose_params_bwd(ptr(th), ptr(scales), ptr(gq), ptr(gs), n, ptr(gt), ptr(gl), stream()))
        return (gt.reshape(ctx.shapes[0]) if gt is not None else None, gl.reshape(ctx.shapes[1]) if gl is not None else None)


class PoseAlgebraFn(torch.autograd.Function):
    """(thetas [n,1], alphas [n,1], rotations [n,4], scales [n,3], depths [n,1], center2ds [n,2], translations [n,3],
    persp [n,3], zooms [n,1]) of a frame's objects from the encoder outputs -- the pose algebra of Derenderer3d.render,
    derender3d/models/__init__.py:95-158, in one launch each way (sdn_pose_algebra / _bwd) instead of ~45 element-wise ops and
    their autograd nodes.  zooms: the given zooms of the training form, the `zoom_tos` of the test-time form."""

    @staticmethod
    def forward(ctx, centre, extent, focals, theta_deltas, log_scales, log_depths, translation2ds, training, image_size,
                render_size):
        c = _f32(centre, 'centre')
        n = c.shape[0]
        c = c.reshape(n, 2)
        e = _f32(extent, 'extent').reshape(n, 2)
        f = _f32(focals, 'focals').reshape(n)
        d = _f32(theta_deltas, 'theta_deltas').reshape(n, 2)
        ls = _f32(log_scales, 'log_scales').reshape(n, 3)
        ld = _f32(log_depths, 'log_depths').reshape(n)
        t2 = _f32(translation2ds, 'translation2ds').reshape(n, 2)
        dev = c.device
        new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)   # noqa: E731
        thetas, alphas, rot, scales, depths = new(n, 1), new(n, 1), new(n, 4), new(n, 3), new(n, 1)
        c2, trans, persp, zooms = new(n, 2), new(n, 3), new(n, 3), new(n, 1)
        check(lib().sdn_pose_algebra(ptr(c), ptr(e), ptr(f), ptr(d), ptr(ls), ptr(ld), ptr(t2), n, int(bool(training)),
                                     float(image_size), float(render_size), ptr(thetas), ptr(alphas), ptr(rot), ptr(scales),
                                     ptr(depths), ptr(c2), ptr(trans), ptr(persp), ptr(zooms), stream()))
        ctx.save_for_backward(c, e, d, thetas, scales, depths, c2, trans)
        ctx.training = int(bool(training))
        ctx.shapes = (theta_deltas.shape, log_scales.shape, log_depths.shape, translation2ds.shape)
        ctx.mark_non_differentiable(zooms)
        ctx.set_materialize_grads(False)
        return thetas, alphas, rot, scales, depths, c2, trans, persp, zooms

    @staticmethod
    def backward(ctx, g_thetas, g_alphas, g_rot, g_scales, g_depths, g_c2, g_trans, g_persp, _g_zooms):
        c, e, d, thetas, scales, depths, c2, trans = ctx.saved_tensors
        n = c.shape[0]
        gs = [g.contiguous() if g is not None else None for g in (g_thetas, g_alphas, g_rot, g_scales, g_depths, g_c2, g_trans,
                                                                  g_persp)]
        need = ctx.needs_input_grad[3:7]
        if all(g is None for g in gs) or not any(need):
            return (None,) * 10
        dev = c.device
        out = [torch.empty(shape, dtype=torch.float32, device=dev) if w else None
               for w, shape in zip(need, ((n, 2), (n, 3), (n,), (n, 2)))]
        check(lib().sdn_pose_algebra_bwd(ptr(c), ptr(e), ptr(d), ptr(thetas), ptr(scales), ptr(depths), ptr(c2), ptr(trans), n,
                                         ctx.training, *[ptr(g) for g in gs], *[ptr(o) for o in out], stream()))
        grads = [o.reshape(s) if o is not None else None for o, s in zip(out, ctx.shapes)]
        return (None, None, None, grads[0], grads[1], grads[2], grads[3], None, None, None)


class SilhouetteLossFn(torch.autograd.Function):
    """mean(mse_loss(masks, target, reduce=False) [* (1 - ignore)] + 100 * mean(ffd ** 2)) -- the loss of the test-time
    optimisation loop, geometric/scripts/main.py:445-451, forward in two launches (+ one memset), backward in one."""

    @staticmethod
    def forward(ctx, masks, target, ffd, ignore=None):
        m = _f32(masks, 'masks')
        t = _f32(target, 'target')
        if t.shape != m.shape:
            raise ValueError('target %s must have the shape of masks %s' % (tuple(t.shape), tuple(m.shape)))
        ig = None
        if ignore is not None:
            ig = _f32(ignore, 'ignore')
            if ig.shape != m.shape:
                raise ValueError('ignore must have the shape of masks')
        f = _f32(ffd, 'ffd') if ffd is not None else None
        sums = torch.empty(3 + 3 * 512, dtype=torch.float64, device=m.device)   # SDN_SIL_LOSS_SUMS
        out = torch.empty((), dtype=torch.float32, device=m.device)
        check(lib().sdn_silhouette_loss_fwd(ptr(m), ptr(t), ptr(ig), m.numel(), ptr(f), f.numel() if f is not None else 0,
                                            ptr(sums), ptr(out), stream()))
        ctx.save_for_backward(m, t, f, ig, sums)
        ctx.ffd_shape = ffd.shape if ffd is not None else None
        return out

    @staticmethod
    def backward(ctx, g):
        m, t, f, ig, sums = ctx.saved_tensors
        want_m, _, want_f, _ = ctx.needs_input_grad
        gm = torch.empty_like(m) if want_m else None
        gf = torch.empty_like(f) if (want_f and f is not None) else None
        if gm is None and gf is None:
            return None, None, None, None
        go = g.contiguous().reshape(1)
        check(lib().sdn_silhouette_loss_bwd(ptr(m), ptr(t), ptr(ig), m.numel(), ptr(f), f.numel() if f is not None else 0,
                                            ptr(sums), ptr(go), ptr(gm), ptr(gf), stream()))
        return gm, None, (gf.reshape(ctx.ffd_shape) if gf is not None else None), None


class PerspectiveTransformFn(torch.autograd.Function):
    """(vertices [n,V,3], zooms [n,1]) = zoom_fit(shear(R(q) (v * s) + t))  -- derender3d/models/transforms.py:102-158 for
    a whole frame in two launches (forward) / three (backward) instead of ~25 element-wise ops and a batched GEMM."""

    @staticmethod
    def forward(ctx, vertices, scales, rotations, translations, persp, zoom_tos, zooms_given=None):
        """zoom_tos [n,1]: zoom-to-fit (test-time form); zooms_given [n,1] (then zoom_tos is None): the training form."""
        v = _f32(vertices, 'vertices')
        n, V, three = v.shape
        if three != 3:
            raise ValueError('vertices must be [n, V, 3]')
        s = _f32(scales, 'scales').reshape(n, 3)
        q = _f32(rotations, 'rotations').reshape(n, 4)
        t = _f32(translations, 'translations').reshape(n, 3)
        p = _f32(persp, 'perspective_translations').reshape(n, 3)
        fixed = zooms_given is not None
        if fixed:
            zg = _f32(zooms_given, 'zooms').reshape(n)
            from . import const_f32
            zt = const_f32([1.0] * n, v.device)   # the backward kernels see zoom = key_ratio * 1 (a cached constant: no fill launch)
        else:
            zg = None
            zt = _f32(zoom_tos, 'zoom_tos').reshape(n)
        out = torch.empty_like(v)
        zooms = torch.empty(n, dtype=torch.float32, device=v.device)
        key = torch.empty(_ptf_scratch(n, V)[0], dtype=torch.uint8, device=v.device)   # key[n] + the per-block minima
        check(lib().sdn_perspective_transform(ptr(v), ptr(s), ptr(q), ptr(t), ptr(p), ptr(zt), ptr(zg), n, V, ptr(out),
                                              ptr(zooms), ptr(key), stream()))
        ctx.save_for_backward(v, s, q, t, p, zt, out, key, zg)
        ctx.shapes = (scales.shape, rotations.shape, translations.shape, persp.shape,
                      (zooms_given if fixed else zoom_tos).shape)
        ctx.fixed = fixed
        # Derenderer3d.render passes ONE tensor as translations and perspective_translations (derender3d/models/__init__.py:185-190):
        # autograd would add the two gradients with an element-wise launch; the library adds them when handed one pointer for both
        ctx.same_tp = translations is persp
        # (the zooms usually take no gradient -- the frame step's loss never reads them: autograd would hand backward a freshly
        # zero-filled tensor for them, one fill launch per step)
        ctx.set_materialize_grads(False)
        return out, zooms.reshape(n, 1)

    @staticmethod
    def backward(ctx, g_out, g_zooms):
        v, s, q, t, p, zt, out, key, zg = ctx.saved_tensors
        n, V, _ = v.shape
        dev = v.device
        if g_out is None and g_zooms is None:
            return (None,) * 7
        g_out = g_out.contiguous() if g_out is not None else torch.zeros_like(v)
        gz = g_zooms.reshape(n).contiguous() if g_zooms is not None else None
        gv = torch.empty_like(v)
        gs = torch.empty(n, 3, dtype=torch.float32, device=dev)
        gq = torch.empty(n, 4, dtype=torch.float32, device=dev)
        gt = torch.empty(n, 3, dtype=torch.float32, device=dev)
        gp = gt if ctx.same_tp else torch.empty(n, 3, dtype=torch.float32, device=dev)
        gzt = torch.empty(n, dtype=torch.float32, device=dev)
        acc = torch.empty(_ptf_scratch(n, V)[1], dtype=torch.uint8, device=dev)
        check(lib().sdn_perspective_transform_bwd(ptr(v), ptr(s), ptr(q), ptr(t), ptr(p), ptr(zt), n, V, ptr(out), ptr(key),
                                                  ptr(g_out), ptr(gz), ptr(gv), ptr(gs), ptr(gq), ptr(gt), ptr(gp),
                                                  ptr(gzt), ptr(acc), stream()))
        sh = ctx.shapes
        gpr = None if ctx.same_tp else gp.reshape(sh[3])   # (same tensor: gt already holds the sum)
        if ctx.fixed:   # the kernel reports d / d zoom_to at zoom_to = 1, zoom = zooms_given * zoom_to
            gzg = (gzt / zg).reshape(sh[4]) if ctx.needs_input_grad[6] else None   # (the optimisation loop's zooms are constants)
            return gv, gs.reshape(sh[0]), gq.reshape(sh[1]), gt.reshape(sh[2]), gpr, None, gzg
        return gv, gs.reshape(sh[0]), gq.reshape(sh[1]), gt.reshape(sh[2]), gpr, gzt.reshape(sh[4]), None


class SegmentMeanFn(torch.autograd.Function):
    """Instance-wise average pooling (textural/models/networks.py:310-325): out[n, c, p] = mean of x[., c, .] over the
    pixels that share p's segment id.  `seg` [N, H, W] int32 holds dense ids in [0, K).  Returns (out, means [C, K]).
    d out / d x is the same averaging operator (it is symmetric and idempotent), so backward is one more call."""

    @staticmethod
    def forward(ctx, x, seg, K):
        x = _f32(x, 'x')
        if seg.dtype != torch.int32 or not seg.is_cuda:
            raise TypeError('seg must be an int32 CUDA tensor')
        N, C, H, W = x.shape
        seg = seg.contiguous()
        sums = torch.empty(C, K, dtype=torch.float32, device=x.device)
        counts = torch.empty(K, dtype=torch.float32, device=x.device)
        out = torch.empty_like(x)
        check(lib().sdn_segment_mean(ptr(x), ptr(seg), N, C, H * W, K, ptr(sums), ptr(counts), ptr(out), stream()))
        ctx.save_for_backward(seg)
        ctx.K = K
        means = sums / counts
        ctx.mark_non_differentiable(means)
        return out, means

    @staticmethod
    def backward(ctx, g_out, g_means):
        (seg,) = ctx.saved_tensors
        g = g_out.contiguous()
        N, C, H, W = g.shape
        K = ctx.K
        sums = torch.empty(C, K, dtype=torch.float32, device=g.device)
        counts = torch.empty(K, dtype=torch.float32, device=g.device)
        gx = torch.empty_like(g)
        check(lib().sdn_segment_mean(ptr(g), ptr(seg), N, C, H * W, K, ptr(sums), ptr(counts), ptr(gx), stream()))
        return gx, None, None


class AvgPool3x3S2Fn(torch.autograd.Function):
    """nn.AvgPool2d(3, stride=2, padding=1, count_include_pad=False) on [N, C, H, W] fp32 (any strides; channels-last
    views stay channels-last): textural/models/networks.py:190, 392, 406.  Own kernels because torch's backward of this op
    is wrong on ROCm for channels-last-strided inputs (csrc/fast_pool.hip)."""

    @staticmethod
    def forward(ctx, x):
        import ctypes
        x = want_strided(x, 'input')
        N, C, H, W = x.shape
        OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        inner_c = x.stride(1) == 1 and C > 1
        if inner_c:
            out = torch.empty((N, OH, OW, C), dtype=torch.float32, device=x.device).permute(0, 3, 1, 2)
        else:
            out = torch.empty((N, C, OH, OW), dtype=torch.float32, device=x.device)
        L4 = ctypes.c_long * 4
        check(lib().sdn_avgpool3x3s2_fwd(ptr(x), N, C, H, W, L4(*x.stride()), ptr(out), L4(*out.stride()), int(inner_c),
                                         stream()))
        ctx.shape, ctx.inner_c = (N, C, H, W), inner_c
        return out

    @staticmethod
    def backward(ctx, g):
        import ctypes
        N, C, H, W = ctx.shape
        g = want_strided(g, 'grad')
        if ctx.inner_c:
            gin = torch.empty((N, H, W, C), dtype=torch.float32, device=g.device).permute(0, 3, 1, 2)
        else:
            gin = torch.empty((N, C, H, W), dtype=torch.float32, device=g.device)
        L4 = ctypes.c_long * 4
        check(lib().sdn_avgpool3x3s2_bwd(ptr(g), N, C, H, W, L4(*g.stride()), ptr(gin), L4(*gin.stride()), int(ctx.inner_c),
                                         stream()))
        return gin


def want_strided(t, name):
    """fp32 GPU tensor of any (non-overlapping) strides: no copy is made"""
    if not isinstance(t, torch.Tensor):
        raise TypeError('%s must be a torch.Tensor' % name)
    if not t.is_cuda:
        raise NotImplementedError('%s is on %s; this op only runs on the GPU' % (name, t.device))
    if t.dtype != torch.float32:
        raise TypeError('%s must be float32, got %s' % (name, t.dtype))
    if t.dim() != 4:
        raise ValueError('%s must be [N, C, H, W]' % name)
    return t


def avg_pool_3x3_s2(x):
    return AvgPool3x3S2Fn.apply(x)


def _dense_flat(t):
    """1-D view over the storage of a dense (non-overlapping, gap-free) tensor in memory order, or None."""
    order = sorted(range(t.dim()), key=lambda d: (-t.stride(d), d))
    p = t.permute(order)
    return p.reshape(-1) if p.is_contiguous() else None


def l1_loss_supported(a, b):
    """Both operands fp32 on the GPU, same shape and strides, dense and 16-byte aligned: what L1LossFn takes."""
    return (a.is_cuda and b.is_cuda and a.dtype == torch.float32 and b.dtype == torch.float32 and a.shape == b.shape
            and a.stride() == b.stride() and a.numel() > 0 and a.data_ptr() % 16 == 0 and b.data_ptr() % 16 == 0
            and _dense_flat(a) is not None)


class L1LossFn(torch.autograd.Function):
    """torch.nn.L1Loss()(a, b) = mean |a - b| as one read of both operands, the gradient as one more read and one write
    (csrc/fast_loss.hip).  The operands are walked in MEMORY order: they must share one dense layout (the discriminator
    feature maps are NCHW views of channels-last buffers), see l1_loss_supported."""

    @staticmethod
    def forward(ctx, a, b):
        if not l1_loss_supported(a, b):
            raise ValueError('L1LossFn: operands must be fp32 CUDA tensors of one dense, 16-byte aligned layout')
        fa, fb = _dense_flat(a), _dense_flat(b)
        acc = torch.empty(1, dtype=torch.float64, device=a.device)
        out = torch.empty((), dtype=torch.float32, device=a.device)
        check(lib().sdn_l1_loss_fwd(ptr(fa), ptr(fb), fa.numel(), ptr(acc), ptr(out), stream()))
        ctx.save_for_backward(a, b)
        return out

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        need_a, need_b = ctx.needs_input_grad
        g = g.to(torch.float32).reshape(1).contiguous()
        ga = torch.empty_like(a) if need_a else None   # preserve_format: the operands' own (dense) strides
        gb = torch.empty_like(b) if need_b else None
        if ga is None and gb is None:
            return None, None
        fga = _dense_flat(ga) if ga is not None else None
        fgb = _dense_flat(gb) if gb is not None else None
        if (ga is not None and (fga is None or ga.stride() != a.stride())) or \
                (gb is not None and (fgb is None or gb.stride() != a.stride())):
            raise RuntimeError('L1LossFn: empty_like did not keep the operands\' layout')
        check(lib().sdn_l1_loss_bwd(ptr(_dense_flat(a)), ptr(_dense_flat(b)), a.numel(), ptr(g), ptr(fga), ptr(fgb), stream()))
        return ga, gb
