"""The per-view hot path of test_rnr.py:265-377 as one object: camera poses in, rendered frames out, every stage a
hand-written gfx950 kernel launched on the current HIP stream.

    projection -> face setup -> tiled z-resolve + attribute interpolation      (raster.hip)
    -> TBN / view dir / SH9 / 4-level neural texture / 26 rays -> net input     (shade.hip)
    -> 22 implicit-GEMM MFMA convolutions with fused train-mode BatchNorm      (conv.hip)
    -> bias + tanh + light-transport x env-map ray renderer                     (shade.hip)

Views are independent (BatchNorm statistics are per view), so a batch of N poses is just N frames.

Calling modes:
    render(poses[N])                    one batch on the current stream (bench.py's headline: 8 views per call)
    RNRPipeline(streams=k).render(..)   the batch split into k view groups on k HIP streams (kernel tails overlap)
    RNRPipeline(inflight=d).submit(..)  the reference's mode — ONE view per call, test_rnr.py:265 — with up to d calls in
                                        flight: call i runs on HIP stream i % d with its own G-buffer / activations, so
                                        the rasterizer and shading kernels of view i+1 and the tails of every short
                                        convolution launch run under the U-Net of view i.  submit() returns a FrameHandle;
                                        handle.wait() orders the current stream behind that frame.
"""
import numpy as np
import torch

from . import ops
from .unet import UNetPlan


class FrameHandle:
    """A frame submitted with RNRPipeline.submit: `image` [N,3,S,S] is complete once `event` has passed."""

    def __init__(self, image, event):
        self.image, self.event = image, event

    def wait(self):
        """Order the current HIP stream behind the frame (no host synchronisation) and return the image."""
        torch.cuda.current_stream(self.image.device).wait_event(self.event)
        return self.image

    def synchronize(self):
        self.event.synchronize()
        return self.image


class _Slot:
    """Private state of one call in flight (RNRPipeline(inflight=d)): stream, U-Net activations, G-buffer, frames."""


class RNRPipeline:
    def __init__(self, mesh, img_size, textures, unet_state_dict, pivots_spec, pivots_diff, lp, nf0, num_down=5,
                 sh_start_ch=6, max_views=1, device='cuda:0', near=0.0, far=1e5, global_RT=None, sh_coeff=None, sh_lmax=10,
                 skip_background_tiles=True, streams=1, precision='f32', inflight=1, fuse_ray=False,
                 conv_algo=None):
        """
        mesh: dict v/vt/vn/f_v_idx/f_vt_idx/f_vn_idx (numpy or torch; global_RT applied here if given, as
              network.Rasterizer.__init__ does, network.py:126-128)
        textures: list of [1,S_l,S_l,C] tensors (TextureMapper.textures, network.py:43-57)
        unet_state_dict: RenderingNet.state_dict() of the reference (keys `net.*`)
        pivots_spec / pivots_diff: RaySampler.pivots_dir buffers [3,R]
        lp: environment map [1,Hl,Wl,3] or [Hl,Wl,3] (LightingSH.reconstruct_lp output); may be None when
            sh_coeff [L,(lmax+1)^2,3] is given: then the probe is reconstructed from the coefficients on every call
            (like `lighting_model(lighting_idx, is_lp=True)` inside RayRenderer.forward, network.py:494-495)
        """
        self.dev = torch.device(device)
        # fuse_ray: the ray renderer split in two — ops.ray_weights (everything that does not depend on the U-Net) and the out
        # layer's epilogue (rnr_conv2d_ray: bias + tanh + the sum over the 26 rays, straight from the accumulators) — instead
        # of ray_render_kernel behind a 320 B/px round trip of the out layer's output.  Exact fp32, 80-column out layer only.
        self.fuse_ray = bool(fuse_ray)
        # the ray renderer outputs exactly 0 on background pixels whatever the U-Net produced there (network.py:469-470,
        # 497): the out layer need not compute pixel tiles that contain no foreground pixel.  Frames are bit-identical.
        self.skip_background_tiles = bool(skip_background_tiles)
        # streams > 1: a batch is split into that many view groups rendered on separate HIP streams, so that the tail of
        # one group's kernel overlaps the next kernel of another group (+2 % at 8 views)
        self.n_streams = max(1, int(streams))
        self.S = int(img_size)
        self.near, self.far = float(near), float(far)
        v = torch.as_tensor(mesh['v'], dtype=torch.float32)
        vn = torch.as_tensor(mesh['vn'], dtype=torch.float32)
        if global_RT is not None:
            g = torch.as_tensor(global_RT, dtype=torch.float32)
            v = torch.matmul(g, torch.cat((v, torch.ones(v.shape[0], 1)), 1).t()).t()[:, :3]
            vn = torch.nn.functional.normalize(torch.matmul(g[:3, :3], vn.t()).t(), dim=1)
        self.mesh = ops.DeviceMesh(v, mesh['vt'], vn, mesh['f_v_idx'], mesh['f_vt_idx'], mesh['f_vn_idx'], self.dev)
        self.textures = [torch.as_tensor(t, dtype=torch.float32).reshape(t.shape[-3], t.shape[-2], t.shape[-1])
                         .contiguous().to(self.dev) for t in textures]
        self.C = self.textures[0].shape[-1]
        self.pivots_spec = torch.as_tensor(pivots_spec, dtype=torch.float32).cpu().contiguous()
        self.pivots_diff = torch.as_tensor(pivots_diff, dtype=torch.float32).cpu().contiguous()
        self.n_spec, self.n_diff = self.pivots_spec.shape[1], self.pivots_diff.shape[1]
        self.sh_start_ch = int(sh_start_ch)
        self.c_in = 3 * (self.n_spec + self.n_diff) + 6 + self.C
        self.max_views = int(max_views)
        self.n_streams = min(self.n_streams, self.max_views)
        lane_views = (self.max_views + self.n_streams - 1) // self.n_streams
        self.unet = UNetPlan(unet_state_dict, self.c_in, 3 * (self.n_spec + self.n_diff), nf0, num_down,
                             (self.S, self.S), lane_views if self.n_streams > 1 else self.max_views, self.dev,
                             precision=precision, conv_algo=conv_algo)
        self._lane_unets = [self.unet] + [UNetPlan(unet_state_dict, self.c_in, 3 * (self.n_spec + self.n_diff), nf0, num_down,
                                                   (self.S, self.S), lane_views, self.dev, share_weights_with=self.unet,
                                                   precision=precision, conv_algo=conv_algo)
                                          for _ in range(self.n_streams - 1)]
        self._lane_streams = ops.side_streams(self.dev, self.n_streams) if self.n_streams > 1 else []
        self.sh_lighting, self.sh_coeff = None, None
        if sh_coeff is not None:
            from .lighting import SHLighting
            self.sh_lighting = SHLighting(sh_lmax, self.dev)
            self.sh_coeff = torch.as_tensor(sh_coeff, dtype=torch.float32).to(self.dev)
            self.lp = None
        else:
            self.set_light_probe(lp)
        N, S = self.max_views, self.S
        self._gb = {}
        self._gb_maps = ['face_index_map', 'alpha', 'uv_map', 'normal_map']
        self._net_in = torch.empty(N, S, S, self.unet.in_c_pad, dtype=torch.float32, device=self.dev)
        # two frame buffers: a caller that overlaps the all-gather of step k with the rendering of step k+1 alternates
        self._images = [torch.empty(N, 3, S, S, dtype=torch.float32, device=self.dev) for _ in range(2)]
        self._flip = 0
        ws_views = N if self.n_streams == 1 else (N + self.n_streams - 1) // self.n_streams
        self._lane_ws = [torch.empty(ops._lib.load().rnr_gbuffer_workspace_bytes(ws_views, self.mesh.num_faces, S),
                                     dtype=torch.uint8, device=self.dev) for _ in range(self.n_streams)]
        for m in self._gb_maps:
            dt, tail = ops.GBUFFER_MAPS[m]
            self._gb[m] = torch.empty((N, S, S) + tail, dtype=dt, device=self.dev)
        self.last = {}
        self._v_uvz_override = None
        # calls in flight (submit): slot 0 is the pipeline's own state; the others share the packed weights and own the rest
        self.inflight = max(1, int(inflight))
        if self.inflight > 1 and self.n_streams > 1:
            raise ValueError('streams > 1 (view groups of one batch) and inflight > 1 (several calls in flight) are exclusive')
        self._slots, self._next_slot = [], 0
        slot_streams = ops.side_streams(self.dev, self.inflight) if self.inflight > 1 else []
        for i in range(self.inflight if self.inflight > 1 else 0):
            sl = _Slot()
            sl.stream = slot_streams[i]
            sl.unet = self.unet if i == 0 else UNetPlan(unet_state_dict, self.c_in, 3 * (self.n_spec + self.n_diff), nf0,
                                                        num_down, (S, S), N, self.dev, share_weights_with=self.unet,
                                                        precision=precision, conv_algo=conv_algo)
            sl.ws = self._lane_ws[0] if i == 0 else torch.empty_like(self._lane_ws[0])
            sl.gb = self._gb if i == 0 else {m: torch.empty_like(t) for m, t in self._gb.items()}
            sl.net_in = self._net_in if i == 0 else torch.empty_like(self._net_in)
            sl.images = [torch.empty(N, 3, S, S, dtype=torch.float32, device=self.dev) for _ in range(2)]
            sl.flip = 0
            self._slots.append(sl)

    def _prep(self, slot):
        """Outputs of ops.frame_prepare private to the pipeline's own call state or to a slot in flight: projected vertices
        [max_views, nv, 3], per-face tangents [nf, 3] and (with SH lighting) the reconstructed light probe [h, w, 3]."""
        owner = self if slot is None else slot
        b = getattr(owner, '_prep_buf', None)
        if b is None:
            b = {'v_uvz': torch.empty(self.max_views, self.mesh.num_vertices, 3, dtype=torch.float32, device=self.dev),
                 'tangents': torch.empty(self.mesh.num_faces, 3, dtype=torch.float32, device=self.dev),
                 'lp': None if self.sh_lighting is None else
                 torch.empty(self.sh_lighting.h, self.sh_lighting.w, 3, dtype=torch.float32, device=self.dev)}
            owner._prep_buf = b
        return b

    def _ray_w(self, slot, lane):
        """[max_views,S,S,c_out_pad] ray-weight buffer of the pipeline (shared by the view-group lanes, which use disjoint
        view ranges) or of a slot."""
        owner = self if slot is None else slot
        buf = getattr(owner, '_ray_w_buf', None)
        if buf is None:
            buf = torch.empty(self.max_views, self.S, self.S, self.unet.out.c_pad, dtype=torch.float32, device=self.dev)
            owner._ray_w_buf = buf
        return buf

    def set_light_probe(self, lp):
        lp = torch.as_tensor(lp, dtype=torch.float32)
        self.lp = lp.reshape(lp.shape[-3], lp.shape[-2], 3).contiguous().to(self.dev)

    def render(self, proj, pose, proj_inv, R_inv, keep_intermediates=False, lighting_idx=0, stage_events=None, v_uvz=None):
        """proj/proj_inv/R_inv [N,3,3], pose [N,4,4] device float32 -> image [N,3,S,S].  The result is a view into one of
        two internal buffers used alternately: it stays valid until the call after next.
        stage_events: optional list; (name, torch.cuda.Event) pairs are appended at the stage boundaries
        (measurement only — bench.py's per-stage HBM figures).
        v_uvz: optional [N,nv,3] projected NDC vertices (u, v, z_cam) to rasterize INSTEAD of projecting `proj` / `pose`
        here (parity tests feed the reference's own projection: the integer maps then do not depend on how this device
        rounds a 3x3 matmul); single-stream calls only."""
        with ops.on_device(self.dev):
            self._v_uvz_override = v_uvz
            try:
                return self._render(proj, pose, proj_inv, R_inv, keep_intermediates, lighting_idx, stage_events)
            finally:
                self._v_uvz_override = None

    def submit(self, proj, pose, proj_inv, R_inv, lighting_idx=0):
        """One call of the reference's per-view loop (test_rnr.py:265-377), asynchronous: the poses [N <= max_views] are
        rendered on the next of the `inflight` private HIP streams (after everything already queued on the current stream,
        which produced the poses) and a FrameHandle is returned at once.  The image stays valid until 2 * inflight further
        submits.  With inflight == 1 this is render() plus an event."""
        with ops.on_device(self.dev):
            if not self._slots:
                image = self._render(proj, pose, proj_inv, R_inv, False, lighting_idx, None)
                ev = torch.cuda.Event()
                ev.record()
                return FrameHandle(image, ev)
            N = proj.shape[0]
            if N > self.max_views:
                raise RuntimeError('pipeline built for max_views=%d, got %d poses' % (self.max_views, N))
            sl = self._slots[self._next_slot]
            self._next_slot = (self._next_slot + 1) % len(self._slots)
            sl.stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(sl.stream):
                # per-call work of the reference inside the slot's stream too (tangents, SH light probe, projection): one
                # launch (ops.frame_prepare inside _render_group), outputs private to the slot
                lp = self.lp if self.sh_lighting is None else ('sh', lighting_idx)
                image = sl.images[sl.flip][:N]
                sl.flip ^= 1
                args = [t.contiguous() for t in (proj, pose, proj_inv, R_inv)]
                for t in args:
                    # the caller's tensors are read by kernels of THIS stream: tell the caching allocator, or a caller that
                    # drops its pose tensors right after submit() gets their memory recycled under the running kernels
                    t.record_stream(sl.stream)
                self._render_group(0, 0, N, *args, lp, image, lambda name: None, slot=sl, fused=True)
                ev = torch.cuda.Event()
                ev.record(sl.stream)
            return FrameHandle(image, ev)

    def _render(self, proj, pose, proj_inv, R_inv, keep_intermediates, lighting_idx, stage_events):

        def mark(name):
            if stage_events is not None:
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                stage_events.append((name, e))
        mark('start')
        N = proj.shape[0]
        if N > self.max_views:
            raise RuntimeError('pipeline built for max_views=%d, got %d poses' % (self.max_views, N))
        proj, pose, proj_inv, R_inv = proj.contiguous(), pose.contiguous(), proj_inv.contiguous(), R_inv.contiguous()
        image = self._images[self._flip][:N]
        self._flip ^= 1
        lanes = 1 if (stage_events is not None or keep_intermediates or self._v_uvz_override is not None) else min(self.n_streams, N)
        fused = lanes == 1 and self._v_uvz_override is None
        if fused:
            # per-face tangents (recomputed per call, as get_TBN_map does, render.py:135-150), the light probe and the vertex
            # projection share one launch with the rasterizer's workspace clearing: ops.frame_prepare in _render_group
            lp = self.lp if self.sh_lighting is None else ('sh', lighting_idx)
        else:
            self.mesh._tangents = None
            self.mesh.tangents()
            lp = self.lp if self.sh_lighting is None else self.sh_lighting.light_probe(self.sh_coeff[lighting_idx])
        if lanes == 1:
            if N > self._lane_unets[0].N:
                raise RuntimeError('pipeline built with streams=%d: a single-stream call takes at most %d poses'
                                   % (self.n_streams, self._lane_unets[0].N))
            inter = self._render_group(0, 0, N, proj, pose, proj_inv, R_inv, lp, image, mark, fused=fused)
            if keep_intermediates:
                self.last = inter
            return image
        # view groups on separate streams; shared per-call work (tangents, light probe) was issued on the caller's stream
        cur = torch.cuda.current_stream()
        per = (N + lanes - 1) // lanes
        for i in range(lanes):
            lo, hi = i * per, min(N, (i + 1) * per)
            if lo >= hi:
                break
            st = self._lane_streams[i]
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                self._render_group(i, lo, hi, proj, pose, proj_inv, R_inv, lp, image, lambda name: None)
        for st in self._lane_streams[:lanes]:
            cur.wait_stream(st)
        return image

    def _render_group(self, lane, lo, hi, proj, pose, proj_inv, R_inv, lp, image, mark, slot=None, fused=False):
        """Views [lo, hi) of the batch on the current stream, with lane-private scratch and U-Net activations (or, for a
        submitted call, everything private to its slot)."""
        n = hi - lo
        unet = self._lane_unets[lane] if slot is None else slot.unet
        gbufs = self._gb if slot is None else slot.gb
        net_in = self._net_in if slot is None else slot.net_in
        ov = getattr(self, '_v_uvz_override', None) if slot is None else None
        gb = {m: gbufs[m][lo:hi] for m in self._gb_maps}
        ws = self._lane_ws[lane] if slot is None else slot.ws
        fused_prepare = fused and ov is None
        if fused_prepare:
            pb = self._prep(slot)
            sh_lp = isinstance(lp, tuple)
            ops.frame_prepare(self.mesh, proj[lo:hi], pose[lo:hi], self.S, v_uvz=pb['v_uvz'][:n], tangents=pb['tangents'],
                              lp_basis=self.sh_lighting.basis_recon if sh_lp else None,
                              lp_coeff=self.sh_coeff[lp[1]] if sh_lp else None, light_probe=pb['lp'] if sh_lp else None,
                              workspace=ws)
            v_uvz, tangents = pb['v_uvz'][:n], pb['tangents']
            if sh_lp:
                lp = pb['lp']
            ops.rasterize_gbuffer(self.mesh, v_uvz, None, self.S, self.near, self.far, maps=self._gb_maps, out=gb, workspace=ws,
                                  prepared=True)
        else:
            if ov is None:
                v_uvz = ops.project_vertices(self.mesh.v, proj[lo:hi], pose[lo:hi, :3, :3].contiguous(),
                                             pose[lo:hi, :3, 3].contiguous(), self.S)
            else:
                v_uvz = ov[lo:hi].contiguous()
            tangents = None             # mesh.tangents(): computed by the caller for this call
            ops.rasterize_gbuffer(self.mesh, v_uvz, None, self.S, self.near, self.far, maps=self._gb_maps, out=gb, workspace=ws)
        mark('raster')
        sh = ops.shade_inputs(gb, self.mesh, proj_inv[lo:hi], R_inv[lo:hi], self.textures, self.pivots_spec,
                              self.pivots_diff, self.sh_start_ch, c_pad=unet.in_c_pad, net_in=net_in[lo:hi], tangents=tangents)
        mark('shade_inputs')
        if self.fuse_ray:
            ray_w = ops.ray_weights(sh['net_in'], gb['alpha'], lp, self.n_spec, self.n_diff, unet.out.c_pad, albedo_diff_ch=0,
                                    albedo_spec_ch=3, out=self._ray_w(slot, lane)[lo:hi] if slot is None else self._ray_w(slot, lane)[:n])
            mark('ray_weights')
            unet.forward(sh['net_in'], n, gb['alpha'] if self.skip_background_tiles else None, ray=(ray_w, image[lo:hi]))
            mark('unet')
            raw = None
        else:
            raw = unet.forward(sh['net_in'], n, gb['alpha'] if self.skip_background_tiles else None)
            mark('unet')
            ops.ray_render(raw, unet.out_bias, sh['net_in'], gb['alpha'], lp, self.n_spec, self.n_diff, albedo_diff_ch=0,
                           albedo_spec_ch=3, image=image[lo:hi])
            mark('ray_render')
        return {'v_uvz': v_uvz, 'gb': gb, 'net_in': sh['net_in'], 'unet_raw': raw}
