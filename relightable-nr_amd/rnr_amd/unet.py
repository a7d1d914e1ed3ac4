"""RenderingNet's U-Net (network.py:219-253, pytorch_prototyping.py:370-536) as a static plan of
rnr_conv2d / rnr_bn_finalize launches on channel-last HBM tensors.

Built from a reference state-dict (strict key names).  Only the LIVE path is planned:
`UnetSkipConnectionBlock.forward` recomputes `y` under `if self.flag_outer:` after the `if self.gcn:` branch
(pytorch_prototyping.py:407-419), so `fuse.*` weights and `v_fea` never influence the output; they are accepted
and ignored.  BatchNorm uses per-view batch statistics (the reference forces BN into train mode at inference,
test_rnr.py:229-233, with N = 1 per call); Dropout2d is the identity in eval mode.
"""
import ctypes
import os

import torch

from . import _lib
from ._lib import (ACT_LRELU02, ACT_NONE, ACT_RELU, CONV3x3_REFLECT, CONV4x4S2_REFLECT, CONVT4x4S2, RnrConvBn, RnrConvDesc,
                   RnrConvSrc, check)
from .ops import _ptr, _stream, on_device


def _pad16(c):
    return (int(c) + 15) // 16 * 16


class _Act:
    """A raw conv output living in HBM plus the affine+activation its consumers must apply."""

    def __init__(self, c, c_pad, h, w, act):
        self.c, self.c_pad, self.h, self.w, self.act = c, c_pad, h, w, act
        self.data = None      # [N,h,w,c_pad]
        self.scale = None     # [N,c_pad] or None
        self.shift = None


DEFAULT_CONV_ALGO = 'winograd4'


class UNetPlan:
    def __init__(self, state_dict, in_channels, out_channels, nf0, num_down, img_hw, max_views, device,
                 prefix='net.', in_c_pad=None, bn_mode='batch', share_weights_with=None, precision='f32',
                 update_running_stats=False, check_finite=None, conv_algo=None):
        """bn_mode 'batch': BatchNorm2d in train mode with PER-VIEW batch statistics — what test_rnr.py:229-233 forces,
        evaluated the way the reference evaluates it (one view per call); a batch of N poses is N independent frames.
        'batch_all': train-mode BatchNorm2d exactly as torch computes it for ONE call with an [N,C,H,W] input: statistics
        over the whole batch (N,H,W), identical for every view (what the drop-in `Unet.forward` must return for N > 1);
        with update_running_stats the running_mean / running_var tensors of the state-dict are updated in place
        (momentum 0.1, unbiased variance) like torch does.
        'running': eval-mode BatchNorm from the running_mean / running_var buffers of the state-dict.
        precision 'f32': exact fp32 MFMA (v_mfma_f32_32x32x2_f32).  'bf16x6' / 'f16x3': fp32 emulated on the 16-bit matrix
        cores — operands split into three bf16 terms (exactly; six partial products) or two fp16 terms (22 significand
        bits; three partial products), accumulated in fp32; error of the order of fp32's own rounding, 2.7x / 5.3x fewer
        MFMA cycles (include/rnr_hip.h, RNR_CONV_F32_EMU_BF16X6 / RNR_CONV_F32_EMU_F16X3).
        conv_algo 'winograd': the 3x3 convolutions run as Winograd F(2x2, 3x3) and the 4x4 stride-2 ones (both directions) as
        F(2x2, 2x2) — fp32 operands and accumulation on the same matrix-core instruction, 2.25x / 1.78x fewer multiplications
        (include/rnr_hip.h, RNR_CONV_WINOGRAD; 'f32' only) — where the layer shape allows; 'winograd4' (the DEFAULT since r04): additionally
        F(4x4, 3x3) — 4x fewer multiplications than the direct form, ~3.5x the rounding error of F(2x2, 3x3) (~4.5x the direct
        form's rms; every layer shape <= 1e-4 of the output peak vs float64, all 720 spiral frames <= 1.6e-6 from the direct path) — for the 3x3
        layers whose grid fills the chip (RNR_CONV_WINOGRAD4); 'direct': every convolution as a direct implicit GEMM.
        None: $RNR_CONV_ALGO, else DEFAULT_CONV_ALGO.
        share_weights_with: another UNetPlan of the same network whose packed weights / BN parameters are reused
        (activations, statistics and scratch stay private) — one plan per HIP stream of RNRPipeline."""
        if precision not in _lib.EMU_FLAGS:
            raise ValueError("precision must be one of %s" % sorted(_lib.EMU_FLAGS))
        if bn_mode not in ('batch', 'batch_all', 'running'):
            raise ValueError("bn_mode must be 'batch', 'batch_all' or 'running'")
        if share_weights_with is not None:
            # the donor's packed buffers are used as they are: their layout (direct image only / + Winograd image / 16-bit
            # term image) is fixed by the donor's conv_algo and precision, so both are inherited and a contradicting
            # explicit choice is an error, not a silent read past the end of the donor's buffers
            want_algo = conv_algo if precision == 'f32' else 'direct'
            if want_algo is not None and want_algo != share_weights_with.conv_algo:
                raise ValueError("share_weights_with: conv_algo %r differs from the donor plan's %r" % (conv_algo, share_weights_with.conv_algo))
            if precision != share_weights_with.precision:
                raise ValueError("share_weights_with: precision %r differs from the donor plan's %r" % (precision, share_weights_with.precision))
            conv_algo = share_weights_with.conv_algo
        if conv_algo is None:
            conv_algo = os.environ.get('RNR_CONV_ALGO') or DEFAULT_CONV_ALGO
        if conv_algo not in ('direct', 'winograd', 'winograd4'):
            raise ValueError("conv_algo must be 'direct', 'winograd' or 'winograd4'")
        if precision != 'f32':
            conv_algo = 'direct'        # the emulation kernels have no Winograd form
        self.conv_algo = conv_algo
        self.bn_mode = bn_mode
        self.precision = precision
        # f16x3 splits activations into fp16 terms: |act(scale * x + shift)| must stay below 65504, which BatchNorm outputs do
        # but the four un-normalised convolutions of the innermost block need not for exotic weights.  check_finite=True makes
        # every forward verify (one host synchronisation) that its result holds no inf / NaN — a debugging aid, off by default.
        # Winograd spreads an inf / NaN activation over the whole 2 x 2 tiles whose patch contains it (a direct convolution
        # confines it to the windows that contain it; include/rnr_hip.h, RNR_CONV_WINOGRAD).  check_finite=None (the default)
        # therefore checks the FIRST forward of a Winograd plan — input and output, one host synchronisation, once — and warns;
        # True checks every forward and raises; False never checks.
        self.check_finite = 'first' if (check_finite is None and conv_algo != 'direct') else bool(check_finite)
        self.L = _lib.load()
        self.dev = device
        self.N = int(max_views)
        self.H, self.W = int(img_hw[0]), int(img_hw[1])
        self.num_down = int(num_down)
        self.in_channels, self.out_channels = int(in_channels), int(out_channels)
        self.in_c_pad = _pad16(in_channels) if in_c_pad is None else int(in_c_pad)
        sd = {k: v for k, v in state_dict.items()}
        g = lambda k: sd[prefix + k].detach().to(device=device, dtype=torch.float32).contiguous()
        has = lambda k: (prefix + k) in sd
        self.steps = []
        self.ws_bytes = 256
        # One launch per convolution (rnr_conv2d_fused): the BatchNorm behind it is finalised by the last workgroup of each
        # view and shallow split-K slices meet inside the launch.  'batch_all' (whole-batch statistics, running buffers)
        # keeps the separate rnr_bn_finalize_batch launch; RNR_UNET_UNFUSED=1 runs the separate launches everywhere (A/B).
        self.fused = bn_mode != 'batch_all' and os.environ.get('RNR_UNET_UNFUSED') != '1'
        # ... except that a ONE-view call of a 'batch_all' plan is the same arithmetic (whole batch = the view): it takes the fused
        # launches too, the running statistics updated by the launch that finalises the layer (rnr_conv_bn.running_mean / _var;
        # r05: the drop-in RenderingNet of the reference's one-view loop, 2.65 -> 2.3 ms)
        self.fused_single = bn_mode == 'batch_all' and os.environ.get('RNR_UNET_UNFUSED') != '1'
        self._tile_mask = None
        self._keep = []

        def conv_step(kind, srcs, wkey, bn_key, bias_key, act, c_out):
            """srcs: list of _Act (1 or 2).  Returns the produced _Act."""
            s0 = srcs[0]
            s1 = srcs[1] if len(srcs) > 1 else None
            desc = RnrConvDesc(kind, s0.c, s0.c_pad, s1.c if s1 else 0, s1.c_pad if s1 else 0, c_out, _pad16(c_out))
            desc.flags |= _lib.EMU_FLAGS[precision]
            if conv_algo != 'direct':
                desc.flags |= _lib.CONV_WINOGRAD
            if conv_algo == 'winograd4' and kind == CONV3x3_REFLECT and desc.c_out_pad % 64 == 0:
                desc.flags |= _lib.CONV_WINOGRAD4
            if share_weights_with is not None:
                packed = share_weights_with.steps[len(self.steps)]['packed']
                if packed.numel() != self.L.rnr_packed_weight_floats(ctypes.byref(desc)):
                    raise ValueError('share_weights_with: layer %d of the donor plan was packed for another descriptor' % len(self.steps))
            else:
                w = g(wkey)
                packed = torch.empty(self.L.rnr_packed_weight_floats(ctypes.byref(desc)), dtype=torch.float32, device=device)
                with on_device(device):
                    check(self.L.rnr_pack_conv_weight(ctypes.byref(desc), _ptr(w), _ptr(packed), _stream()))
            if kind == CONV3x3_REFLECT:
                oh, ow = s0.h, s0.w
            elif kind == CONV4x4S2_REFLECT:
                oh, ow = s0.h // 2, s0.w // 2
            else:
                oh, ow = s0.h * 2, s0.w * 2
            out = _Act(c_out, desc.c_out_pad, oh, ow, act)
            out.data = torch.empty(self.N, oh, ow, desc.c_out_pad, dtype=torch.float32, device=device)
            step = {'desc': desc, 'packed': packed, 'srcs': srcs, 'out': out, 'in_hw': (s0.h, s0.w), 'bn': None, 'sync': None,
                    'cbn': None}
            if self.fused or self.fused_single:       # arrival counters + statistics shards: zero now, left at zero by every call
                step['sync'] = torch.zeros(self.L.rnr_conv_sync_bytes(ctypes.byref(desc), self.N, s0.h, s0.w),
                                           dtype=torch.uint8, device=device)
            if bn_key is not None and has(bn_key + '.weight') and bn_mode == 'running':
                gamma, beta = g(bn_key + '.weight'), g(bn_key + '.bias')
                sc = gamma / torch.sqrt(g(bn_key + '.running_var') + 1e-5)
                sh = beta - g(bn_key + '.running_mean') * sc
                pad = lambda t: torch.cat([t, torch.zeros(desc.c_out_pad - c_out, device=device)])[None].repeat(self.N, 1)
                out.scale, out.shift = pad(sc).contiguous(), pad(sh).contiguous()
            elif bn_key is not None and has(bn_key + '.weight'):
                gamma, beta = g(bn_key + '.weight'), g(bn_key + '.bias')
                out.scale = torch.empty(self.N, desc.c_out_pad, dtype=torch.float32, device=device)
                out.shift = torch.empty(self.N, desc.c_out_pad, dtype=torch.float32, device=device)
                # statistics start at zero and every rnr_bn_finalize_reset leaves them at zero: no memset per layer
                desc.flags |= _lib.CONV_STATS_PREZEROED
                step['bn'] = {'gamma': gamma, 'beta': beta, 'running_mean': None, 'running_var': None,
                              'stats': None if self.fused else
                              torch.zeros(self.N, desc.c_out_pad, 2, dtype=torch.float64, device=device)}
                if self.fused:
                    step['cbn'] = RnrConvBn(gamma.data_ptr(), beta.data_ptr(), out.scale.data_ptr(), out.shift.data_ptr(), 1e-5)
                if bn_mode == 'batch_all' and update_running_stats and has(bn_key + '.running_mean'):
                    rm, rv = sd[prefix + bn_key + '.running_mean'], sd[prefix + bn_key + '.running_var']
                    if not (rm.is_cuda and rm.dtype == torch.float32 and rm.is_contiguous() and
                            rv.is_cuda and rv.dtype == torch.float32 and rv.is_contiguous()):
                        raise RuntimeError('update_running_stats needs float32 device-resident running buffers')
                    step['bn']['running_mean'], step['bn']['running_var'] = rm, rv      # updated IN PLACE
                if self.fused_single:
                    rm, rv = step['bn']['running_mean'], step['bn']['running_var']
                    step['cbn'] = RnrConvBn(gamma.data_ptr(), beta.data_ptr(), out.scale.data_ptr(), out.shift.data_ptr(), 1e-5,
                                            rm.data_ptr() if rm is not None else None, rv.data_ptr() if rv is not None else None, 0.1)
            elif bias_key is not None and has(bias_key):
                b = torch.zeros(desc.c_out_pad, dtype=torch.float32, device=device)
                b[:c_out] = g(bias_key)
                out.shift = b[None].repeat(self.N, 1).contiguous()
                step['bias'] = b
            # split-K depends on the number of views actually passed at call time: size the scratch for every n <= N
            for n_views in range(1, self.N + 1):
                self.ws_bytes = max(self.ws_bytes, self.L.rnr_conv_workspace_bytes(ctypes.byref(desc), n_views, s0.h, s0.w))
            self.steps.append(step)
            return out

        # network input (already channel-last, no affine / activation)
        x = _Act(self.in_channels, self.in_c_pad, self.H, self.W, ACT_NONE)
        self.input = x
        h = conv_step(CONV3x3_REFLECT, [x], 'in_layer.0.net.1.weight', 'in_layer.1', None, ACT_LRELU02, nf0)
        max_c = 8 * nf0

        def block(y, path, depth):
            inner = min(2 ** (depth + 1) * nf0, max_c)
            outer = y.c
            d, u = path + 'down.net.', path + 'up.net.'
            if depth == self.num_down - 1:      # innermost: norm=None => convs carry a bias (pytorch_prototyping.py:484-489)
                inner = outer                   # both ends are min(2^(num_down-1) nf0, max) there
                t = conv_step(CONV3x3_REFLECT, [y], d + '1.weight', None, d + '1.bias', ACT_LRELU02, outer)
                t = conv_step(CONV4x4S2_REFLECT, [t], d + '5.weight', None, d + '5.bias', ACT_LRELU02, inner)
                t = conv_step(CONVT4x4S2, [t], u + '0.weight', None, u + '0.bias', ACT_RELU, outer)
                t = conv_step(CONV3x3_REFLECT, [t], u + '3.net.1.weight', None, u + '3.net.1.bias', ACT_RELU, outer)
            else:
                t = conv_step(CONV3x3_REFLECT, [y], d + '1.weight', d + '2', None, ACT_LRELU02, outer)
                t = conv_step(CONV4x4S2_REFLECT, [t], d + '6.weight', d + '7', None, ACT_LRELU02, inner)
                cat = block(t, path + 'submodule.', depth + 1)
                t = conv_step(CONVT4x4S2, cat, u + '0.weight', u + '1', None, ACT_RELU, outer)
                t = conv_step(CONV3x3_REFLECT, [t], u + '4.net.1.weight', u + '5', None, ACT_RELU, outer)
            return [y, t]                       # torch.cat([x, y], 1) (pytorch_prototyping.py:429)

        cat = block(h, 'unet_block.', 0)
        out = conv_step(CONV3x3_REFLECT, cat, 'out_layer.0.net.1.weight', None, None, ACT_NONE, self.out_channels)
        self.out = out
        self.out_bias = torch.zeros(out.c_pad, dtype=torch.float32, device=device)
        if has('out_layer.0.net.1.bias'):
            self.out_bias[:self.out_channels] = g('out_layer.0.net.1.bias')
        self.workspace = torch.empty(self.ws_bytes, dtype=torch.uint8, device=device)
        self.flops_per_view = self._count_flops()

    def mfma_flops_per_view(self, n_views, masked_out_layer=True):
        """Multiply-add FLOPs the matrix cores execute per view when `n_views` are passed per call: the direct-form count of
        every convolution (flops_per_view) divided by 2.25 / 1.78 where rnr_conv_algorithm says a Winograd kernel runs (the
        out layer runs direct when it is tile-masked)."""
        total = 0.0
        for i, s in enumerate(self.steps):
            d, (h, w) = s['desc'], s['in_hw']
            algo = self.L.rnr_conv_algorithm(ctypes.byref(d), int(n_views), h, w)
            if masked_out_layer and i == len(self.steps) - 1 and algo != 3:
                algo = 0
            total += self._layer_flops(d, h, w) / {0: 1.0, 1: 36.0 / 16.0, 2: 16.0 / 9.0, 3: 36.0 / 16.0, 4: 4.0}[algo]
        return total

    @staticmethod
    def _layer_flops(d, h, w):
        cin = d.c_in0 + d.c_in1
        if d.kind == CONV3x3_REFLECT:
            return 2 * h * w * 9 * cin * d.c_out
        if d.kind == CONV4x4S2_REFLECT:
            return 2 * (h // 2) * (w // 2) * 16 * cin * d.c_out
        return 2 * (2 * h) * (2 * w) * 4 * cin * d.c_out

    def _count_flops(self):
        total = 0
        for s in self.steps:
            d, (h, w) = s['desc'], s['in_hw']
            cin = d.c_in0 + d.c_in1
            if d.kind == CONV3x3_REFLECT:
                total += 2 * h * w * 9 * cin * d.c_out
            elif d.kind == CONV4x4S2_REFLECT:
                total += 2 * (h // 2) * (w // 2) * 16 * cin * d.c_out
            else:
                total += 2 * (2 * h) * (2 * w) * 4 * cin * d.c_out
        return total

    def _src(self, a, n):
        return RnrConvSrc(a.data.data_ptr(), a.scale.data_ptr() if a.scale is not None else None,
                          a.shift.data_ptr() if a.shift is not None else None, a.c_pad, a.act)

    def forward(self, net_in, n_views=None, consumer_alpha=None, ray=None):
        """net_in [n,H,W,in_c_pad] channel-last -> raw out-layer output [n,H,W,out_c_pad] (bias/tanh NOT applied).
        consumer_alpha [n,H,W]: promise that the caller reads the result only where alpha > 0 (the ray renderer zeroes
        background pixels); the out layer then skips pixel tiles without any such pixel and leaves them unwritten.
        ray = (ray_w [n,H,W,out_c_pad], image [n,3,H,W]): the out layer's epilogue applies bias + tanh and the ray weights
        (ops.ray_weights) and writes the FRAME into `image` (rnr_conv2d_ray); the raw output is then not produced and the
        call returns `image`.  Raises if the out layer is not on the plan that has this epilogue (`supports_ray`)."""
        with on_device(self.dev):
            return self._forward(net_in, n_views, consumer_alpha, ray)

    def _forward(self, net_in, n_views, consumer_alpha, ray=None):
        n = net_in.shape[0] if n_views is None else n_views
        if net_in.device != torch.device(self.dev) and not (net_in.is_cuda and torch.device(self.dev).index is None):
            raise RuntimeError('net_in lives on %s, the plan on %s' % (net_in.device, self.dev))
        if n > self.N:
            raise RuntimeError('UNetPlan built for at most %d views, got %d' % (self.N, n))
        if net_in.shape[-1] != self.in_c_pad or net_in.shape[1] != self.H or net_in.shape[2] != self.W:
            raise RuntimeError('net_in shape %s does not match plan (H=%d W=%d c_pad=%d)' %
                               (tuple(net_in.shape), self.H, self.W, self.in_c_pad))
        self.input.data = net_in
        L, st = self.L, _stream()
        last = self.steps[-1]
        mask = None
        # the ray-renderer epilogue lives in the direct 80-column kernel: that call (and its tile mask) use the out layer's
        # descriptor without the Winograd flag (same packed buffer: the direct image comes first)
        out_desc = last['desc']
        if ray is not None and (out_desc.flags & _lib.CONV_WINOGRAD):
            out_desc = RnrConvDesc(out_desc.kind, out_desc.c_in0, out_desc.c_in0_pad, out_desc.c_in1, out_desc.c_in1_pad,
                                   out_desc.c_out, out_desc.c_out_pad, out_desc.flags & ~(_lib.CONV_WINOGRAD | _lib.CONV_WINOGRAD4))
        self._out_desc = out_desc
        if consumer_alpha is not None:
            h, w = last['in_hw']
            tiles = L.rnr_conv_tile_count(ctypes.byref(out_desc), n, h, w)
            if tiles and last['bn'] is None:
                if self._tile_mask is None or self._tile_mask.numel() < tiles:
                    self._tile_mask = torch.empty(tiles, dtype=torch.uint8, device=self.dev)
                mask = self._tile_mask
                check(L.rnr_conv_active_tiles(ctypes.byref(out_desc), _ptr(consumer_alpha), _ptr(mask), n, h, w, st))
        try:
            self._run_steps(n, mask, L, st, ray)
        except Exception:
            for s in self.steps:            # a failed launch may leave statistics half-accumulated: restore the invariant
                if s['bn'] and s['bn']['stats'] is not None:
                    s['bn']['stats'].zero_()
                if s['sync'] is not None:
                    s['sync'].zero_()
            raise
        res = self.out.data[:n] if ray is None else ray[1]
        if self.check_finite == 'first' and not torch.cuda.is_current_stream_capturing():
            self.check_finite = False
            if not (bool(torch.isfinite(net_in[:n]).all()) and (mask is not None or bool(torch.isfinite(res).all()))):
                import warnings
                warnings.warn("UNetPlan(conv_algo=%r): non-finite values in the network input or output of the first call; the "
                              "Winograd kernels spread an inf / NaN activation over every output tile whose input patch contains it "
                              "(2 x 2 outputs of a 4 x 4 patch for F(2x2, .), 4 x 4 outputs of a 6 x 6 patch for F(4x4, 3x3) — the "
                              "default 'winograd4') where a direct convolution confines it to the windows that contain it "
                              "(include/rnr_hip.h, RNR_CONV_WINOGRAD / RNR_CONV_WINOGRAD4) — use conv_algo='direct' to localise it"
                              % self.conv_algo, RuntimeWarning, stacklevel=2)
        elif self.check_finite is True and mask is None and not bool(torch.isfinite(res).all()):
            raise FloatingPointError('UNetPlan(precision=%r): non-finite values in the network output (f16x3 needs activations '
                                     'below 65504: include/rnr_hip.h, RNR_CONV_F32_EMU_F16X3)' % self.precision)
        return res

    def _run_steps(self, n, mask, L, st, ray=None):
        last = self.steps[-1]
        for s in self.steps:
            srcs = s['srcs']
            s0 = self._src(srcs[0], n)
            s1 = self._src(srcs[1], n) if len(srcs) > 1 else None
            out, bn = s['out'], s['bn']
            h, w = s['in_hw']
            if s is last and ray is not None:
                ray_w, image = ray
                check(L.rnr_conv2d_ray(ctypes.byref(self._out_desc), ctypes.byref(s0), ctypes.byref(s1) if s1 else None,
                                       _ptr(s['packed']), _ptr(ray_w), _ptr(self.out_bias), _ptr(image), n, h, w,
                                       _ptr(mask), st))
                continue
            if self.fused or (self.fused_single and n == 1):
                check(L.rnr_conv2d_fused(ctypes.byref(s['desc']), ctypes.byref(s0), ctypes.byref(s1) if s1 else None,
                                         _ptr(s['packed']), _ptr(out.data), ctypes.byref(s['cbn']) if s['cbn'] else None, n, h, w,
                                         _ptr(self.workspace), self.ws_bytes, _ptr(s['sync']), s['sync'].numel(),
                                         _ptr(mask) if s is last else None, st))
                continue
            check(L.rnr_conv2d_masked(ctypes.byref(s['desc']), ctypes.byref(s0), ctypes.byref(s1) if s1 else None,
                                      _ptr(s['packed']), _ptr(out.data), _ptr(bn['stats']) if bn else None, n, h, w,
                                      _ptr(self.workspace), self.ws_bytes, _ptr(mask) if s is last else None, st))
            if bn and self.bn_mode == 'batch_all':
                check(L.rnr_bn_finalize_batch(_ptr(bn['stats']), _ptr(bn['gamma']), _ptr(bn['beta']), _ptr(out.scale),
                                              _ptr(out.shift), _ptr(bn['running_mean']), _ptr(bn['running_var']), 0.1, n,
                                              out.c, out.c_pad, float(out.h * out.w), 1e-5, st))
            elif bn:
                check(L.rnr_bn_finalize_reset(_ptr(bn['stats']), _ptr(bn['gamma']), _ptr(bn['beta']), _ptr(out.scale),
                                        _ptr(out.shift), n, out.c, out.c_pad, float(out.h * out.w), 1e-5, st))
        return self.out.data[:n]
