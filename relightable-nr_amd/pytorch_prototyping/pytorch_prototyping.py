"""Drop-in for the 2-D U-Net pieces of pytorch_prototyping/pytorch_prototyping.py (96-121, 124-277, 370-536).

These classes are PARAMETER CONTAINERS with the reference's module tree, so `state_dict()` has exactly the
reference's keys (incl. the aliases `in_layer.0.weight`, `up.net.4.weight`, `out_layer_weight`)
and reference checkpoints load with strict=True.  Their compute is not torch: `Unet.forward` hands the weights to
`rnr_amd.unet.UNetPlan`, i.e. the MFMA implicit-GEMM kernels of librnr_hip.so.
"""
import torch
import torch.nn as nn

from rnr_amd import ops
from rnr_amd.unet import UNetPlan


class Conv2dSame(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, bias=True, padding_layer=nn.ReflectionPad2d):
        super().__init__()
        ka = kernel_size // 2
        kb = ka - 1 if kernel_size % 2 == 0 else ka
        self.net = nn.Sequential(padding_layer((ka, kb, ka, kb)),
                                 nn.Conv2d(in_channels, out_channels, kernel_size, bias=bias, stride=1))
        self.weight = self.net[1].weight
        self.bias = self.net[1].bias


class UpBlock(nn.Module):
    def __init__(self, in_channels, out_channels, post_conv=True, use_dropout=False, dropout_prob=0.1,
                 norm=nn.BatchNorm2d, upsampling_mode='transpose'):
        super().__init__()
        if upsampling_mode != 'transpose':
            raise NotImplementedError("only upsampling_mode='transpose' (the one RenderingNet uses) is built")
        net = [nn.ConvTranspose2d(in_channels, out_channels, kernel_size=4, stride=2, padding=1, bias=norm is None)]
        if norm is not None:
            net += [norm(out_channels, affine=True)]
        net += [nn.ReLU(True)]
        if use_dropout:
            net += [nn.Dropout2d(dropout_prob, False)]
        if post_conv:
            net += [Conv2dSame(out_channels, out_channels, kernel_size=3, bias=norm is None)]
            if norm is not None:
                net += [norm(out_channels, affine=True)]
            net += [nn.ReLU(True)]
            if use_dropout:
                net += [nn.Dropout2d(0.1, False)]
        self.net = nn.Sequential(*net)


class DownBlock(nn.Module):
    def __init__(self, in_channels, out_channels, prep_conv=True, middle_channels=None, use_dropout=False,
                 dropout_prob=0.1, norm=nn.BatchNorm2d, stride=2, kernal_size=4):
        super().__init__()
        middle_channels = in_channels if middle_channels is None else middle_channels
        net = []
        if prep_conv:
            net += [nn.ReflectionPad2d(1), nn.Conv2d(in_channels, middle_channels, kernel_size=3, padding=0, stride=1,
                                                     bias=norm is None)]
            if norm is not None:
                net += [norm(middle_channels, affine=True)]
            net += [nn.LeakyReLU(0.2, True)]
            if use_dropout:
                net += [nn.Dropout2d(dropout_prob, False)]
        net += [nn.ReflectionPad2d(1), nn.Conv2d(middle_channels, out_channels, kernel_size=kernal_size, padding=0,
                                                 stride=stride, bias=norm is None)]
        if norm is not None:
            net += [norm(out_channels, affine=True)]
        net += [nn.LeakyReLU(0.2, True)]
        if use_dropout:
            net += [nn.Dropout2d(dropout_prob, False)]
        self.net = nn.Sequential(*net)


class UnetSkipConnectionBlock(nn.Module):
    def __init__(self, outer_nc, inner_nc, upsampling_mode, norm=nn.BatchNorm2d, submodule=None, use_dropout=False,
                 dropout_prob=0.1, flag_outer=True, gcn=False, out_channels_gcn=512, highway_mode='concat'):
        super().__init__()
        if highway_mode not in ('concat', 'residual', 'no_highway'):
            raise ValueError('Unrecognized option for highway_mode')
        self.submodule, self.flag_outer, self.gcn, self.highway_mode = submodule, flag_outer, gcn, highway_mode
        if gcn:   # dead at the output (pytorch_prototyping.py:407-419); kept so that checkpoints load strictly
            self.fuse = DownBlock(inner_nc + out_channels_gcn, inner_nc, use_dropout=use_dropout,
                                  dropout_prob=dropout_prob, norm=norm, stride=1, kernal_size=3)
        self.down = DownBlock(outer_nc, inner_nc, use_dropout=use_dropout, dropout_prob=dropout_prob, norm=norm)
        self.up = UpBlock(2 * inner_nc if flag_outer else inner_nc, outer_nc, use_dropout=use_dropout,
                          dropout_prob=dropout_prob, norm=norm, upsampling_mode=upsampling_mode)


class Unet(nn.Module):
    def __init__(self, in_channels, out_channels, nf0, num_down, max_channels, use_dropout, upsampling_mode='transpose',
                 dropout_prob=0.1, norm=nn.BatchNorm2d, outermost_linear=False, out_channels_gcn=512, use_gcn=True,
                 outermost_highway_mode='no_highway'):
        super().__init__()
        assert num_down > 0, 'Need at least one downsampling layer in UNet.'
        if norm is not nn.BatchNorm2d or not outermost_linear or outermost_highway_mode != 'concat' or \
                max_channels != 8 * nf0:
            raise NotImplementedError('only the RenderingNet configuration (BatchNorm2d, outermost_linear=True, '
                                      "highway 'concat', max_channels = 8 nf0; network.py:236-247) is built")
        self.use_gcn, self.outermost_highway_mode = use_gcn, outermost_highway_mode
        self.cfg = (in_channels, out_channels, nf0, num_down)
        layers = [Conv2dSame(in_channels, nf0, kernel_size=3, bias=False), norm(nf0, affine=True), nn.LeakyReLU(0.2, True)]
        if use_dropout:
            layers += [nn.Dropout2d(dropout_prob)]
        self.in_layer = nn.Sequential(*layers)
        c = min(2 ** (num_down - 1) * nf0, max_channels)
        self.unet_block = UnetSkipConnectionBlock(c, c, use_dropout=use_dropout, dropout_prob=dropout_prob, norm=None,
                                                  upsampling_mode=upsampling_mode, flag_outer=False)
        for i in list(range(1, num_down - 1))[::-1]:
            self.unet_block = UnetSkipConnectionBlock(min(2 ** i * nf0, max_channels), min(2 ** (i + 1) * nf0, max_channels),
                                                      use_dropout=use_dropout, dropout_prob=dropout_prob,
                                                      submodule=self.unet_block, norm=norm, upsampling_mode=upsampling_mode)
        self.unet_block = UnetSkipConnectionBlock(min(nf0, max_channels), min(2 * nf0, max_channels), use_dropout=use_dropout,
                                                  dropout_prob=dropout_prob, submodule=self.unet_block, norm=norm,
                                                  upsampling_mode=upsampling_mode, gcn=use_gcn,
                                                  out_channels_gcn=out_channels_gcn, highway_mode=outermost_highway_mode)
        self.out_layer = nn.Sequential(Conv2dSame(2 * nf0, out_channels, kernel_size=3, bias=True))
        self.out_layer_weight = self.out_layer[0].weight
        self._plans, self._plan_versions = {}, {}
        self.register_load_state_dict_post_hook(lambda m, k: m._plans.clear())

    def _apply(self, fn, *a, **k):     # .to()/.cuda() moves the weights: plans are rebuilt lazily
        self._plans, self._plan_versions = {}, {}
        self.__dict__.pop('_structure_cache', None)
        return super()._apply(fn, *a, **k)

    def _structure(self):
        """Module lists of the (static) network, collected once: three full `modules()` walks per forward call cost 1.4 ms of
        host time per view in the drop-in loop (cProfile, scripts/exp_sh_roundtrip.py).  Parameters / buffers are looked up
        through their owning modules on every call, so re-assigned tensors (`.data = ...`, load_state_dict) are still seen."""
        st = self.__dict__.get('_structure_cache')
        if st is None:
            mods = list(self.named_modules())
            st = {'bn': [m for _, m in mods if isinstance(m, nn.BatchNorm2d)],
                  'dropout': [m for _, m in mods if isinstance(m, nn.Dropout2d)],
                  'live_bn': [m for name, m in mods if isinstance(m, nn.BatchNorm2d) and '.fuse.' not in '.' + name + '.'],
                  'owners': [m for _, m in mods if m._parameters or m._buffers]}
            self.__dict__['_structure_cache'] = st
        return st

    def _weights_version(self):
        """Changes whenever a parameter or buffer is modified in place (optimizer step, `.data` assignment, copy_)."""
        ver = []
        for m in self._structure()['owners']:
            for t in m._parameters.values():
                if t is not None:
                    ver.append((id(t), t._version))
            for n, t in m._buffers.items():
                if t is not None and 'num_batches' not in n:
                    ver.append((id(t), t._version))
        return tuple(ver)

    def _plan(self, n, h, w, device):
        st = self._structure()
        bn_train = [m.training for m in st['bn']]
        if any(m.training for m in st['dropout']):
            raise NotImplementedError('Dropout2d in training mode: only inference (module.eval()) is built')
        if len(set(bn_train)) > 1:
            raise NotImplementedError('mixed BatchNorm train/eval modes')
        # train-mode BatchNorm2d reduces over the WHOLE batch of the call (torch semantics; the reference's scripts
        # call with N = 1, where this equals per-view statistics)
        mode = 'batch_all' if (bn_train and bn_train[0]) else 'running'
        key = (h, w, str(device), mode)
        ver = self._weights_version()
        plan = self._plans.get(key)
        # one plan per (size, device, mode), sized for the largest batch seen; rebuilt when a weight changed in place
        if plan is None or plan.N < n or self._plan_versions.get(key) != ver:
            cin, cout, nf0, nd = self.cfg
            # the kernels finalise BatchNorm with torch's defaults (eps 1e-5, momentum 0.1: what pytorch_prototyping.py builds);
            # a module edited to other values would silently diverge from torch, so it is refused
            for m in self._live_batchnorms():
                if m.eps != 1e-5 or m.momentum != 0.1:
                    raise NotImplementedError('Unet: BatchNorm2d(eps=%r, momentum=%r) — the HIP U-Net implements the reference\'s '
                                              'eps=1e-5, momentum=0.1 only' % (m.eps, m.momentum))
            sd = {'net.' + k: v for k, v in self.state_dict().items()}
            self._plans.pop(key, None)
            plan = UNetPlan(sd, cin, cout, nf0, nd, (h, w), max(n, plan.N if plan is not None else 0), device, bn_mode=mode,
                            update_running_stats=(mode == 'batch_all'))
            self._plans[key] = plan
            self._plan_versions[key] = ver
        return plan

    def _live_batchnorms(self):
        """BatchNorm modules the live path runs through (the `fuse` block of the GCN branch never reaches the output)."""
        return self._structure()['live_bn']

    def forward(self, x, v_fea=None):
        """x [N,Cin,H,W] -> raw out-layer output [N,Cout,H,W] (bias applied).  v_fea is accepted and unused: the
        reference's GCN branch never reaches the output (pytorch_prototyping.py:407-419).

        BatchNorm semantics follow torch: in train mode (what test_rnr.py:229-233 forces at inference) statistics are
        taken over the whole [N,H,W] batch of THIS call and running_mean / running_var / num_batches_tracked of the live
        layers are updated.  Deviation: with use_gcn=True the reference additionally runs the dead GCN pass, which
        updates the running statistics of `down`/`submodule`/`up`/`fuse` a second time from v_fea-dependent inputs; that
        pass is not executed here, so those side effects are absent (outputs are unaffected)."""
        return self.forward_fused(x, apply_tanh=False)

    def forward_fused(self, x, apply_tanh):
        n, _, h, w = x.shape
        plan = self._plan(n, h, w, x.device)
        raw = plan.forward(ops.nchw_to_nhwc(x.float().contiguous(), plan.in_c_pad))
        if plan.bn_mode == 'batch_all':
            # the kernels just updated running_mean / running_var in place: eval-mode plans folded the old values
            for k in [k for k in self._plans if k[3] == 'running']:
                del self._plans[k]
            tracked = [m.num_batches_tracked for m in self._live_batchnorms() if m.num_batches_tracked is not None]
            if tracked:
                torch._foreach_add_(tracked, 1)         # one launch instead of seventeen
        return ops.nhwc_to_nchw(raw, plan.out_channels, bias=plan.out_bias, apply_tanh=apply_tanh)


class Identity(nn.Module):
    def forward(self, x):
        return x
