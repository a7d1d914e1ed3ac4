"""Lighting front-end on the HIP operators (config 5's "4096-sample env-map SH projection").

  envmap_to_sh        LightingLP.__init__ sampling (network.py:665-672) + LightingLP.fit_sh (network.py:694-699):
                      bilinear taps of an equirect environment map at the sphere sample directions, then the uniform-
                      quadrature SH projection.  (The reference first resizes the probe to 1600x3200 with cv2
                      INTER_AREA; that host-side resize is out of scope — pass the probe at the resolution you want.)
  SHLighting          LightingSH (network.py:534-627) reduced to what the frame path needs: basis on the 100x200
                      reconstruction grid, per-call reconstruction of the light probe from coefficients.
"""
import numpy as np
import torch

from . import ops


def spherical_mapping(l_dir):
    """render.py:87-93: directions [3,n] -> equirect uv [2,n] (y-up)."""
    return torch.stack((torch.atan2(l_dir[2], l_dir[0]) * 0.5 / np.pi + 0.5, torch.acos(l_dir[1]) * 1.0 / np.pi), dim=0)


def spherical_mapping_inv(uv):
    """render.py:105-121."""
    y = torch.cos(uv[1] * np.pi)
    s = (1 - y ** 2).sqrt()
    a = uv[0] * 2 - 1
    x = s * torch.cos(a * np.pi)
    z = s * torch.sin(a * np.pi)
    z = z * ((~(a == 1.0)).to(s.dtype) * 2 - 1)
    z = z * ((~(a == -1.0)).to(s.dtype) * 2 - 1)
    return torch.nn.functional.normalize(torch.stack((x, y, z), dim=0), dim=0)


def envmap_to_sh(envmap, l_dir, lmax):
    """envmap [H,W,3] (device), l_dir [3,ns] unit sample directions (device) -> SH coefficients [(lmax+1)^2, 3]."""
    H, W = envmap.shape[0], envmap.shape[1]
    uv = spherical_mapping(l_dir.float())
    x = (uv[0] * float(W)).clamp(max=W - 1).contiguous()
    y = (uv[1] * float(H)).clamp(max=H - 1).contiguous()
    samples = ops.interpolate_bilinear(envmap.float().contiguous(), x, y)                  # [ns,3]
    basis = ops.sh_basis(l_dir.t().contiguous().float(), lmax)                             # [ns,nb]
    return ops.sh_fit(samples.contiguous(), basis), samples, basis


class SHLighting:
    """Light-probe reconstruction from SH coefficients on the lp_recon_h x lp_recon_w equirect grid."""

    def __init__(self, lmax, device, lp_recon_h=100, lp_recon_w=200):
        self.lmax, self.h, self.w = int(lmax), int(lp_recon_h), int(lp_recon_w)
        vv, uu = torch.meshgrid(torch.arange(self.h, dtype=torch.float32) / (self.h - 1),
                                torch.arange(self.w, dtype=torch.float32) / (self.w - 1), indexing='ij')
        dirs = spherical_mapping_inv(torch.stack([uu, vv]).flatten(1)).permute(1, 0).contiguous()   # network.py:574-579
        self.basis_recon = ops.sh_basis(dirs.to(device), self.lmax)                                  # [h*w, nb]

    def light_probe(self, coeff):
        """coeff [(lmax+1)^2, 3] -> [h,w,3]  (LightingSH.reconstruct_lp, network.py:622-627)."""
        return ops.sh_reconstruct(self.basis_recon, coeff.float().contiguous()).reshape(self.h, self.w, -1)
