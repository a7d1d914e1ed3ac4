"""Drop-in `dataio.ViewDataset` for inference (reference: dataio.py:11-247): calib.mat -> per-view camera tensors.

Only what `test_rnr.py:112-127, 268-278` needs: `load_img=False`, `load_precompute=False` (image loading and the
precomputed-map loader are outside the hot-path build and raise) and a cv2-free `LightProbeDataset`.

Unlike the reference, `read_view` is pure: the reference writes the cropped intrinsics back into `calib['projs']`
through a numpy view (dataio.py:189-193), so calling it twice for the same index compounds the crop.  The values
returned here equal the reference's FIRST call, which is what `buffer_all()` stores and the scripts consume.
"""
import os

import numpy as np
import scipy.io
import torch


class ViewDataset():
    def __init__(self, root_dir, calib_path, calib_format, img_size, sampling_pattern, load_img=True, img_dir=None,
                 ignore_dist_coeffs=True, load_precompute=False, precomp_high_dir=None, precomp_low_dir=None, img_gamma=1.0):
        if load_img or load_precompute:
            raise NotImplementedError('ViewDataset: image / precomputed-map loading is outside the inference hot path')
        if not os.path.isdir(root_dir):
            raise ValueError("Error! root dir is wrong")
        if calib_format != 'convert':
            raise ValueError('Unknown calib format')
        if not os.path.isfile(calib_path):
            raise ValueError("Error! calib path is wrong")
        self.root_dir, self.calib_format, self.img_size = root_dir, calib_format, img_size
        self.ignore_dist_coeffs, self.load_img, self.load_precompute = ignore_dist_coeffs, load_img, load_precompute
        self.img_gamma = img_gamma
        calib = scipy.io.loadmat(calib_path)
        self.global_RT = calib['global_RT']
        self.global_RT_inv = np.linalg.inv(self.global_RT)
        num_view = calib['poses'].shape[0]
        keep = self._keep_indices(sampling_pattern, num_view, calib)
        self.calib = {k: np.array(calib[k][keep, ...]) for k in ['img_hws', 'projs', 'poses', 'dist_coeffs']}
        self.calib['global_RT'] = self.global_RT
        self.poses_all = [self.calib['poses'][i] for i in range(len(keep))]
        self.img_fp_all = ['x.x'] * len(keep)
        self.img_idx2fn = ['x.x'] * len(keep)
        self.img_fn2idx = {'x.x': len(keep) - 1} if keep else {}
        self.views_all = None

    @staticmethod
    def _keep_indices(pattern, n, calib):
        """dataio.py:72-125."""
        if pattern == 'all':
            return list(range(n))
        if pattern == 'filter':
            return [int(i) for i in calib['keep_id'][0, :]]
        kind, val = pattern.split('_')[0], pattern.split('_')[-1]
        if kind == 'first':
            return list(range(int(val)))
        if kind == 'after':
            return list(range(int(val), n))
        if kind == 'skip':
            return list(range(0, n, int(val)))
        if kind == 'skipinv':
            return [i for i in range(n) if i % int(val) != 0]
        if kind == 'only':
            return [int(val)]
        raise ValueError("Unknown sampling pattern!")

    def __len__(self):
        return len(self.img_fp_all)

    def buffer_all(self):
        self.views_all = [self.read_view(i) for i in range(len(self))]

    def buffer_one(self):
        self.views_all = [self.read_view(0)]

    def read_view(self, idx):
        """dataio.py:161-213 with load_img=False: square centre crop of the original image resized to img_size."""
        img_hw = self.calib['img_hws'][idx, :]
        min_dim = np.amin(img_hw)
        center = img_hw // 2
        center_new = np.array([min_dim // 2, min_dim // 2])
        crop = np.array([min_dim, min_dim])
        pose = np.dot(self.poses_all[idx], self.global_RT_inv)
        proj_orig = np.array(self.calib['projs'][idx], copy=True)
        dist = np.array(self.calib['dist_coeffs'][idx], copy=True)
        if self.ignore_dist_coeffs:
            dist[:] = 0.0
        offset = np.array([center_new[0] - center[0], center_new[1] - center[1]], dtype=np.float32)
        scale = np.array([self.img_size[0] * 1.0 / (crop[0] * 1.0), self.img_size[1] * 1.0 / (crop[1] * 1.0)], dtype=np.float32)
        proj = proj_orig.copy()
        proj[0, -1] = (proj[0, -1] + offset[1]) * scale[1]
        proj[1, -1] = (proj[1, -1] + offset[0]) * scale[0]
        proj[0, 0] *= scale[1]
        proj[1, 1] *= scale[0]
        f32 = lambda a: torch.from_numpy(np.asarray(a).astype(np.float32))
        return {'proj_orig': f32(proj_orig), 'proj': f32(proj), 'pose': f32(pose), 'dist_coeffs': f32(dist),
                'offset': torch.from_numpy(offset), 'scale': torch.from_numpy(scale), 'view_dir': f32(-pose[2, :3]),
                'proj_inv': f32(np.linalg.inv(proj)), 'R_inv': f32(pose[:3, :3].transpose()), 'idx': idx, 'img_fn': 'x.x'}

    def __getitem__(self, idx):
        """The scripts index the dataset and take element 0 (test_rnr.py:268-272)."""
        view = self.views_all[idx] if self.views_all is not None else self.read_view(idx)
        return [view]


def _read_radiance_hdr(fp):
    """Radiance .hdr (RGBE, flat or new-style RLE scanlines, -Y H +X W orientation) -> float32 [H,W,3] RGB."""
    with open(fp, 'rb') as fh:
        data = fh.read()
    pos = data.index(b'\n\n') + 2
    end = data.index(b'\n', pos)
    dims = data[pos:end].split()
    if len(dims) != 4 or dims[0] != b'-Y' or dims[2] != b'+X':
        raise ValueError('%s: unsupported Radiance orientation %r' % (fp, data[pos:end]))
    H, W = int(dims[1]), int(dims[3])
    buf = np.frombuffer(data, np.uint8, offset=end + 1)
    rgbe = np.empty((H, W, 4), np.uint8)
    p = 0
    for y in range(H):
        if W < 8 or W > 0x7fff or buf[p] != 2 or buf[p + 1] != 2 or (buf[p + 2] & 0x80):
            rgbe[y] = buf[p:p + 4 * W].reshape(W, 4)           # flat scanline
            p += 4 * W
            continue
        p += 4
        for ch in range(4):
            x = 0
            while x < W:
                n = int(buf[p]); p += 1
                if n > 128:                                     # run
                    n -= 128
                    rgbe[y, x:x + n, ch] = buf[p]; p += 1
                else:                                           # literal
                    rgbe[y, x:x + n, ch] = buf[p:p + n]; p += n
                x += n
    e = rgbe[..., 3].astype(np.int32)
    scale = np.where(e > 0, np.ldexp(1.0, e - 136), 0.0).astype(np.float32)
    return rgbe[..., :3].astype(np.float32) * scale[..., None]


class LightProbeDataset():
    """dataio.py:262-311 without cv2: items are {'lp_img': float32 tensor [3,H,W], RGB, ** img_gamma}.  Readers: Radiance
    `.hdr` (numpy RGBE decoder), `.npy` ([H,W,3] float RGB), 8-bit `.png/.jpg/.JPEG/.bmp` via PIL (/255).  `.exr` / `.mat`
    files are LISTED (they take part in the order) but raise when read: they need a codec this image does not ship."""
    # data_util.glob_imgs' list (data_util.py:57), case-sensitive like glob, so that the sorted order — which lighting_idx
    # indexes — is the reference's; '.npy' is this build's extra (a probe the reference could not list)
    _EXT = ('.png', '.jpg', '.JPEG', '.bmp', '.exr', '.hdr', '.mat', '.npy')

    def __init__(self, data_dir, img_gamma=1.0):
        self.data_dir, self.img_gamma = data_dir, img_gamma
        if not os.path.isdir(data_dir):
            raise ValueError("Error! data dir is wrong")
        self.lp_fp_all = sorted(os.path.join(data_dir, f) for f in os.listdir(data_dir) if f.endswith(self._EXT))
        self.lp_all = [None] * len(self.lp_fp_all)

    def buffer_one(self, idx):
        if self.lp_all[idx] is not None:
            return
        fp = self.lp_fp_all[idx]
        ext = os.path.splitext(fp)[1].lower()
        if ext == '.hdr':
            img = _read_radiance_hdr(fp)
        elif ext == '.npy':
            img = np.load(fp).astype(np.float32)[:, :, :3]
        elif ext in ('.exr', '.mat'):
            raise NotImplementedError('%s: %s probes need a codec this image does not ship (cv2 / OpenEXR / the reference\'s own '
                                      '.mat convention) - convert the probe to .hdr or .npy' % (fp, ext))
        else:
            from PIL import Image
            im = Image.open(fp)
            if im.mode not in ('RGB', 'RGBA', 'L', 'P'):       # 16-bit / float images: cv2.IMREAD_UNCHANGED keeps their range, /255 here would not
                raise NotImplementedError('%s: %s-mode image; only 8-bit probes are read here - convert to .hdr or .npy' % (fp, im.mode))
            img = np.asarray(im.convert('RGB'), np.float32) / 255.0
        img = np.ascontiguousarray(img.transpose(2, 0, 1)) ** self.img_gamma
        self.lp_all[idx] = {'lp_img': torch.from_numpy(img.astype(np.float32))}

    def buffer_all(self):
        for idx in range(len(self.lp_fp_all)):
            self.buffer_one(idx)

    def __len__(self):
        return len(self.lp_fp_all)

    def __getitem__(self, idx):
        self.buffer_one(idx)
        return self.lp_all[idx]
