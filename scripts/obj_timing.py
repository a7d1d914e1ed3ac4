"""Timing of nr.load_obj on the bench mesh (65 536-face UV sphere, `f v/vt/vn`), three parsers:
  native     the product: one C++ pass behind the C ABI (csrc/objparse.hip, rnr_obj_scan + rnr_obj_parse)
  py_loop    the round-1 product: one Python pass over the lines (kept here only as the comparison point)
  reference  /root/reference/neural_renderer/neural_renderer/load_obj.py:108-209 (four Python passes), only when the
             reference tree exists (build container)
Usage: python scripts/obj_timing.py        (CPU only; prints one JSON line)"""
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'relightable-nr_amd'))
import numpy as np  # noqa: E402
import torch  # noqa: E402


def py_loop(fp):
    v, vn, vt, fv, fvt, fvn = [], [], [], [], [], []
    with open(fp) as fh:
        for line in fh:
            tok = line.split()
            if not tok:
                continue
            key = tok[0]
            if key == 'v':
                v.append((float(tok[1]), float(tok[2]), float(tok[3])))
            elif key == 'vn':
                vn.append((float(tok[1]), float(tok[2]), float(tok[3])))
            elif key == 'vt':
                vt.append((float(tok[1]), float(tok[2])))
            elif key == 'f':
                parts = [p.split('/') for p in tok[1:]]
                fv.append([int(p[0]) for p in parts])
                fvt.append([int(p[1]) for p in parts])
                fvn.append([int(p[-1]) for p in parts])
    f32 = lambda a, w: np.asarray(a, np.float32).reshape(-1, w)
    i32 = lambda a: np.asarray(a, np.int32).reshape(-1, 3) - 1
    return f32(v, 3), f32(vn, 3), f32(vt, 2), i32(fv), i32(fvt), i32(fvn)


def best(fn, n=5):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        out = fn()
        ts.append(time.perf_counter() - t0)
    return min(ts), out


def main():
    from rnr_amd import scene
    import neural_renderer as nr
    mesh = scene.uv_sphere(128, 256)
    tmp = tempfile.mkdtemp(prefix='rnr_obj_')
    fp = os.path.join(tmp, 'sphere65k.obj')
    scene.write_obj(fp, mesh)
    res = {'file_bytes': os.path.getsize(fp), 'faces': int(mesh['f_v_idx'].shape[0]), 'vertices': int(mesh['v'].shape[0])}
    t_nat, (va, fa) = best(lambda: nr.load_obj(fp, normalization=False, use_cuda=False))
    t_py, ref = best(lambda: py_loop(fp), 3)
    res['native_s'], res['py_loop_s'] = t_nat, t_py
    got = [va['v'].numpy(), va['vn'].numpy(), va['vt'].numpy(), fa['f_v_idx'].numpy(), fa['f_vt_idx'].numpy(), fa['f_vn_idx'].numpy()]
    res['native_equals_py_loop'] = all(np.array_equal(a, b) for a, b in zip(got, ref))
    if os.path.isdir('/root/reference'):
        sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
        import make_golden
        m = make_golden.import_all()
        t_ref, (rv, rf) = best(lambda: m['nr'].load_obj(fp, normalization=False, use_cuda=False), 3)
        res['reference_s'] = t_ref
        refs = [rv['v'].numpy(), rv['vn'].numpy(), rv['vt'].numpy(), rf['f_v_idx'].numpy(), rf['f_vt_idx'].numpy(), rf['f_vn_idx'].numpy()]
        res['native_equals_reference'] = all(np.array_equal(a, b) for a, b in zip(got, refs))
        res['speedup_vs_reference'] = t_ref / t_nat
    res['speedup_vs_py_loop'] = t_py / t_nat
    print(json.dumps(res))


if __name__ == '__main__':
    main()
