"""ISA statistics of one kernel of conv.hip (cross-compiled here, no GPU): instruction mix, registers, spills, LDS.
Usage: python scripts/isa_stats.py KERNEL_SUBSTRING [FILE.hip] [extra hipcc flags ...]   e.g. conv_wino_kernel, or shade_inputs_kernel shade.hip"""
import re, subprocess, sys, collections, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
name = sys.argv[1]
hips = [f for f in sys.argv[2:] if f.endswith('.hip')]
src = os.path.join(ROOT, 'relightable-nr_amd', 'csrc', os.path.basename(hips[0]) if hips else 'conv.hip')
flags = [f for f in sys.argv[2:] if not f.endswith('.hip')]
out = '/tmp/isa_%d.s' % os.getpid()
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC'] + (['-fno-slp-vectorize'] if src.endswith('conv.hip') else ['-ffp-contract=off']) + flags + ['-S', '--cuda-device-only', src, '-o', out], stderr=subprocess.DEVNULL)
s = open(out).read()
for m in re.finditer(r'^(_Z\w*%s\w*):[^\n]*\n' % re.escape(name), s, re.M):
    i = m.end(); j = s.index('.end_amdhsa_kernel', i)
    body = s[i:s.index('s_endpgm', i)]
    c = collections.Counter(re.findall(r'^\s+((?:ds|buffer|global|flat|scratch|v_mfma|s_barrier|s_waitcnt|v_|s_)[a-z_0-9]*)', body, re.M))
    groups = collections.Counter()
    for k, v in c.items():
        g = ('mfma' if k.startswith('v_mfma') else 'valu' if k.startswith('v_') else 'lds' if k.startswith('ds_') else
             'vmem' if k.split('_')[0] in ('buffer', 'global', 'flat', 'scratch') else 'salu')
        groups[g] += v
    meta = dict(re.findall(r'\.amdhsa_(next_free_vgpr|next_free_sgpr|accum_offset|group_segment_fixed_size|private_segment_fixed_size)\s+(\d+)', s[j - 5000:j]))
    print(m.group(1)[:70]); print('  ', dict(groups)); print('  ', meta)
    print('  ', {k: v for k, v in sorted(c.items()) if k.split('_')[0] in ('ds', 'buffer', 'global', 'flat', 'scratch') or k in ('s_barrier',)})
os.remove(out)
