#!/usr/bin/env python3
"""tools/collect_r3.py — copies what tools/gpu_profile_r3.sh left in gpurun_out/ into profiles/r3_* (the files DESIGN.md
section 7 quotes).  Run here after the GPU call; `tools/summarize_profile.py jit maray_jit_pixels --round r3` first."""
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, 'gpurun_out')
P = os.path.join(ROOT, 'profiles')


def last_json_line(path):
    return [l for l in open(path) if l.startswith('{')][-1]


subprocess.check_call([sys.executable, os.path.join(ROOT, 'tools', 'summarize_profile.py'), 'jit', 'maray_jit_pixels', '--round', 'r3'])
open(os.path.join(P, 'r3_jit_chess4096_bench.json'), 'w').write(last_json_line(os.path.join(G, 'bench_r3.json')))
for tag in ('board', 'sky', 'allops', 'radial', 'soup1000'):
    src = os.path.join(G, 'pmc_r3_%s.json' % tag)
    if os.path.exists(src):
        shutil.copy(src, os.path.join(P, 'r3_crop_%s_pmc.json' % tag))
for src, dst in (('ablations_r3.jsonl', 'r3_ablations.jsonl'), ('other_configs_r3.json', 'r3_other_configs.json'),
                 ('soups_r3.jsonl', 'r3_soups.jsonl'), ('sizes_r3.jsonl', 'r3_sizes.jsonl')):
    if os.path.exists(os.path.join(G, src)):
        shutil.copy(os.path.join(G, src), os.path.join(P, dst))
with open(os.path.join(P, 'r3_config4_one_gpu.jsonl'), 'w') as f:
    f.write(open(os.path.join(G, 'strong_r3.jsonl')).read())
    f.write(last_json_line(os.path.join(G, 'bench_strong_r3.json')))
for tag in ('tape_smem', 'tape_lds'):
    d = os.path.join(G, 'prof_' + tag)
    stats = max(glob.glob(d + '/trace/*/*_kernel_stats.csv'), key=os.path.getmtime)
    shutil.copy(stats, os.path.join(P, 'r3_%s_chess4096_kernel_stats.csv' % tag))
    open(os.path.join(P, 'r3_%s_chess4096_bench_under_rocprof.json' % tag), 'w').write(last_json_line(d + '/bench_trace.json'))
b = json.loads(last_json_line(os.path.join(G, 'bench_r3.json')))
print('bench: %.0f Mpx/s, %.4f ms/step, roofline %.3f, code key %s' % (b['value'], b['ms_per_step'], b['roofline']['frac'], b['config']['code_key']))
pm = json.load(open(os.path.join(P, 'r3_jit_chess4096_pmc.json')))
print('pmc profile: commit %s code key %s' % (pm.get('commit'), pm.get('code_key')))
