#!/usr/bin/env python3
"""tools/collect_r4.py — copies what tools/gpu_profile_r4.sh left in gpurun_out/ into profiles/r4_* (the files DESIGN.md
section 7 quotes).  Run here after the GPU call; `tools/summarize_profile.py jit maray_jit_pixels --round r4` first."""
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


subprocess.check_call([sys.executable, os.path.join(ROOT, 'tools', 'summarize_profile.py'), 'jit', 'maray_jit_pixels', '--round', 'r4'])
open(os.path.join(P, 'r4_jit_chess4096_bench.json'), 'w').write(last_json_line(os.path.join(G, 'bench_r4.json')))
commit = subprocess.check_output(['git', '-C', ROOT, 'rev-parse', '--short=12', 'HEAD'], text=True).strip()
dirty = bool(subprocess.check_output(['git', '-C', ROOT, 'status', '--porcelain', '--', 'maray_amd', 'include', 'bench.py'], text=True).strip())
for tag in ('board', 'sky', 'textured', 'allops', 'radial'):
    src = os.path.join(G, 'pmc_r4_%s.json' % tag)
    if os.path.exists(src):
        j = json.load(open(src))
        j['commit'] = commit + ('+uncommitted changes' if dirty else '')      # the tree the GPU call was sent from (collect right after it)
        json.dump(j, open(os.path.join(P, 'r4_crop_%s_pmc.json' % tag), 'w'), indent=1)
for src, dst in (('ablations_r4.jsonl', 'r4_ablations.jsonl'), ('other_configs_r4.json', 'r4_other_configs.json'), ('other_configs_r4_texel_per_app.json', 'r4_other_configs_texel_per_app.json'),
                 ('r4_first_call.txt', 'r4_first_call.txt'), ('bench_r4_n2.json', 'r4_bench_gpus2_rehearsal.json'),
                 ('soups_r4.jsonl', 'r4_soups.jsonl'), ('sizes_r4.jsonl', 'r4_sizes.jsonl')):
    if os.path.exists(os.path.join(G, src)):
        shutil.copy(os.path.join(G, src), os.path.join(P, dst))
with open(os.path.join(P, 'r4_config4_one_gpu.jsonl'), 'w') as f:
    f.write(last_json_line(os.path.join(G, 'bench_strong_r4.json')))
for tag in ('tape_smem', 'tape_lds'):
    d = os.path.join(G, 'prof_' + tag)
    stats = max(glob.glob(d + '/trace/*/*_kernel_stats.csv'), key=os.path.getmtime)
    shutil.copy(stats, os.path.join(P, 'r4_%s_chess4096_kernel_stats.csv' % tag))
    open(os.path.join(P, 'r4_%s_chess4096_bench_under_rocprof.json' % tag), 'w').write(last_json_line(d + '/bench_trace.json'))
b = json.loads(last_json_line(os.path.join(G, 'bench_r4.json')))
print('bench: %.0f Mpx/s, %.4f ms/step, roofline %.3f, code key %s' % (b['value'], b['ms_per_step'], b['roofline']['frac'], b['config']['code_key']))
pm = json.load(open(os.path.join(P, 'r4_jit_chess4096_pmc.json')))
print('pmc profile: commit %s code key %s' % (pm.get('commit'), pm.get('code_key')))
