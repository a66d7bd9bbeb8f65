#!/usr/bin/env python3
"""tools/summarize_profile.py TAG KERNEL [--round r1] — condense gpurun_out/prof_TAG (tools/profile_bench.sh) into profiles/.

Writes profiles/<round>_<TAG>_chess4096_{kernel_stats.csv, bench_under_rocprof.json, pmc.json}: the rocprofv3
kernel-trace statistics as they are, and per-launch averages of every PMC counter for KERNEL plus the figures derived
from them (HBM bytes with the gfx950 x2 correction on FETCH_SIZE, instructions per wavefront, cycles per VALU
instruction per SIMD).
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag, kernel = sys.argv[1], sys.argv[2]
    rnd = sys.argv[sys.argv.index('--round') + 1] if '--round' in sys.argv else 'r2'
    src = os.path.join(ROOT, 'gpurun_out', 'prof_' + tag)
    dst = os.path.join(ROOT, 'profiles', '%s_%s_chess4096_' % (rnd, tag))
    newest = lambda pattern: max(glob.glob(pattern), key=os.path.getmtime)      # gpurun_out/ keeps earlier runs' files too
    stats = newest(src + '/trace/*/*_kernel_stats.csv')
    shutil.copy(stats, dst + 'kernel_stats.csv')
    line = [l for l in open(src + '/bench_trace.json') if l.startswith('{')][-1]
    open(dst + 'bench_under_rocprof.json', 'w').write(line)
    bench = json.loads(line)

    avg, res, cmd_steps = {}, {}, None
    for f in [newest(d + '/*/*_counter_collection.csv') for d in sorted(glob.glob(src + '/pmc_*')) if os.path.isdir(d)]:
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if kernel not in r['Kernel_Name']:
                continue
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
            res = {k: r[k] for k in ('VGPR_Count', 'Accum_VGPR_Count', 'SGPR_Count', 'Scratch_Size', 'LDS_Block_Size',
                                     'Workgroup_Size', 'Grid_Size')}
        for k, v in agg.items():
            avg[k] = sum(v) / len(v)
    px = bench['config']['pixels_per_step']
    waves = avg.get('SQ_WAVES', px / 64)
    d = {}
    # FETCH_SIZE / WRITE_SIZE count in KiB on this stack; gfx950 under-reports fetches by 2x (MI355X_MICROARCH.md)
    if 'WRITE_SIZE' in avg:
        d['hbm_write_bytes'] = avg['WRITE_SIZE'] * 1024
        d['hbm_write_bytes_algorithmic'] = px * 3
    if 'FETCH_SIZE' in avg:
        d['hbm_fetch_bytes_raw_counter'] = avg['FETCH_SIZE'] * 1024
        d['hbm_fetch_bytes_x2_gfx950_correction'] = avg['FETCH_SIZE'] * 2048
    if 'SQ_INSTS_VALU' in avg:
        d['valu_insts_per_wave'] = avg['SQ_INSTS_VALU'] / waves
        d['salu_insts_per_wave'] = avg['SQ_INSTS_SALU'] / waves
    if 'SQ_INSTS_SMEM' in avg:
        d['smem_insts_per_wave'] = avg['SQ_INSTS_SMEM'] / waves
        d['lds_insts_per_wave'] = avg['SQ_INSTS_LDS'] / waves
    if 'GRBM_GUI_ACTIVE' in avg and 'SQ_INSTS_VALU' in avg:
        cyc = avg['GRBM_GUI_ACTIVE'] / 8                       # the counter sums the 8 XCDs
        d['gpu_cycles_per_launch (GRBM_GUI_ACTIVE/8 XCDs)'] = cyc
        d['cycles_per_valu_inst_per_simd'] = cyc / (avg['SQ_INSTS_VALU'] / (256 * 4))
        d['salu_insts_per_cycle_per_cu'] = avg['SQ_INSTS_SALU'] / 256 / cyc
    if 'SQ_WAVE_CYCLES' in avg:
        d['wave_cycles_per_wave (SQ_WAVE_CYCLES*4/SQ_WAVES)'] = avg['SQ_WAVE_CYCLES'] * 4 / waves
    import subprocess
    commit = subprocess.run(['git', '-C', ROOT, 'rev-parse', 'HEAD'], capture_output=True, text=True).stdout.strip() or None
    out = {'commit': commit, 'code_key': bench['config'].get('code_key'),
           'command': 'rocprofv3 --pmc <counters> --output-format csv -- python3 bench.py %s --steps 3 --warmup 1 '
                      '(tools/profile_bench.sh: one run per counter group)' % ' '.join(sys.argv[3:] if '--round' not in sys.argv else []),
           'kernel': kernel, 'workload': bench['config']['workload'] + ', one launch = %d pixels' % px,
           'per_launch_average': avg, 'resources': res, 'derived': d}
    json.dump(out, open(dst + 'pmc.json', 'w'), indent=1)
    print(json.dumps(d, indent=1))
    print(open(stats).read()[:600])


if __name__ == '__main__':
    main()
