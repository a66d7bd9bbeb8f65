#!/usr/bin/env python3
"""tools/summarize_crop.py DIR — per-launch averages of the PMC counters of tools/pmc_crop.sh for every maray kernel,
and the figures derived from them: instructions per wavefront, VALU issue-port occupancy against the f64 peak."""
import collections
import csv
import glob
import json
import sys

d = sys.argv[1]
run = json.loads([l for l in open(d + '/run.json') if l.startswith('{')][-1])
out = {'run': run, 'code_key': run.get('code_key'), 'library': run.get('library'), 'commit': None, 'kernels': {}}      # commit: stamped by tools/collect_r4.py (the GPU box has no .git)
stats = glob.glob(d + '/trace/*/*_kernel_stats.csv')
if stats:
    out['kernel_stats'] = [r for r in csv.DictReader(open(stats[0])) if 'maray' in r['Name']]
per = collections.defaultdict(lambda: collections.defaultdict(list))
res = {}
for f in sorted(glob.glob(d + '/pmc_*/*/*_counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        if 'maray' not in r['Kernel_Name']:
            continue
        per[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
        res[r['Kernel_Name']] = {k: r[k] for k in ('VGPR_Count', 'Accum_VGPR_Count', 'SGPR_Count', 'Scratch_Size', 'LDS_Block_Size', 'Workgroup_Size', 'Grid_Size')}
for k, c in per.items():
    avg = {n: sum(v) / len(v) for n, v in c.items()}
    dv = {}
    waves = avg.get('SQ_WAVES')
    if waves:
        for n in ('SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_INSTS_SMEM', 'SQ_INSTS_LDS', 'SQ_INSTS_VMEM_WR', 'SQ_INSTS_VMEM_RD'):
            if n in avg:
                dv[n.lower().replace('sq_insts_', '') + '_insts_per_wave'] = avg[n] / waves
        if 'SQ_WAVE_CYCLES' in avg:
            dv['wave_cycles_per_wave (SQ_WAVE_CYCLES*4/SQ_WAVES)'] = avg['SQ_WAVE_CYCLES'] * 4 / waves
    if 'GRBM_GUI_ACTIVE' in avg:
        cyc = avg['GRBM_GUI_ACTIVE'] / 8          # the counter sums the 8 XCDs
        dv['gpu_cycles_per_launch'] = cyc
        if 'SQ_INSTS_VALU' in avg:
            per_simd = avg['SQ_INSTS_VALU'] / 1024
            dv['valu_insts_per_simd'] = per_simd
            dv['cycles_per_valu_inst_per_simd'] = cyc / per_simd
            # a wave64 VALU instruction occupies its SIMD's 16 lanes for 4 cycles (f64 and 32-bit alike at full rate)
            dv['valu_issue_port_busy_frac (4 cycles per inst)'] = 4 * per_simd / cyc
            dv['valu_insts_per_s_vs_39.3T_peak_lane_ops'] = None
        if 'SQ_INSTS_SALU' in avg:
            dv['salu_insts_per_cycle_per_cu'] = avg['SQ_INSTS_SALU'] / 256 / cyc
    if 'SQ_ACTIVE_INST_VALU' in avg and 'SQ_BUSY_CYCLES' in avg:
        dv['SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES'] = avg['SQ_ACTIVE_INST_VALU'] / avg['SQ_BUSY_CYCLES']
    if 'TCP_TOTAL_CACHE_ACCESSES_sum' in avg and 'TCP_TCC_READ_REQ_sum' in avg and avg['TCP_TOTAL_CACHE_ACCESSES_sum']:
        dv['tcp_hit_rate (1 - TCP_TCC_READ_REQ / TCP_TOTAL_CACHE_ACCESSES)'] = 1 - avg['TCP_TCC_READ_REQ_sum'] / avg['TCP_TOTAL_CACHE_ACCESSES_sum']
    if 'TCC_HIT_sum' in avg and avg.get('TCC_HIT_sum', 0) + avg.get('TCC_MISS_sum', 0):
        dv['l2_hit_rate'] = avg['TCC_HIT_sum'] / (avg['TCC_HIT_sum'] + avg['TCC_MISS_sum'])
    if 'FETCH_SIZE' in avg:
        dv['hbm_fetch_bytes_x2_gfx950_correction'] = avg['FETCH_SIZE'] * 2048
    if 'WRITE_SIZE' in avg:
        dv['hbm_write_bytes'] = avg['WRITE_SIZE'] * 1024
    out['kernels'][k] = {'per_launch_average': avg, 'resources': res.get(k), 'derived': dv}
print(json.dumps(out, indent=1))
