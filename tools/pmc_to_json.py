"""Summarise gpurun_out/<tag>_{sq,fetch,write}_counters.csv (tools/gpu_pmc.sh) into profiles/pmc_dominant_kernel.json:
per-launch HBM bytes (FETCH_SIZE x2 for the gfx950 128-B-request under-count, + WRITE_SIZE), MFMA-busy, clock."""
import collections
import csv
import json
import os
import sys

tag, batch = sys.argv[1], int(sys.argv[2])
match = sys.argv[3] if len(sys.argv) > 3 else 'conv_mfma_kernel'      # kernel-name substring
out_name = sys.argv[4] if len(sys.argv) > 4 else 'pmc_dominant_kernel.json'
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
res, kernel = {}, None
for grp in ('sq', 'fetch', 'write'):
    rows = list(csv.DictReader(open(os.path.join(root, 'gpurun_out', '%s_%s_counters.csv' % (tag, grp)))))
    agg, dur = collections.defaultdict(list), []
    for r in rows:
        if match in r['Kernel_Name']:
            kernel = r['Kernel_Name']
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
            dur.append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    for c, v in agg.items():
        res[c] = sum(v) / len(v)
    res['duration_us_' + grp] = sum(dur) / len(dur) / 1e3
simd_cycles = res['GRBM_GUI_ACTIVE'] / 8 * 1024
out = {
    'kernel': kernel.replace('void ', '').replace('(ConvParamsBf16)', '').replace('(ConvDmaParams)', '').replace('(ConvParams)', '').replace('(WinoParams)', '').replace('(WinoWgradParams)', ''), 'batch': batch,
    'shape': '3x3 256->256 on (B,160,160,256), GN stats epilogue (tools/conv_single.py %s)' % os.environ.get('CONV_ARGS', ''),
    'hbm_read_bytes_per_launch': res['FETCH_SIZE'] * 1024 * 2, 'hbm_write_bytes_per_launch': res['WRITE_SIZE'] * 1024,
    'hbm_bytes_per_launch': res['FETCH_SIZE'] * 1024 * 2 + res['WRITE_SIZE'] * 1024,
    'algorithmic_bytes_per_launch': batch * 160 * 160 * 256 * 4 * 2 + 256 * 2304 * 4,
    'mfma_busy_frac': res['SQ_VALU_MFMA_BUSY_CYCLES'] / simd_cycles,
    'waves_per_simd': 4 * res['SQ_WAVE_CYCLES'] / simd_cycles,
    'effective_clock_ghz': res['GRBM_GUI_ACTIVE'] / 8 / res['duration_us_sq'] / 1e3,
    'tcc_hit_rate': res['TCC_HIT_sum'] / (res['TCC_HIT_sum'] + res['TCC_MISS_sum']),
    'lds_bank_conflict_cycles': res['SQ_LDS_BANK_CONFLICT'], 'duration_us': res['duration_us_sq'],
    'wait_any_frac': res['SQ_WAIT_ANY'] / res['SQ_WAVE_CYCLES'], 'source': 'tools/gpu_pmc.sh (rocprofv3 --pmc, 3 passes)',
}
sys.path.insert(0, root)
from bench import kernel_source_sha16  # noqa: E402
out['source_sha16'] = kernel_source_sha16(match)      # bench.py only replays this figure for the same kernel source
if 'bf16' in match:
    out['algorithmic_bytes_per_launch'] = batch * 160 * 160 * 256 * 2 * 2 + 256 * 2304 * 2
if os.environ.get('ALGO_BYTES'):          # other shapes: the caller states the algorithmic bytes and the shape
    out['algorithmic_bytes_per_launch'] = int(float(os.environ['ALGO_BYTES']))
    out['shape'] = '%s (tools/conv_single.py %s)' % (os.environ.get('SHAPE_DESC', ''), os.environ.get('CONV_ARGS', ''))
out['traffic_over_algorithmic'] = out['hbm_bytes_per_launch'] / out['algorithmic_bytes_per_launch']
json.dump(out, open(os.path.join(root, 'profiles', out_name), 'w'), indent=1)
print(json.dumps(out, indent=1))
