"""The bench line contract (driver-facing): the committed output of the default command carries every field the
contract names, with consistent values.  (The bench itself needs a GPU; this checks the committed evidence.)"""
import glob
import json
import os

from conftest import ROOT


def _latest():
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*', 'bench_default_*.json')))
    files = [f for f in files if 'under_rocprof' not in f]
    assert files, 'no committed default bench line under profiles/'
    return max(files, key=os.path.getmtime)


def test_committed_bench_line_has_the_contract_fields():
    with open(_latest()) as f:
        lines = [l for l in f.read().splitlines() if l.strip()]
    assert len(lines) == 1, 'bench.py prints exactly one JSON line'
    d = json.loads(lines[0])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
              'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in d, k
    assert d['higher_is_better'] is True and d['scaling'] == 'weak' and d['data'] == 'synthetic' and d['vs_baseline'] is None
    assert 'workload' in d['config'] and 'model' not in d['config']
    r = d['roofline']
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
        assert k in r, k
    assert r['bound'] in ('hbm', 'mfma') and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-9
    c = d['cpu_baseline']
    for k in ('value', 'unit', 'cores', 'kind', 'sample'):
        assert k in c, k
    assert c['kind'] in ('reference', 'port') and c['cores'] >= 1
    # value = streams * steps / time
    s = d['config']['streams_per_gpu'] * d['n_gpus']
    assert abs(d['value'] - s / (d['ms_per_step'] * 1e-3)) / d['value'] < 1e-6


def test_pmc_constants_come_from_the_committed_summary():
    """bench.py has no hard-coded counter values: the traffic / VALU figures of `roofline` are read from the newest committed
    rocprofv3 PMC summary, and a committed round-3 bench line must carry exactly those values."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(ROOT, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    path = os.path.join(ROOT, bench.LK_PMC['path'])
    e5 = json.load(open(path))['lk_track_g16_kernel<15>']
    # round 5: every counter family at 512 streams per launch, with the round's kernel (profiles/r05/pmc_frontend_s512_summary.json)
    assert bench.LK_PMC['path'] == os.path.join('profiles', 'r05', 'pmc_frontend_s512_summary.json')
    assert bench.LK_PMC['write_kb'] == e5['WRITE_SIZE'] and bench.LK_PMC['valu'] == e5['SQ_INSTS_VALU'] and bench.LK_PMC['fetch_kb'] == e5['FETCH_SIZE']
    assert bench.LK_PMC['streams'] == bench.LK_PMC['fetch_streams'] == 512
    assert abs(bench.LK_TRAFFIC_BYTES_PER_LAUNCH_S64 - (bench.FETCH_SIZE_FACTOR * e5['FETCH_SIZE'] + e5['WRITE_SIZE']) * 1024) < 1e-6
    e64 = json.load(open(os.path.join(ROOT, 'profiles', 'r05', 'pmc_frontend_s64_summary.json')))['lk_track_g16_kernel<15>']
    for c in ('FETCH_SIZE', 'SQ_INSTS_VALU'):                 # per-stream figures move by < 6 % between 64 and 512 streams: linear scaling holds
        assert abs((e5[c] / 512) / (e64[c] / 64) - 1.0) < 0.06, c
    # rounds 3 / 4: the committed lines of those rounds carry the constants of their own summaries
    e = json.load(open(os.path.join(ROOT, 'profiles', 'r03', 'pmc_frontend_s64_summary.json')))['lk_track_g16_kernel<15>']
    sc = json.load(open(os.path.join(ROOT, 'profiles', 'r04', 'pmc_fetch_scaling.json')))
    big = max(sc['streams'], key=int)
    f_big = sc['streams'][big]['lk_track_g16_kernel<15>']['FETCH_SIZE_KB_mean']
    per = sc['lk_fetch_kb_per_stream']
    assert abs(per[big] / per['64'] - 1.0) < 0.05
    calib = json.load(open(os.path.join(ROOT, 'profiles', 'r03', 'fetch_calib.json')))['kernels']
    for k in ('read16', 'read4', 'read1'):                      # the measured factor: FETCH_SIZE reports half of the bytes, whatever the lane width
        assert abs(1.0 / calib[k]['raw_over_known'] - bench.FETCH_SIZE_FACTOR) < 0.01, k
    # (the line printed under rocprofv3 --kernel-trace exists for its kernel stats; it was taken before the PMC passes of the same call)
    lines = [f for f in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r03', 'bench_*.json'))) if 'under_rocprof' not in f]
    for f in lines:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        if 'pmc_constants' in d and d['pmc_constants']['path'] == os.path.join('profiles', 'r03', 'pmc_frontend_s64_summary.json'):
            assert d['pmc_constants']['fetch_kb'] == e['FETCH_SIZE'] and d['pmc_constants']['valu'] == e['SQ_INSTS_VALU'], f
    for f in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r04', 'bench_*.json'))):
        if 'under_rocprof' in f:
            continue
        d = json.loads(open(f).read().strip().splitlines()[-1])
        if 'pmc_constants' in d:
            assert d['pmc_constants']['fetch_kb'] == f_big and d['pmc_constants']['valu'] == e['SQ_INSTS_VALU'], f
    for f in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r05', 'bench_*_r05.json'))):
        d = json.loads(open(f).read().strip().splitlines()[-1])
        if 'pmc_constants' in d:
            assert d['pmc_constants']['fetch_kb'] == e5['FETCH_SIZE'] and d['pmc_constants']['valu'] == e5['SQ_INSTS_VALU'], f
