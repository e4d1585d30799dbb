"""ONE pipeline of 2,048 streams against TWO of 1,024 / FOUR of 512 (own front-end engine, own batched filter, own host thread and HIP
streams each) on one GPU -- do independent pipelines fill each other's gaps (every kernel boundary of an in-order stream drains the
machine; LK is bound by VALU issue, the pyramids by HBM, the glue kernels by latency)?  Front-end alone (argv[1] = fe) or the complete
path (default).  Frames: 16 rendered streams x replicas (identical replicas: throughput only)."""
import json, sys, threading, time
import numpy as np
import torch
sys.path.insert(0, '.')
from uav_airvision_amd.config import ConfigEuRoC
from uav_airvision_amd.frontend import FrontendEngine
from uav_airvision_amd.msckf_ops import BatchedMSCKF
from uav_airvision_amd import _native as N
FE_ONLY = len(sys.argv) > 1 and sys.argv[1] == 'fe'
from uav_airvision_amd.synth import SyntheticStream

cfg = ConfigEuRoC(grid_max_feature_num=15, grid_min_feature_num=8)
U, F, S = 16, 44, 2048
dev = torch.device('cuda:0')
from uav_airvision_amd.synth import make_texture
streams = [SyntheticStream(cfg, seed=300, n_frames=F)]                   # stream 0 renders (texture + rays); the others borrow its rays
for u in range(1, U):
    st = SyntheticStream(cfg, seed=300 + u, n_frames=F, render=False, motion_scale=0.7 + 0.06 * u, rest=0.1 * u, tex_offset=(137.0 * u, 91.0 * u))
    st.tex, st.rays0, st.rays1 = streams[0].tex, streams[0].rays0, streams[0].rays1
    streams.append(st)
state0 = streams[0].torch_state(dev)
states = [state0 for _ in streams]
g = torch.Generator(device=dev); g.manual_seed(5)
base0 = torch.empty((F, U, cfg.height if hasattr(cfg, 'height') else 480, cfg.width if hasattr(cfg, 'width') else 752), dtype=torch.uint8, device=dev)
base1 = torch.empty_like(base0)
for u in range(U):
    for k in range(F):
        a, b = streams[u].frame_torch(k, states[u], g)
        base0[k, u], base1[k, u] = a, b
rep = S // U
img0 = [base0[k].repeat(rep, 1, 1).contiguous() for k in range(F)]      # [S, h, w], stream s = scene s % U
img1 = [base1[k].repeat(rep, 1, 1).contiguous() for k in range(F)]
ts = [[streams[s % U].frame_time(k) for s in range(S)] for k in range(F)]
imu_u = []                                                               # per unique stream, per frame: (t [m], w [m, 3])
for u in range(U):
    per = []
    for k in range(F):
        lo_t = streams[u].frame_time(k - 1) if k else -1e9
        hi_t = streams[u].frame_time(k)
        sel = [m for m in streams[u].imu if lo_t < m.timestamp <= hi_t]
        per.append((np.array([m.timestamp for m in sel]), np.array([m.angular_velocity for m in sel]).reshape(-1, 3), np.array([m.linear_acceleration for m in sel]).reshape(-1, 3)))
    imu_u.append(per)

def imu_rows(k, lo, hi):
    idx, tt, ww, aa = [], [], [], []
    for s in range(lo, hi):
        t, w, a = imu_u[s % U][k]
        idx.append(np.full(len(t), s - lo, np.int32)); tt.append(t); ww.append(w); aa.append(a)
    return np.concatenate(idx), np.concatenate(tt), np.concatenate(ww), np.concatenate(aa)

ROWS = {}                                                                # (k, lo, hi) -> arrays, built outside the timed loops

def run(engs, flts, parts, k0, k1, threads):
    for k in range(k0, k1):
        for lo, hi in parts:
            if (k, lo, hi) not in ROWS: ROWS[(k, lo, hi)] = imu_rows(k, lo, hi)
    def loop(e, f, lo, hi, stream, fstream):
        with torch.cuda.stream(stream):
            for k in range(k0, k1):
                i_, t_, w_, a_ = ROWS[(k, lo, hi)]
                if len(i_): e.push_imu_batch(i_, t_, w_)
                e.step(img0[k][lo:hi], img1[k][lo:hi], ts[k][lo:hi])
                if f is not None:
                    if k + 1 < F: e.prestage(img0[k + 1][lo:hi], img1[k + 1][lo:hi])
                    if len(i_): f.push_imu(i_, t_, w_, a_)
                    ms = N.current_stream()
                    with torch.cuda.stream(fstream):
                        f.submit_dev(e, np.asarray(ts[k][lo:hi], dtype=np.float64), msg_stream=ms)
            if f is not None: f.wait(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    mk = lambda: torch.cuda.Stream(device=dev)
    if threads:
        th = [threading.Thread(target=loop, args=(e, f, lo, hi, mk(), mk())) for e, f, (lo, hi) in zip(engs, flts, parts)]
        [t.start() for t in th]; [t.join() for t in th]
    else:
        for e, f, (lo, hi) in zip(engs, flts, parts): loop(e, f, lo, hi, mk(), mk())
    torch.cuda.synchronize()
    return time.perf_counter() - t0

out = {}
PRE, K = 24, 20
for name, parts, threads in (('one_engine_2048', [(0, S)], False), ('two_engines_1024_two_threads', [(0, S // 2), (S // 2, S)], True),
                             ('four_engines_512_four_threads', [(i * S // 4, (i + 1) * S // 4) for i in range(4)], True)):
    engs = [FrontendEngine(cfg, n_streams=hi - lo, device=0, inputs_persist=True) for lo, hi in parts]
    flts = [None if FE_ONLY else BatchedMSCKF(cfg, hi - lo, device=0, rows_cap=4096, max_features=engs[0].max_features) for lo, hi in parts]
    run(engs, flts, parts, 0, PRE, threads)
    dt = run(engs, flts, parts, PRE, PRE + K, threads)
    n = [int(np.mean([len(f[0]) for f in e.read_features()])) for e in engs]
    out[name] = {'ms_per_2048_stream_step': dt / K * 1e3, 'frames_per_s': S * K / dt, 'features_per_stream': n}
    if not FE_ONLY: out[name]['filter'] = [{k_: c[k_] for k_ in ('min_cam_states', 'prune_stream_steps', 'failed_streams') if k_ in c} for c in (f.counters() for f in flts)]
    for f in flts:
        if f is not None: f.close()
    for e in engs: e.close()
    del engs, flts
    torch.cuda.empty_cache()
print(json.dumps(out, indent=1))
