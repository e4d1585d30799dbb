import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, ctypes as C
from oracle import cvops, frontend as ofe
from uav_airvision_amd.config import ConfigEuRoC
from uav_airvision_amd.synth import SyntheticStream, replay
lib=cvops.lib()
cfg=ConfigEuRoC()
st=SyntheticStream(cfg, seed=3, n_frames=8, motion_scale=1.5)
orig=cvops.calc_optical_flow_pyr_lk
stats=[]
def wrapped(img0,img1,p0,p1,**kw):
    n=len(p0)
    tap=np.zeros((n,8),np.int32)
    lib.orc_lk_set_iter_tap(tap.ctypes.data_as(C.POINTER(C.c_int)))
    r=orig(img0,img1,p0,p1,**kw)
    lib.orc_lk_set_iter_tap(None)
    stats.append(tap[:, :4].copy())
    return r
cvops.calc_optical_flow_pyr_lk=wrapped
fe=ofe.OracleFrontend(cfg)
replay(st, [fe.imu_callback], lambda m: fe.stereo_callback(m))
tot_pts=0; sum_iter=0; sum_wave=0; hist=np.zeros(32,int)
for tap in stats:
    n=len(tap)
    if n==0: continue
    tot_pts+=n
    for lev in range(4):
        it=tap[:,lev]
        hist+=np.bincount(np.minimum(it,31),minlength=32)
        sum_iter+=it.sum()
        pad=(-n)%4
        w=np.concatenate([it,np.zeros(pad,int)]).reshape(-1,4)
        sum_wave+=w.max(1).sum()*4
print('calls',len(stats),'points',tot_pts)
print('mean iterations per point-level %.2f' % (sum_iter/(4*tot_pts)))
print('wave cost (max over 4 consecutive points) / ideal (sum): %.3f' % (sum_wave/sum_iter))
print('hist', hist[:32])
sw=0
for tap in stats:
    n=len(tap)
    for lev in range(4):
        s=np.sort(tap[:,lev]); pad=(-n)%4
        sw+=np.concatenate([np.zeros(pad,int),s]).reshape(-1,4).max(1).sum()*4
print('with per-level perfect sorting: %.3f' % (sw/sum_iter))
sw2=0
for tap in stats:
    n=len(tap); o=np.argsort(tap.sum(1)); pad=(-n)%4
    for lev in range(4):
        s=tap[o,lev]
        sw2+=np.concatenate([np.zeros(pad,int),s]).reshape(-1,4).max(1).sum()*4
print("sorted by the point's own total (oracle knowledge): %.3f" % (sw2/sum_iter))
