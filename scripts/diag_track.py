import sys, numpy as np
sys.path.insert(0, '.')
from densemonoslam_amd import odometry as dms
from oracle import orc
from tests import helpers
z = np.load('tests/golden/gputest_pair.npz')
pair = dict(rgb1=z['rgb1'], rgb2=z['rgb2'], depth1_raw=z['depth1'], depth2=(z['depth2']//5).astype(np.uint16), K=(528.,528.,320.,240.))
def fresh():
    K = pair['K']
    verts, norms = helpers.gputest_model_maps(pair['depth1_raw'], K)
    g = dms.RGBDOdometry(640,480,K[2],K[3],K[0],K[1]); o = orc.Odometry(640,480,K[2],K[3],K[0],K[1])
    for t in (g,o):
        t.initICPModel(verts,norms,20.0,np.eye(4,dtype=np.float32)); t.initRGBModel(helpers.rgba(pair['rgb1']))
        t.initICP(pair['depth2'],20.0); t.initRGB(helpers.rgba(pair['rgb2'])); t.initFirstRGB(helpers.rgba(pair['rgb1']))
    return g,o
for cfg in [dict(rgbOnly=False,icpWeight=10.,pyramid=True,fastOdom=False,so3=True),
            dict(rgbOnly=False,icpWeight=10.,pyramid=True,fastOdom=False,so3=False),
            dict(rgbOnly=False,icpWeight=100.,pyramid=True,fastOdom=False,so3=True),
            dict(rgbOnly=False,icpWeight=100.,pyramid=True,fastOdom=False,so3=False),
            dict(rgbOnly=True,icpWeight=10.,pyramid=True,fastOdom=False,so3=False),
            dict(rgbOnly=False,icpWeight=10.,pyramid=False,fastOdom=True,so3=False)]:
    g,o = fresh()
    tg,Rg,rg = g.getIncrementalTransformation(np.zeros(3),np.eye(3),**cfg)
    to,Ro,ro = o.getIncrementalTransformation(np.zeros(3),np.eye(3),**cfg)
    print(cfg)
    print('  dt %.3e m  dR %.3e deg | iters g %s o %s so3 %d/%d | so3err %.6g/%.6g cnt %g/%g | icp %.6g/%.6g %g/%g | rgb %.6g/%.6g %g/%g' % (
        np.linalg.norm(tg-to), helpers.rot_angle_deg(Rg,Ro), list(rg.iterations_run), list(ro.iterations_run), rg.so3_iterations_run, ro.so3_iterations_run,
        rg.lastSO3Error, ro.lastSO3Error, rg.lastSO3Count, ro.lastSO3Count, rg.lastICPError, ro.lastICPError, rg.lastICPCount, ro.lastICPCount,
        rg.lastRGBError, ro.lastRGBError, rg.lastRGBCount, ro.lastRGBCount))
    A_g=np.array(rg.lastA).reshape(6,6); A_o=np.array(ro.lastA).reshape(6,6)
    print('  lastA rel diff %.3e  cond %.3e' % (np.abs(A_g-A_o).max()/np.abs(A_o).max(), np.linalg.cond(A_o)))
