import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpc_collisionavoidance_amd import scenario
from oracle import binding as ob
name="usv_model_pf_ca"; B=1024; N,K=40,10; dt=scenario.DT[name]
wl = scenario.make_batch(name, N, K, B, dt=dt, seed=1234)
spec = ob.spec(2, N, N*dt, K)
x,u = wl["x_init"].copy(), wl["u_init"].copy(); x0 = wl["x0"].copy()
first_fail = -np.ones(B,int)
for t in range(20):
    xp, up, x0p = x.copy(), u.copy(), x0.copy()
    st,it = ob.rti_batch(spec, x,u,x0, wl["yref"],wl["yref_e"],wl["p"],wl["lh"], threads=8)
    newf = np.where((st!=0)&(first_fail<0))[0]
    for b in newf:
        first_fail[b]=t
        r = ob.rti(spec, xp[b], up[b], x0p[b], wl["yref"][b], wl["yref_e"][b], wl["p"][b], wl["lh"][b])
        # geometry: min clearance of iterate to obstacles
        pos = xp[b][:, 10:12]; obs = wl["p"][b][0].reshape(K,2); lh = wl["lh"][b][0]
        d = np.sqrt(((pos[:,None,:]-obs[None])**2).sum(-1)) - lh[None]
        print("tick",t,"inst",b,"qp_status",r["qp_status"],"iter",r["qp_iter"],"res",r["res"], "min clear %.4f at stage %d obs %d"%(d.min(), *np.unravel_index(d.argmin(), d.shape)),
              "u=%.3f v=%.4f r=%.3f T=(%.1f,%.1f)"%(x0p[b][3],x0p[b][4],x0p[b][5],x0p[b][12],x0p[b][13]), "max|v| iter %.3f"%np.abs(xp[b][:,4]).max())
    x0 = x[:,1].copy()
