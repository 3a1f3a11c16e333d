import sys, time; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/oracle')
import numpy as np, mpc_code_amd as m
from mpc_code_amd import capi
import oracle_c
p = m.load_problem(m.example_path('cstr_lmpc.py'))
s = capi.Solver(p); print(s.build_info())
oc = oracle_c.OracleC(p)
rng = np.random.default_rng(20250614)
B=300
x0 = rng.uniform([-0.5,-8,-5],[0.5,8,5],size=(B,3))
xs = rng.uniform(-0.2,0.2,size=(B,3)); us = rng.uniform(-1,1,size=(B,2)); d = rng.uniform(-0.1,0.1,size=(B,3))
r = s.ocp_solve(x0, xs, us, d, us, want_w=True); q = oc.ocp_solve(x0, xs, us, d, us, want_w=True)
print('ocp status eq', np.array_equal(r['status'],q['status']), np.bincount(r['status']), 'iters eq', (r['iters']==q['iters']).mean())
ok = q['status']!=2
print('u0 err', np.abs(r['u0']-q['u0'])[ok].max(), 'x1 err', np.abs(r['x1']-q['x1'])[ok].max(), 'w err', np.nanmax(np.abs(r['w']-q['w'])[ok]))
print('kernel ms', s.last_kernel_ms())
t = s.target_solve(np.zeros(2), np.array([0.2,0,0]), np.zeros(3), d*30, us); t2 = oc.target_solve(np.zeros(2), np.array([0.2,0,0]), np.zeros(3), d*30, us)
print('target', np.abs(t['xs']-t2['xs']).max(), np.abs(t['us']-t2['us']).max(), np.array_equal(t['status'],t2['status']), np.bincount(t['status']), (t['iters']==t2['iters']).mean())
ne=6
xi = rng.normal(size=(B,6)); Pk = np.broadcast_to(p.P0,(B,6,6)).copy()+ 1e-3*np.eye(6); y = rng.normal(size=(B,3))
yhat = xi[:,:3]@p.C.T + xi[:,3:]@p.Cd.T
a1,b1 = s.kf_update(y, xi, Pk); a2,b2 = oc.kf_update(y, yhat, xi, Pk)
print('kf', np.abs(a1-a2).max(), np.abs(b1-b2.reshape(B,6,6)).max())
# closed loop
nst=30
s.loop_alloc(B, nst, capi.LOG_ALL); s.loop_set_state(x0, x0); s.loop_set_schedule(p.schedules(nst))
t0=time.time(); s.loop_run(0, nst); s.loop_sync(); print('loop wall', time.time()-t0, s.last_kernel_ms())
lc = oc.closed_loop(nst, x0, x0)
for k in ('U','X_HAT','XS','US','YS','Xp','D_HAT'):
    print(k, np.abs(s.loop_get_log(k)-lc[k]).max())
for k in ('STATUS_DYN','STATUS_SS','ITERS_DYN','ITERS_SS'):
    print(k, (s.loop_get_log(k)==lc[k]).mean())
