#!/usr/bin/env python3
"""CPU soak of the candidate table (cat_sim.hip build_grids, rules 1 - 3): for random origins -- a third of them within a few px of a
wall bb edge -- and every ray, the oracle's sequential wall query over the walls the table lists must equal the query over all walls,
bit for bit.  3.4 M rays on agh-map / labyrinth / lbirinth at 4-px cells: 0 mismatches (round 3).  Usage: python tools/table_soak.py"""
import sys, ctypes as C, numpy as np
from pathlib import Path; sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from tests.test_spatial_grid import _grid
from as_cops_and_thieves_amd.config import SimConfig
from as_cops_and_thieves_amd.maps import load_preset
from oracle.cat_oracle import OracleSim
tot=0
for name,rays,cell in [("agh-map",64,4),("labyrinth",64,4),("agh-map",90,4),("lbirinth",64,4)]:
    cmap=load_preset(name).compile(); cfg=SimConfig(n_envs=1,n_rays=rays)
    L,h,rdx,rdy=_grid(cmap,cfg,cell); orc=OracleSim(cfg,[cmap])
    rng=np.random.default_rng(99); out=(C.c_int*256)()
    lo=cmap.shape_bb[:,:2].min(0)-30; hi=cmap.shape_bb[:,2:].max(0)+30
    bad=0
    for trial in range(12000):
        ax,ay=rng.uniform(lo,hi)
        if trial%3==0:
            s_=rng.integers(cmap.n_shapes); ax=cmap.shape_bb[s_,rng.choice([0,2])]+rng.normal(0,2.0); ay=rng.uniform(cmap.shape_bb[s_,1]-3,cmap.shape_bb[s_,3]+3)
        for k in range(rays):
            b=(ax+rdx[k],ay+rdy[k])
            n=L.cat_grid_lookup_host(h,float(ax),float(ay),int(k),out,256)
            full=orc.segment_query(0,-1,(ax,ay),b,cfg.ray_radius,los=True)
            part=orc.segment_query(0,-1,(ax,ay),b,cfg.ray_radius,los=True,walls=list(out[:n]))
            tot+=1
            if (full[0],np.float64(full[1]).tobytes(),full[2])!=(part[0],np.float64(part[1]).tobytes(),part[2]):
                bad+=1; print("MISMATCH",name,ax,ay,k,full,part,list(out[:n]),flush=True)
    print(name,rays,cell,"done, mismatches",bad,"total so far",tot,flush=True)
    L.cat_grid_free_host(h)
