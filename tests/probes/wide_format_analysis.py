"""Host-side, no GPU: expected wide-node visits per random line (surface-area metric) for regroupings of the reference tree of
the synthetic triangle soup -- width W, B-bit child boxes on a per-node power-of-two grid (rounded outward), leaves = the reference's
leaves -- times the 16-byte pieces a node of that format costs to fetch.  The walk of a big tree is bound by pieces per ray
(DESIGN.md section 5): this is how node formats are priced BEFORE one is built.
    python tests/probes/wide_format_analysis.py [n_triangles]"""
import sys, importlib, heapq
import numpy as np
import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
pkg=importlib.import_module("raytracing-rust_amd"); hb=importlib.import_module("raytracing-rust_amd.hip_backend"); abi=pkg.abi
import scenes
n=int(sys.argv[1]) if len(sys.argv)>1 else 100000
g=hb.HipScene(scenes.random_triangle_mesh(n, seed=42, extent=10.0), device=abi.RT_DEVICE_NONE)
nodes=g.nodes()
mn=nodes['min'].astype(np.float64); mx=nodes['max'].astype(np.float64); ch=nodes['children']
def area(lo,hi):
    d=hi-lo; return d[0]*d[1]+d[1]*d[2]+d[2]*d[0]
root_area=area(mn[0],mx[0])
def analyse(W,B):
    q=(1<<B)-1
    visits=0.0; leaf_visits=0.0; n_wide=0
    stack=[(0, mn[0], mx[0])]   # (host node, conservative box as seen by the walk)
    while stack:
        h, clo, chi = stack.pop()
        n_wide+=1
        visits += area(clo,chi)/root_area     # P(line hits the conservative box) ~ visits of this node
        kids=[int(ch[h][0]), int(ch[h][1])]
        while len(kids)<W:
            best=-1; ba=-1
            for i,k in enumerate(kids):
                if ch[k][0]>=0:
                    a=area(mn[k],mx[k])
                    if a>ba: ba=a; best=i
            if best<0: break
            k=kids[best]; kids[best:best+1]=[int(ch[k][0]), int(ch[k][1])]
        # grid of this node: origin = exact min of the node, step = smallest power of two covering extent/q
        ext=mx[h]-mn[h]
        step=np.where(ext>0, 2.0**np.ceil(np.log2(np.maximum(ext,1e-300)/q)), 2.0**-126)
        for k in kids:
            lo=mn[h]+np.floor((mn[k]-mn[h])/step)*step
            hi=mn[h]+np.ceil((mx[k]-mn[h])/step)*step
            if ch[k][0]>=0:
                stack.append((k,lo,hi))
            else:
                leaf_visits += area(lo,hi)/root_area
    return visits, leaf_visits, n_wide
for W,B,pieces in ((2,32,4),(4,8,3),(4,8,4),(6,5,3),(5,6,3),(8,8,5),(6,8,4)):
    v,l,nw=analyse(W,B)
    print(f"W={W} B={B}: wide nodes {nw}, expected node visits per line {v:.2f}, leaf-box visits {l:.2f}, node pieces per line {v*pieces:.1f} (at {pieces} pieces per step)")
