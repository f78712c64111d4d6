"""Randomised differential test: random scenes, ray generators, packet orders, split factors,
packet widths, treelet sizes and integral modes; hit counts, column densities and per-hit
outputs against the oracle's brute force on a subset of the rays; from round 3 on also the cached
paths (the same call three times: uncached, filling the cache, validated cache).  (The same loop
ran 3192 configurations in round 1, 4743 in round 2 and 3296 in round 3 --
profiles/recipes/fuzz_trace.py on MI355X -- without a failure; 60 fixed seeds are kept here.)"""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", range(60))
def test_random_configuration(gh, oracle, cuda, seed):
    O = oracle
    dev = cuda
    rng = np.random.default_rng(seed)
    try:
        _one(gh, O, dev, rng, seed)
    finally:
        gh.set_ray_reorder(True); gh.set_packet_split(-1); gh.set_treelet_size(-1)
        gh.set_packet_width(-1); gh.set_exact_integrals(False)


def _one(gh, O, dev, rng, seed):
    n=int(rng.choice([150, 3000, 20000, 60000, 200000]))
    mpl=int(rng.choice([1,8,32,100]))
    if n<=mpl: mpl=1
    rmax=float(rng.choice([0.002,0.02,0.08]))
    s=O.random_real4(n,(0,0,0,rmax*0.1),(1,1,1,rmax),first=int(rng.integers(0,10**6)))
    d=torch.from_numpy(s).to(dev)
    tree=gh.Tree(n,mpl,device=dev); gh.build_tree(d,tree,(0,0,0),(1,1,1))
    ss=d.cpu().numpy()
    kind=rng.choice(["iso","ortho","points","pinhole","octant"])
    nr=int(rng.choice([32,96,640,4096,20000]))
    if kind=="iso":
        o=rng.uniform(-0.2,1.2,3); rays=gh.uniform_random_rays(nr,o,float(rng.uniform(0.3,2.5)),seed=seed,device=dev)
    elif kind=="octant":
        o=rng.uniform(0.2,0.8,3); rays=gh.uniform_random_rays_single_octant(nr,o,float(rng.uniform(0.3,2.5)),int(rng.integers(0,8)),seed=seed,device=dev)
    elif kind=="points":
        pts=torch.from_numpy(rng.uniform(0,1,(nr,3)).astype(np.float32)).to(dev)
        rays=gh.one_to_many_rays(rng.uniform(0,1,3),pts,int(rng.integers(0,3)))
    elif kind=="pinhole":
        side=int(math.sqrt(nr)//8*8) or 8
        rays=gh.pinhole_camera_rays(side,side,rng.uniform(1.2,2.5,3),(0.5,0.5,0.5),(0,0,1),float(rng.uniform(0.2,1.2)),float(rng.uniform(1.0,4.0)),device=dev)
    else:
        side=int(math.sqrt(nr)//8*8) or 8
        ax=int(rng.integers(0,3)); sense=1 if rng.random()<0.5 else -1
        u,v=np.meshgrid((np.arange(side)+0.5)/side,(np.arange(side)+0.5)/side)
        r=np.zeros((side*side,7),np.float32); perp=[k for k in range(3) if k!=ax]
        r[:,ax]=sense; r[:,3+perp[0]]=u.ravel(); r[:,3+perp[1]]=v.ravel()
        start=rng.uniform(-0.1,0.6,len(r)).astype(np.float32) if rng.random()<0.5 else np.full(len(r),-0.1,np.float32)
        r[:,3+ax]=start if sense>0 else (1-start)
        r[:,6]=rng.uniform(0.1,1.3,len(r)).astype(np.float32)
        rays=torch.from_numpy(r).to(dev)
    R=len(rays)
    if R%32: rays=rays[:R//32*32].contiguous(); R=len(rays)
    if R==0: return
    gh.set_ray_reorder(bool(rng.random()<0.7)); gh.set_packet_split(int(rng.choice([-1,-1,1,2,4,8]))); gh.set_treelet_size(int(rng.choice([-1,-1,0,64,512]))); gh.set_packet_width(int(rng.choice([-1,-1,-1,64,32,16])))
    exact=bool(rng.random()<0.5); gh.set_exact_integrals(exact)
    hc=torch.empty(R,dtype=torch.int32,device=dev); cu=torch.empty(R,dtype=torch.float32,device=dev)
    gh.trace_hitcounts_sph(rays,d,tree,hc); gh.trace_cumulative_sph(rays,d,tree,cu)
    offs,idx,w,dist=gh.trace_sph(rays,d,tree)
    # the same call twice more: the second derives its records into the library's cache, the third
    # runs on cached records validated by signature (round 3) -- the same bits every time
    cu2=torch.empty_like(cu); cu3=torch.empty_like(cu)
    gh.trace_cumulative_sph(rays,d,tree,cu2); gh.trace_cumulative_sph(rays,d,tree,cu3)
    gh.trace_status()
    assert torch.equal(cu,cu2) and torch.equal(cu,cu3), dict(n=n, kind=str(kind), R=R, cached="differs")
    sub=np.unique(rng.integers(0,R,min(R,160)))
    rr=rays.cpu().numpy()
    ref_c=O.brute_hitcounts(rr[sub],ss); c32,c64=O.brute_cumulative(rr[sub],ss)
    ro,ri,rw,rd=O.brute_hits(rr[sub],ss)
    hcn=hc.cpu().numpy(); cun=cu.cpu().numpy(); offn=offs.cpu().numpy(); idn=idx.cpu().numpy(); wn=w.cpu().numpy(); dn=dist.cpu().numpy()
    ok=np.array_equal(hcn[sub],ref_c)
    ok&=np.array_equal(offn,np.concatenate([[0],np.cumsum(hcn)[:-1]]))
    term=1.91/ (rmax*0.1)**2
    err=np.abs(cun[sub]-c64)
    ok&=bool(np.all(err<=1e-5*np.abs(c64)+2e-6*term))
    if exact: ok&=np.array_equal(cun[sub].view(np.uint32),c32.view(np.uint32))
    for k,r_ in enumerate(sub):
        a=offn[r_]; b=a+hcn[r_]; ra=ro[k]; rb=ro[k+1] if k+1<len(sub) else len(ri)
        if not (b-a==rb-ra and np.array_equal(idn[a:b],ri[ra:rb]) and np.array_equal(wn[a:b].view(np.uint32),rw[ra:rb].view(np.uint32)) and np.array_equal(dn[a:b].view(np.uint32),rd[ra:rb].view(np.uint32))):
            ok=False; break
    assert ok, dict(n=n, mpl=mpl, rmax=rmax, kind=str(kind), R=R, exact=exact)


@pytest.mark.parametrize("seed", range(1000, 1120))
def test_random_build_is_the_oracle_tree(gh, oracle, cuda, seed):
    """Random sizes (2 ... 120 000), max_per_leaf (1 ... 400: all three leaf-head code paths and
    the sparse-table path's boundaries 3/4/5, 31/32/33, 63/64/65),
    duplicate / clustered centres, Euclidean and XOR deltas: nodes, leaves and root identical to
    the oracle's sequential restatement.  (96 298 such builds ran once without a difference.)"""
    O = oracle
    rng = np.random.default_rng(seed)
    n = int(rng.choice([2, 3, 5, 17, 64, 65, 257, 1000, 4097, 30000, 120000]))
    mpl = int(rng.choice([1, 2, 3, 4, 5, 7, 16, 31, 32, 33, 63, 64, 65, 100, 256, 257, 400]))
    if n <= mpl:
        mpl = max(1, n - 1)
    s = O.random_real4(n, (0, 0, 0, 0), (1, 1, 1, 0.05), first=int(rng.integers(0, 10**6)))
    if rng.random() < 0.4:
        k = max(1, n // int(rng.choice([2, 10, 100])))
        s[:, :3] = s[rng.integers(0, k, n), :3]
    use_xor = bool(rng.random() < 0.5)
    keys = O.morton_keys30(s, (0, 0, 0), (1, 1, 1))
    keys_s, ss, _ = O.sort_by_key(keys, s)
    ss = np.ascontiguousarray(ss)
    dl = O.deltas_xor(keys_s) if use_xor else O.deltas_euclid(ss)
    nodes, leaves, root, _ = O.albvh(ss, dl, mpl)
    d = torch.from_numpy(ss).to(cuda)
    tree = gh.Tree(n, mpl, device=cuda)
    if use_xor:
        dk = torch.from_numpy(keys_s.view(np.int32)).to(cuda)
        dx = torch.empty(n + 1, dtype=torch.int32, device=cuda)
        gh.XOR_deltas_sph(dk, dx)
        gh.ALBVH_sph(d, dx, tree)
    else:
        df = torch.empty(n + 1, dtype=torch.float32, device=cuda)
        gh.euclidean_deltas_sph(d, df)
        gh.ALBVH_sph(d, df, tree)
    assert np.array_equal(tree.leaves.cpu().numpy(), leaves)
    assert np.array_equal(tree.nodes.cpu().numpy(), nodes)
    assert int(tree.root_index.item()) == root


@pytest.mark.parametrize("seed", range(5000, 5030))
def test_random_triangle_scene(gh, oracle, cuda, seed):
    """Random height-field meshes (some flat: inflated boxes), pinhole / orthographic / isotropic
    rays, packet orders and widths: closest triangle per ray == brute force over all triangles.
    (3589 such configurations ran once without a difference.)"""
    from test_gpu_triangles import heightfield_mesh
    rng = np.random.default_rng(seed)
    gx = int(rng.choice([8, 40, 128, 300])); gy = int(rng.choice([8, 64, 200]))
    tris = heightfield_mesh(gx, gy, seed=int(rng.integers(0, 1000)), flat=bool(rng.random() < 0.2))
    mpl = int(rng.choice([1, 8, 32]))
    if len(tris) <= mpl:
        mpl = 1
    d = torch.from_numpy(tris).to(cuda)
    tree = gh.Tree(len(tris), mpl, device=cuda)
    gh.build_tree_tris(d, tree)
    side = int(rng.choice([8, 32, 96]))
    kind = rng.choice(["pinhole", "ortho", "iso"])
    if kind == "pinhole":
        cam = rng.uniform(-0.5, 1.5, 3); cam[2] = rng.uniform(0.05, 2.0)
        rays = gh.pinhole_camera_rays(side, side, cam, rng.uniform(0.2, 0.8, 3) * np.array([1, 1, 0]),
                                      (0, 1, 0.1), float(rng.uniform(0.2, 1.4)),
                                      float(rng.uniform(0.5, 5)), device=cuda)
    elif kind == "ortho":
        rays = gh.orthographic_projection_rays(side, side, (rng.uniform(0, 1), rng.uniform(0, 1), 1.0),
                                               (0.5, 0.5, 0.0), (0, 1, 0), float(rng.uniform(0.3, 1.5)),
                                               3.0, device=cuda)
    else:
        rays = gh.uniform_random_rays(side * side, (rng.uniform(0, 1), rng.uniform(0, 1), rng.uniform(0.2, 1)),
                                      3.0, seed=seed, device=cuda)
    try:
        gh.set_ray_reorder(bool(rng.random() < 0.7)); gh.set_packet_width(int(rng.choice([-1, 64, 32, 16])))
        cl = torch.empty(len(rays), dtype=torch.int32, device=cuda)
        gh.trace_closest_tri(rays, d, tree, cl)
        gh.trace_status()
    finally:
        gh.set_ray_reorder(True); gh.set_packet_width(-1)
    sub = np.unique(rng.integers(0, len(rays), min(len(rays), 200)))
    ref, _ = oracle.brute_closest_tri(rays.cpu().numpy()[sub], d.cpu().numpy())
    assert np.array_equal(cl.cpu().numpy()[sub], ref)
