import sys, ctypes, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from conftest import pkg
from helpers import patch_sim, seeded_fields
capi = pkg("_capi")
hip = capi.load_hip_library()
ora = capi.bind(ctypes.CDLL("/root/repo/oracle/libfdtd_oracle.so"))
shape = tuple(int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (53, 47, 31)
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 1
for use_classes in (False,):
    res = []
    for lib in (hip, ora):
        s = patch_sim(*shape, boundary="CPML", cpml_cells=8, nr_ts=300, use_classes=use_classes)
        e = s.build(lib); seeded_fields(e, 1)
        for st in range(steps):
            e.half_step(0)
            res.append(e.fields().copy())
            e.half_step(1)
            res.append(e.fields().copy())
    n = len(res)//2
    for q in range(n):
        a, b = res[q], res[n+q]
        for kind in range(2):
            for c in range(3):
                d = a[kind, c].view(np.uint32) != b[kind, c].view(np.uint32)
                if d.any():
                    idx = np.argwhere(d)
                    print(f"half {q} kind {kind} comp {c}: {d.sum()} diffs; k range {idx[:,0].min()}..{idx[:,0].max()} j {idx[:,1].min()}..{idx[:,1].max()} i {idx[:,2].min()}..{idx[:,2].max()}; first {idx[:5].tolist()}")
                    k,j,i = idx[0]
                    print("   hip", a[kind,c,k,j,i], "ora", b[kind,c,k,j,i])
    print("cpml slots x", s.cpml.slot[0].tolist())
