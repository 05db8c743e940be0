import numpy as np, subprocess, time, os, sys
rng=np.random.default_rng(5); n,D=1000000,128
cen = rng.normal(size=(1000, D)).astype(np.float32)
x = cen[rng.integers(0, 1000, n)] + 0.6 * rng.normal(size=(n, D)).astype(np.float32)
x /= np.linalg.norm(x, axis=1, keepdims=True)
x.tofile('/tmp/h_rows.bin')
for th in sys.argv[1:]:
    cmd=['cvt_amd/bin/hnsw_build','/tmp/h_rows.bin',str(D),'32','80','/tmp/h.idx','ip','-',th]
    t=time.time(); r=subprocess.run(cmd,check=True,capture_output=True,text=True); print(th, round(time.time()-t,1),'s', r.stdout.strip(), flush=True)
