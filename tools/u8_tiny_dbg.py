import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, cvt_amd
dev = torch.device("cuda", 0)
D = 512
g = torch.Generator(device=dev); g.manual_seed(5)
n = 10_000_000
x = torch.randint(0, 256, (n, D), generator=g, device=dev, dtype=torch.uint8)
q = torch.randint(0, 256, (4, D), generator=g, device=dev, dtype=torch.uint8)
truth = torch.empty((4, n), dtype=torch.int64, device=dev)
for a in range(0, n, 1 << 20):
    for j in range(4):
        truth[j, a:a + (1 << 20)] = ((x[a:a + (1 << 20)].to(torch.int32) - q[j:j+1].to(torch.int32)) ** 2).sum(1)
td, ti = torch.sort(truth, dim=1, stable=True)
for chunk in (1 << 21,):
    ix = cvt_amd.FlatIndex(2, D)
    for a in range(0, n, chunk):
        ix.add(x[a:a + chunk])
    for nq in (1, 2, 4):
        for v in (0, 1):
            cvt_amd.set_tuning("flat_variant", v)
            for rep in range(3):
                d, i = ix.search(q[:nq].contiguous(), 10)
                d, i = torch.as_tensor(d), torch.as_tensor(i)
                ok = torch.equal(i.cpu(), ti[:nq, :10].cpu()) and torch.equal(d.cpu().double(), td[:nq, :10].cpu().double())
                print("chunk", chunk, "nq", nq, "variant", v, "rep", rep, "ok", ok)
                if not ok:
                    for j in range(nq):
                        print("    got", i[j].tolist(), d[j].tolist()); print("  truth", ti[j, :10].tolist(), td[j, :10].tolist())
    ix.close()
cvt_amd.set_tuning("flat_variant", 0)
