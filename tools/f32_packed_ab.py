"""fp32 flat search, small batches (1 .. 15 queries: the stream kernels' private rings): fp32 rows split on the fly ("flat_f32_packed" 0) against the
bf16 operand copy's first terms (1, round 6); lists and distance bits compared with the exact kernels."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch, cvt_amd
from cvt_amd import synth
dev = torch.device("cuda", 0)
k = 100
for D in (128, 64, 256, 96):
    n = int(0.5 * 2**30 / (4 * D))
    x = synth.sift_like(n, D, device=dev) if D == 128 else torch.nn.functional.normalize(torch.randn((n, D), device=dev), dim=1)
    for metric in (1, 0):
        ix = cvt_amd.FlatIndex(metric, D); ix.add(x)
        for nq in (1, 4, 8, 15):
            q = (synth.sift_like(nq, D, seed=0xBEEF, device=dev) if D == 128 else torch.nn.functional.normalize(torch.randn((nq, D), device=dev), dim=1)).contiguous()
            cvt_amd.set_tuning("flat_variant", 1); de, ie = ix.search(q, k); cvt_amd.set_tuning("flat_variant", 0)
            out = []
            for pk in (0, 1):
                cvt_amd.set_tuning("flat_f32_packed", pk)
                for _ in range(5): ix.search(q, k)
                torch.cuda.synchronize(); t0 = time.perf_counter(); reps = 50
                for _ in range(reps): d, i = ix.search(q, k)
                torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / reps * 1e3
                same = bool(torch.equal(i, ie) and torch.equal(d.view(torch.int32), de.view(torch.int32)))
                out.append("packed %d: %.4f ms path %d identical=%s" % (pk, ms, ix.last_search()[0], same))
            print("D %d metric %d nq %d: %s" % (D, metric, nq, "; ".join(out)), flush=True)
        ix.close()
cvt_amd.set_tuning("flat_f32_packed", 1)
