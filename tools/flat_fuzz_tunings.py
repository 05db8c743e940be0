"""tools/flat_fuzz_ab.py under every tuning key that changes a route or a kernel form of the flat search (80 random shapes each)"""
import os, sys, importlib.util
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import cvt_amd as amd
spec = importlib.util.spec_from_file_location("f", os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools", "flat_fuzz_ab.py"))
mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
for key, val, dflt in (("flat_u8_tfilter_chunks", 1, 4), ("flat_u8_tfilter_chunks", 2, 4), ("flat_u8_tfilter_sample", 2, 0), ("flat_u8_tfilter_sample", 64, 0),
                       ("flat_f32_tfilter_sample", 2, 0), ("flat_f32_tfilter_sample", 32, 0), ("flat_f32_rows_copy", 0, 4), ("flat_f32_tfilter_wide_band", 0, 1),
                       ("flat_f32_tfilter_retry", 1, 0), ("flat_f32_packed", 0, 1)):
    amd.set_tuning(key, val)
    bad, paths = mod.run(80, 100 + val, verbose=True)
    amd.set_tuning(key, dflt)
    print(key, val, "different", bad, paths, flush=True)
