"""Header and code in step (CPU only)."""
import os


def test_every_tuning_key_is_documented():
    """cvtmi_set_tuning's keys (csrc/api.hip) and their description in include/cvtmi.h stay in step: a key the header does not name is a
    switch nobody can find."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    api = open(os.path.join(root, "cvt_amd", "csrc", "api.hip")).read()
    hdr = open(os.path.join(root, "include", "cvtmi.h")).read()
    i = api.index("int cvtmi_set_tuning(")
    body = api[i:api.index("\n}\n", i)]
    keys = re.findall(r'strcmp\(name, "([a-z0-9_]+)"\)', body)
    assert len(keys) > 30
    missing = [k for k in keys if '"%s"' % k not in hdr]
    assert not missing, missing
