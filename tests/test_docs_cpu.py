"""Documentation that has to follow the code: every environment variable the library reads is listed in INTEGRATION.md (section 4a)."""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_environment_variable_of_the_library_is_documented():
    names = set()
    for f in glob.glob(os.path.join(ROOT, "poly_commit_amd", "csrc", "*.h*")):
        names |= set(re.findall(r'getenv\("([A-Z0-9_]+)"\)', open(f).read()))
    assert len(names) > 20                                                # (the scan itself works)
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    missing = sorted(n for n in names if "`" + n + "`" not in doc)
    assert not missing, "not in INTEGRATION.md section 4a: %s" % missing
