"""Import alias.  The product package lives in the directory ``poly-commit_amd/`` (the
hyphen is part of the project name and is not importable); this shim points the importable
name ``poly_commit_amd`` at it."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "poly-commit_amd")]
with open(_os.path.join(__path__[0], "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(__path__[0], "__init__.py"), "exec"))
