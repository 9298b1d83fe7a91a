import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def _keep_the_heap_mapped():
    """A precaution of the TEST PROCESS, not of the library.  Three of ~36 runs of the GPU suite on the round-6 library died of
    `Memory access fault by GPU node-2 on address 0x58c5..e000` -- a page of the HOST heap -- inside the first host-scalar MSM above
    1 MiB that followed the IPA / KZG / Ligero files (profiles/EXPERIMENTS.md section 00).  The library's own use of
    hipHostRegister (an option of the Ligero slabs) was removed for it.  The other candidate is the HIP runtime itself: it pins the
    caller's pages for pageable copies above 1 MiB and keeps those pins for a while, and glibc gives the top of the heap back to the
    kernel when large arrays are freed -- a later array at the same addresses then meets a pin whose pages are gone.  The same MSM
    tests looped 117 times in one process, where the heap never shrinks, did not fault.  So: arrays up to 32 MB (glibc's maximum for
    the threshold) come from the heap, and the heap is never trimmed."""
    try:
        import ctypes
        libc = ctypes.CDLL("libc.so.6")
        libc.mallopt(-3, 32 << 20)          # M_MMAP_THRESHOLD
        libc.mallopt(-1, 0x7fffffff)        # M_TRIM_THRESHOLD
    except Exception:
        pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    _keep_the_heap_mapped()


@pytest.fixture(scope="session")
def ctx():
    import poly_commit_amd as pc
    c = pc.Context(0)
    yield c
    c.close()
