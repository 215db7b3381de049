"""CPU: the entropy front-end (zj_jpeg.cpp) built alone under AddressSanitizer + UBSan and fed mutated JPEG files
(tools/fuzz_frontend.py: bit flips, truncations, splices, deletions, injected markers; serial and restart-threaded
decoders).  Any status is acceptable, a sanitizer report is not."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_damaged_files_never_corrupt_memory():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_frontend.py"), "--iters", "1500"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "sanitizer-clean" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
