"""CPU: the front-end's threaded paths under ThreadSanitizer (round 6): the decoder's helper threads (zj_crew.h), restart
segments on several threads, and a scan without restart markers entered at one point per thread (scan_baseline_parallel) --
tests/tsan/par_scan_tsan.cpp, built from zj_jpeg.cpp + tests/fuzz/jpeg_stubs.cpp with g++ -fsanitize=thread.  The harness
compares every threaded decode with the one-thread decode; a data race report ends it with a non-zero exit."""
import io
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _jpeg(path, rng, w, h, sub, q, restart_rows, flat):
    from PIL import Image, ImageFile
    ImageFile.MAXBLOCK = max(ImageFile.MAXBLOCK, 1 << 24)
    small = rng.integers(0, 256, (h // 16, w // 16, 3), dtype=np.uint8)
    img = np.asarray(Image.fromarray(small, "RGB").resize((w, h), Image.BICUBIC)).astype(np.int16)
    img = img + rng.integers(-20, 21, (h, w, 3), dtype=np.int16)
    if flat:
        img[h // 3:2 * h // 3] = (90, 140, 200)   # identical two-symbol MCUs: where the stitching gives up
    b = io.BytesIO()
    kw = {"restart_marker_rows": restart_rows} if restart_rows else {}
    Image.fromarray(np.clip(img, 0, 255).astype(np.uint8), "RGB").save(b, "JPEG", quality=q, subsampling=sub, **kw)
    with open(path, "wb") as f:
        f.write(b.getvalue())


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_threaded_front_end_is_race_free(tmp_path):
    exe = str(tmp_path / "par_scan_tsan")
    build = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-pthread", "-fsanitize=thread", "-fno-omit-frame-pointer", "-o", exe,
                            os.path.join(ROOT, "tests", "tsan", "par_scan_tsan.cpp"),
                            os.path.join(ROOT, "zune-jpeg_amd", "csrc", "zj_jpeg.cpp"),
                            os.path.join(ROOT, "tests", "fuzz", "jpeg_stubs.cpp")], capture_output=True, text=True)
    if build.returncode != 0 and "tsan" in build.stderr.lower():
        pytest.skip("this g++ has no ThreadSanitizer runtime")
    assert build.returncode == 0, build.stderr[-2000:]
    rng = np.random.default_rng(5)
    files = []
    for i, (w, h, sub, q, rst, flat) in enumerate([(640, 480, 2, 90, 0, False), (512, 640, 0, 85, 0, True), (640, 480, 1, 92, 1, False)]):
        files.append(str(tmp_path / f"t{i}.jpg"))
        _jpeg(files[-1], rng, w, h, sub, q, rst, flat)
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1")
    for k in ("ZJ_PAR_SCAN", "ZJ_WALKER_V1", "ZJ_PAR_PATIENCE"):
        env.pop(k, None)
    run = subprocess.run([exe] + files, capture_output=True, text=True, env=env, timeout=600)
    assert run.returncode == 0, (run.stdout[-1000:], run.stderr[-3000:])
    assert "equal to one thread" in run.stdout
