"""GPU: the C99 examples (examples/*.c) built against the in-tree libzjhip.so and RUN -- the C ABI from the language a
maintainer of the reference would bind it from (cgo-style: plain pointers and sizes, include/zjhip.h only)."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build(name, tmp_path):
    out = str(tmp_path / name)
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", name + ".c"), "-L", os.path.join(ROOT, "zune-jpeg_amd"), "-lzjhip",
                           "-Wl,-rpath," + os.path.join(ROOT, "zune-jpeg_amd"), "-o", out])
    return out


@pytest.mark.parametrize("wh", [(2500, 1786), (272, 72), (1280, 720), (37, 50)])
def test_padded_rows_example(tmp_path, wh):
    """a device output with its rows at the next multiple of 128 bytes equals the host decode row by row, the padding is
    never written (ragged, aligned, a height with a dropped MCU row, a tiny frame)"""
    exe = build("padded_rows", tmp_path)
    r = subprocess.run([exe, str(wh[0]), str(wh[1])], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.stdout, r.stderr)
    assert "0 rows differ, 0 padding bytes written" in r.stdout, r.stdout


def test_shard_frames_example(tmp_path):
    exe = build("shard_frames", tmp_path)
    r = subprocess.run([exe, "5", "528", "72"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.stdout, r.stderr)
    assert "5 frames of 528x72 decoded over" in r.stdout and "status 0" in r.stdout, r.stdout


@pytest.mark.parametrize("entropy", ["cpu", "gpu"])
def test_decode_file_example_writes_the_pixels_the_python_path_gives(tmp_path, entropy):
    """examples/decode_file.c on the reference's own test image (a copy under tests/golden/): the PPM's payload equals
    Decoder.decode_buffer of the same file, with the Huffman stage on the CPU and on the device"""
    import importlib
    import numpy as np
    zj = importlib.import_module("zune-jpeg_amd")
    exe = build("decode_file", tmp_path)
    src = os.path.join(ROOT, "tests", "golden", "test-baseline.jpg")
    ppm = str(tmp_path / "out.ppm")
    r = subprocess.run([exe, src, ppm] + (["gpu"] if entropy == "gpu" else []), capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.stdout, r.stderr)
    raw = open(ppm, "rb").read()
    assert raw.startswith(b"P6\n")
    header_end = 0
    for _ in range(3):                      # "P6", "W H", "255"
        header_end = raw.index(b"\n", header_end) + 1
    want = zj.Decoder().decode_buffer(open(src, "rb").read())
    assert np.array_equal(np.frombuffer(raw[header_end:], np.uint8), want)


@pytest.mark.parametrize("wh", [(4096, 4096), (1920, 1080), (2500, 333)])
def test_stream_frame_example(tmp_path, wh):
    """zj_frame_begin / _rows_ready / _end from plain C: a frame streamed MCU row by MCU row out of pinned planes into pinned
    pixels equals zj_decode_planes of the finished planes"""
    exe = build("stream_frame", tmp_path)
    r = subprocess.run([exe, str(wh[0]), str(wh[1])], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.stdout, r.stderr)
    assert "0 bytes differ from zj_decode_planes" in r.stdout, r.stdout
