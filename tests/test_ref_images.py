"""The reference's own integration inputs (tests/large_images.rs, tests/medium_images.rs, tests/random_images.rs,
benches/decode.rs), pinned by tests/golden/ref_images.json (tools/make_ref_image_fixtures.py: product CPU front-end ->
oracle pixel path, checked against libjpeg when recorded).  Five of the files travel as data under tests/golden/ref/;
the rest are read from /root/reference when it exists (the build container), never on the GPU box."""
import hashlib
import importlib
import json
import os

import numpy as np
import pytest

import oracle_c as oc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
REC = {r["file"]: r for r in json.load(open(os.path.join(GOLD, "ref_images.json")))["files"]}
LOCAL = {"tests/inputs/huffman_third_index.jpg": "ref/huffman_third_index.jpg",
         "tests/inputs/single_qt.jpeg": "ref/single_qt.jpeg",
         "tests/inputs/medium_horiz_samp_2500x1786.jpg": "ref/medium_horiz_samp_2500x1786.jpg",
         "benches/images/speed_bench.jpg": "ref/speed_bench.jpg",                                  # 7680 x 4320, 4:4:4
         "benches/images/speed_bench_hv_subsampling.jpg": "ref/speed_bench_hv_subsampling.jpg",    # 7680 x 4320, 4:2:0
         "test-images/test-baseline.jpg": "test-baseline.jpg",
         "test-images/test-progressive.jpg": "test-progressive.jpg"}
REF = "/root/reference"


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def zj():
    return importlib.import_module("zune-jpeg_amd")


def test_the_record_covers_every_reference_input():
    assert len(REC) == 21
    assert sum(1 for r in REC.values() if "error" in r) == 1 and "DAC" in REC["test-images/test-arithmetic-coding.jpg"]["error"]
    assert {(r["h_max"], r["v_max"]) for r in REC.values() if "error" not in r} == {(1, 1), (2, 1), (1, 2), (2, 2)}
    assert REC["tests/inputs/single_qt.jpeg"]["restart_interval"] == 1005
    assert REC["tests/inputs/huffman_third_index.jpg"]["scans"] == 14 and REC["tests/inputs/huffman_third_index.jpg"]["progressive"] == 1
    assert all(r["pillow_mean_abs_diff"] <= 2.0 for r in REC.values() if "pillow_mean_abs_diff" in r)


@pytest.mark.parametrize("name", sorted(LOCAL))
def test_committed_reference_files_decode_to_the_recorded_hashes_on_the_cpu_side(zj, name):
    """CPU half (front-end) + oracle pixel path: what the record was made from must still come out."""
    r = REC[name]
    data = open(os.path.join(GOLD, LOCAL[name]), "rb").read()
    assert hashlib.sha256(data).hexdigest() == r["sha256_file"]
    desc, planes, info = zj.Decoder().decode_coefficients(data)
    assert (info.width, info.height, info.scans, info.restart_interval) == (r["width"], r["height"], r["scans"], r["restart_interval"])
    assert sha(np.concatenate(planes)) == r["sha256_planes"]
    qts = list(np.ctypeslib.as_array(desc.qt))
    rc, px = oc.decode_planes(oc.make_frame(r["width"], r["height"], r["h_max"], r["v_max"], 3, oc.RGB, qts), planes)
    assert rc == 0 and sha(px) == r["sha256_rgb"]


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree exists in the build container only")
@pytest.mark.parametrize("name", sorted(n for n in REC if n not in LOCAL))
def test_reference_tree_files_give_the_recorded_planes(zj, name):
    r = REC[name]
    data = open(os.path.join(REF, name), "rb").read()
    assert hashlib.sha256(data).hexdigest() == r["sha256_file"]
    if "error" in r:
        with pytest.raises(zj.DecodeError) as e:   # tests/random_images.rs has no arithmetic-coding case: unsupported
            zj.Decoder().decode_coefficients(data)
        assert str(e.value) == r["error"]
        return
    # one thread; the reference's default of four and seven (restart segments side by side, or -- every baseline file here but
    # single_qt.jpeg and the 73 KB test-baseline.jpg -- the scan entered at one point per thread: scan_baseline_parallel)
    for threads in (1, 4, 7):
        o = zj.ZuneJpegOptions()
        o.num_threads = threads
        dec = zj.Decoder(o)
        desc, planes, info = dec.decode_coefficients(data, copy=False)
        assert sha(np.concatenate(planes)) == r["sha256_planes"], threads
        if threads > 1 and not r["progressive"] and not r["restart_interval"] and r["bytes"] > 200000:
            assert dec.parallel_mcus() > 0, (name, threads)
        dec.close()


@pytest.mark.gpu
@pytest.mark.parametrize("entropy", ["cpu", "gpu_always"])
@pytest.mark.parametrize("name", sorted(LOCAL))
def test_reference_files_on_the_gpu(zj, name, entropy):
    """decode_buffer through the C ABI -- Huffman on the CPU walker or forced onto the device stage, pixel path on the GPU --
    to the hashes the oracle recorded -- RGB, GRAYSCALE and YCbCr, the three outputs the reference's integration tests ask
    for (tests/large_images.rs:39-153, tests/random_images.rs:38-99, tests/medium_images.rs:82-98)."""
    r = REC[name]
    data = open(os.path.join(GOLD, LOCAL[name]), "rb").read()
    ctx = zj.Context(zj.BACKEND_HIP, 0)
    for cs, key in ((zj.ColorSpace.RGB, "sha256_rgb"), (zj.ColorSpace.GRAYSCALE, "sha256_gray"), (zj.ColorSpace.YCbCr, "sha256_ycbcr")):
        o = zj.ZuneJpegOptions()
        o.out_colorspace = cs
        o.entropy = zj.ENTROPY_CPU if entropy == "cpu" else zj.ENTROPY_GPU_ALWAYS
        dec = zj.Decoder(o, ctx)
        px = dec.decode_buffer(data)
        assert px.size == r["width"] * r["height"] * cs.num_components()
        assert sha(px) == r[key], (name, entropy, key)
        dec.close()
    ctx.close()
