"""CPU: libzjhip.so loads, exports every symbol include/zjhip.h declares, pure-host entry points
work, and the product path fails loudly (no CPU fallback) when no HIP device is usable."""
import ctypes as C
import importlib
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def zj():
    m = importlib.import_module("zune-jpeg_amd")
    if not os.path.exists(m.lib_path()):
        import __graft_entry__ as g
        g.build()
    return m


def header_functions():
    src = open(os.path.join(ROOT, "include", "zjhip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = set(re.findall(r"\b(zj_[a-z0-9_]+)\s*\(", src))
    return sorted(n for n in names if not n.endswith("_fn"))


def test_header_symbols_exported(zj):
    L = zj.lib()
    declared = header_functions()
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/zjhip.h but not exported"
    assert sorted(zj.abi_symbols()) == declared
    # ... and NOTHING else: the library is linked with -fvisibility=hidden and a version script, so its dynamic symbol
    # table is the header (no C++ internals, no kernel handles, no micro-benchmark / lab / ablation hooks)
    out = subprocess.check_output(["nm", "-D", "--defined-only", zj.lib_path()], text=True)
    exported = sorted(ln.split()[-1] for ln in out.splitlines() if ln.strip())
    assert exported == declared, sorted(set(exported) ^ set(declared))


def test_lab_kernels_live_in_their_own_library(zj):
    """tools/ubench.py and tools/lab.py use libzjlab.so; the product must not carry their kernels."""
    lab = os.path.join(os.path.dirname(zj.lib_path()), "libzjlab.so")
    if os.path.exists(lab):   # (not part of the default build since round 6: make -C zune-jpeg_amd/csrc lab)
        out = subprocess.check_output(["nm", "-D", "--defined-only", lab], text=True)
        assert "zjlab_ubench" in out and "zjlab_lab" in out
    blob = open(zj.lib_path(), "rb").read()
    for needle in (b"ubench", b"labmem", b"zj_set_ablation", b"zj_set_pad_lds"):
        assert needle not in blob, needle


def test_the_default_build_is_the_product(zj):
    """the default build carries the packed kernel generation only (variants 0 and 2; round 1's wide
    generation comes with `make VARIANTS=all`) and stays under 2 MB (its code objects are stored compressed)."""
    have = zj.variants_available()
    assert 0 in have and 2 in have
    if 1 not in have:
        assert os.path.getsize(zj.lib_path()) < 2_000_000
    assert zj.lib().zj_variant_available(3) == 0 and zj.lib().zj_variant_available(-1) == 0


def test_no_oracle_in_product_library(zj):
    out = subprocess.check_output(["nm", "-D", zj.lib_path()], text=True)
    assert "zjo_" not in out and "zje_" not in out  # oracle / emulator symbols must never be linked


def test_host_only_entry_points(zj):
    L = zj.lib()
    assert L.zj_abi_version() == 8
    assert L.zj_strerror(0) == b"ok"
    assert b"panic" in L.zj_strerror(-5)
    qts = [np.ones(64, np.int32)] * 3
    d = zj.FrameDesc.make(4096, 4096, 2, 2, 3, zj.ColorSpace.RGB, qts)
    assert L.zj_plane_len(C.byref(d), 0) == 16777216
    assert L.zj_plane_len(C.byref(d), 1) == L.zj_plane_len(C.byref(d), 2) == 4194304
    assert L.zj_out_len(C.byref(d)) == 4096 * 4096 * 3
    d = zj.FrameDesc.make(2500, 1786, 1, 1, 3, zj.ColorSpace.GRAYSCALE, qts)
    assert L.zj_plane_len(C.byref(d), 0) == 313 * 64 * 224
    assert [L.zj_num_components(c) for c in range(7)] == [3, 1, 3, 4, 4, 4, 4]
    assert zj.ColorSpace.RGBA.num_components() == 4


def test_dispatch_mirror(zj):
    # HIP arm exists; scalar/avx2 arms belong to the host application -> loud error, no fallback
    assert zj.choose_idct_func(zj.BACKEND_HIP)
    assert zj.choose_upsample_func(zj.BACKEND_HIP, 2, 2)
    assert zj.choose_ycbcr_to_rgb_convert_func(zj.BACKEND_HIP)
    for be in (zj.BACKEND_SCALAR, zj.BACKEND_AVX2):
        with pytest.raises(zj.ZjError):
            zj.choose_idct_func(be)
    with pytest.raises(zj.ZjError):
        zj.choose_upsample_func(zj.BACKEND_HIP, 1, 1)


def test_fails_loudly_without_gpu(zj):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    assert zj.device_count() <= 0
    with pytest.raises(zj.ZjError) as e:
        zj.Context()
    assert e.value.status == -6  # ZJ_ERR_NO_DEVICE
    with pytest.raises(zj.ZjError):
        zj.Context(backend=zj.BACKEND_SCALAR)


def test_round6_entry_points_refuse_a_null_context(zj):
    """zj_frame_* (a frame streamed while its planes are written) and the slot placement queries: argument errors, not
    crashes, without a context / pool"""
    L = zj.lib()
    assert L.zj_frame_begin(None, None, None, None, None, None, 0) == -1      # ZJ_ERR_ARG
    assert L.zj_frame_rows_ready(None, 3) == -1 and L.zj_frame_end(None) == -1 and L.zj_frame_abort(None) == -1
    assert L.zj_pool_slot_numa(None, 0, None, None, None) == -1 and L.zj_multi_slot_numa(None, 0, None, None, None) == -1
    assert L.zj_bind_thread_to_numa_node(-1) == -1 and L.zj_device_pci_bus_id(0, None, 0) == -1


def test_rust_shim_declares_every_symbol():
    """bindings/rust/src/lib.rs cannot be compiled here (no rustc): at least keep its extern block in step with
    include/zjhip.h -- every function it declares exists in the header with the same number of parameters."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "zjhip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    rs = open(os.path.join(root, "bindings", "rust", "src", "lib.rs")).read()
    block = rs[rs.index('extern "C" {'):]
    block = block[:block.index("\n}\n")]
    decls = re.findall(r"pub fn (zj_\w+)\(([^)]*)\)", block, flags=re.S)
    assert len(decls) >= 30
    for name, args in decls:
        m = re.search(r"\b" + name + r"\s*\(([^;]*?)\)\s*;", hdr, flags=re.S)
        assert m, f"{name} is not declared in include/zjhip.h"
        c_args = [a for a in m.group(1).split(",") if a.strip() and a.strip() != "void"]
        r_args = [a for a in args.split(",") if a.strip()]
        assert len(c_args) == len(r_args), (name, len(c_args), len(r_args))
    for field in ("flags", "out_layout", "num_threads", "pinned_planes"):
        assert field in rs
    # both directions: every function of the header is declared in the extern block
    assert sorted(n for n, _ in decls) == header_functions()
    want = int(re.search(r"#define ZJ_ABI_VERSION (\d+)", open(os.path.join(root, "include", "zjhip.h")).read()).group(1))
    assert int(re.search(r"pub const ZJ_ABI_VERSION: c_int = (\d+);", rs).group(1)) == want
    assert f"ABI version {want}" in rs and "ABI version 2" not in rs


def _abi_texts():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    return (open(os.path.join(root, "include", "zjhip.h")).read(),
            open(os.path.join(root, "bindings", "rust", "src", "lib.rs")).read())


def test_rust_shim_types_match_the_header():
    """No rustc anywhere on this pool, so the one thing a compiler + bindgen would prove is checked here by hand: every
    extern fn's return and argument TYPES (pointer depth and constness included: `*const *mut u8` <-> `uint8_t *const *`,
    `usize` <-> `size_t`, `c_int` <-> `int`), the three fn-pointer typedefs and the field order / types / array shapes of
    the four #[repr(C)] structs agree with include/zjhip.h (tests/abi_types.py).  Surface: src/decoder.rs:47,56,
    src/components.rs:14 of the reference."""
    import abi_types
    hdr, rs = _abi_texts()
    assert len(abi_types.c_prototypes(hdr)) == len(header_functions()) >= 70
    assert set(abi_types.c_structs(hdr)) == {"zj_component", "zj_frame_desc", "zj_options", "zj_image_info"}
    assert set(abi_types.c_fn_typedefs(hdr)) == {"zj_idct_fn", "zj_upsample_fn", "zj_color_convert16_fn"}
    assert abi_types.compare(hdr, rs) == []


@pytest.mark.parametrize("side,old,new,expect", [
    ("h", "const int16_t *coeff, size_t n, const int32_t qt[64]", "const int16_t *coeff, int n, const int32_t qt[64]", "zj_idct_strip: parameter 2"),
    ("h", "ZJ_API size_t zj_out_len(", "ZJ_API int zj_out_len(", "zj_out_len: return type"),
    ("h", "uint32_t flags;          /* 0 = the reference", "int32_t flags;          /* 0 = the reference", "struct zj_frame_desc"),
    ("h", "const int16_t *const *d_cb, const int16_t *const *d_cr, uint8_t *const *d_out, void *stream);",
          "const int16_t *const *d_cb, const int16_t **d_cr, uint8_t *const *d_out, void *stream);", "zj_decode_frames_device: parameter 5"),
    ("r", "pub scans: u16, pub restart_interval: u16", "pub scans: u32, pub restart_interval: u16", "struct zj_image_info"),
    ("r", "pub fn zj_plane_len(d: *const zj_frame_desc, comp: c_int) -> usize;", "pub fn zj_plane_len(d: *mut zj_frame_desc, comp: c_int) -> usize;", "zj_plane_len: parameter 0"),
    ("r", "pub quantization_table: [i32; 64],", "pub quantization_table: [i32; 32],", "struct zj_component"),
    ("r", "fn(*mut zj_ctx, *const i16, usize, *mut i16, usize) -> c_int;", "fn(*mut zj_ctx, *const i16, usize, *mut i16, u32) -> c_int;", "fn type zj_upsample_fn"),
    ("r", "pub max_scans: i32, pub num_threads: i32,", "pub num_threads: i32, pub max_scans: i32,", "struct zj_options"),
])
def test_rust_shim_type_check_notices_a_changed_type(side, old, new, expect):
    """the check above fails when ONE type, on either side, is changed"""
    import abi_types
    hdr, rs = _abi_texts()
    src = hdr if side == "h" else rs
    assert src.count(old) == 1, old
    mutated = src.replace(old, new)
    bad = abi_types.compare(mutated if side == "h" else hdr, mutated if side == "r" else rs)
    assert len(bad) == 1 and bad[0].startswith(expect), bad


def test_shard_range_matches_the_python_harness(zj):
    """zj_shard_range (the library's dealer) and shard.shard_range (bench.py's ranks) are the same rule"""
    shard = importlib.import_module("zune-jpeg_amd.shard")
    for n in (0, 1, 7, 8, 128, 1000, 1024, 1025):
        for w in (1, 2, 3, 8):
            got = [zj.shard_range(n, r, w) for r in range(w)]
            assert got == [shard.shard_range(n, r, w) for r in range(w)]
            assert got[0][0] == 0 and got[-1][1] == n and all(a[1] == b[0] for a, b in zip(got, got[1:]))
    assert zj.shard_range(10, 5, 4) == (0, 0) and zj.shard_range(10, -1, 4) == (0, 0)


def test_multi_device_entry_points_fail_loudly_without_gpu(zj):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(zj.ZjError) as e:
        zj.Multi([0, 0])
    assert e.value.status == -6
    with pytest.raises(zj.ZjError) as e:
        zj.Pool(threads=1, devices=[0, 0])
    assert e.value.status == -6
    st = C.c_int(0)
    assert not zj.lib().zj_pool_create_multi(None, 2, 1, None, C.byref(st)) and st.value == -1
    assert not zj.lib().zj_multi_create(None, 0, C.byref(st)) and st.value == -1
    assert zj.pointer_device(None) == -1


def test_rust_shim_facade_mirrors_the_reference_surface():
    """SURVEY.md Appendix B (src/decoder.rs, src/options.rs, src/idct.rs:40, src/upsampler.rs:82,97,
    src/color_convert.rs:61): every public item a user of the reference calls exists in the shim's facade."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rs = open(os.path.join(root, "bindings", "rust", "src", "lib.rs")).read()

    def methods(typ):
        body = rs[rs.index(f"impl {typ} {{"):]
        depth, end = 0, 0
        for i, ch in enumerate(body):
            depth += ch == "{"
            depth -= ch == "}"
            if depth == 0 and ch == "}":
                end = i
                break
        return set(re.findall(r"pub fn (\w+)", body[:end]))

    dec = methods("Decoder")
    assert {"new", "new_with_options", "decode_buffer", "decode_file", "read_headers", "info", "width", "height",
            "get_output_colorspace", "rgba", "set_limits", "set_output_colorspace", "set_num_threads"} <= dec
    opt = methods("ZuneJpegOptions")
    assert {"new", "get_out_colorspace", "set_out_colorspace", "get_use_unsafe", "set_use_unsafe", "get_threads",
            "set_num_threads", "get_max_width", "set_max_width", "get_max_height", "set_max_height", "get_max_scans",
            "set_max_scans", "get_strict_mode", "set_strict_mode"} <= opt
    assert {"set_backend", "set_flags", "set_entropy", "set_out_layout", "set_pinned_planes"} <= opt     # this library's knobs
    for item in ("pub enum Backend { Scalar, Simd, Hip }", "pub fn choose_idct_func(", "pub fn choose_horizontal_samp_function(",
                 "pub fn choose_hv_samp_function(", "pub fn choose_ycbcr_to_rgb_convert_func(", "pub type IDCTPtr",
                 "pub type UpSampler", "pub type ColorConvert16Ptr", "pub enum ColorSpace", "pub struct DecodeErrors",
                 "pub type ImageInfo"):
        assert item in rs, item
    assert "fn num_components" in rs
    # the reference's defaults (src/options.rs:26-40)
    assert "max_width: 1 << 14" in rs and "max_scans: 64" in rs and "num_threads: 4" in rs and "use_unsafe: true" in rs


def test_graft_entry_build_passes():
    """the driver's build check: compiles (or finds up to date) every library and validates symbols + ABI version"""
    import importlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    g = importlib.import_module("__graft_entry__")
    g.build()


@pytest.mark.parametrize("cmd", [["gcc", "-std=c99", "-pedantic"], ["g++", "-std=c++11", "-x", "c++"]])
def test_header_is_plain_c_and_cxx(cmd, tmp_path):
    """include/zjhip.h is the drop-in boundary: plain C99 (and C++11) with no torch / HIP types in it"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "h.c"
    src.write_text('#include "zjhip.h"\nint main(void) { return ZJ_ABI_VERSION == 0; }\n')
    subprocess.check_call([cmd[0], *cmd[1:], "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-I", os.path.join(root, "include"), str(src)])
    hdr = open(os.path.join(root, "include", "zjhip.h")).read()
    assert "hip/hip_runtime" not in hdr and "torch" not in hdr.lower()


@pytest.mark.parametrize("name", ["decode_file", "shard_frames", "padded_rows", "stream_frame"])
def test_c_example_builds_and_links(name):
    """examples/*.c: the C ABI used from plain C99, linked against libzjhip.so (decode_file: one JPEG file; shard_frames:
    frames sharded over the node's GPUs with zj_multi_*; padded_rows: a device output at a pitch that is a multiple of 128 B;
    stream_frame: a frame whose strips go to the GPU while its planes are still being written, zj_frame_*)"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    importlib.import_module("zune-jpeg_amd").lib()  # make sure the library is built
    out = os.path.join(root, "examples", name)
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(root, "include"),
                           os.path.join(root, "examples", name + ".c"), "-L", os.path.join(root, "zune-jpeg_amd"), "-lzjhip",
                           "-Wl,-rpath," + os.path.join(root, "zune-jpeg_amd"), "-o", out])
    assert os.path.exists(out)
    if name in ("shard_frames", "padded_rows", "stream_frame"):  # without a GPU the example must refuse loudly, not compute anything
        import torch
        if not torch.cuda.is_available():
            r = subprocess.run([out, "2", "64", "64"] if name == "shard_frames" else [out, "272", "64"], capture_output=True)
            assert r.returncode == 1 and b"no HIP device" in r.stderr
    os.remove(out)


def test_rust_shim_source_is_at_least_well_formed():
    """no rustc in this image: the cheapest checks a compiler would make first -- balanced delimiters outside strings and
    comments, every `fn` with a body or a `;`, no tabs, the crate-level attributes in front of the first item"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rs = open(os.path.join(root, "bindings", "rust", "src", "lib.rs")).read()
    code = re.sub(r"//[^\n]*", "", rs)                      # line comments (doc comments included)
    code = re.sub(r'"(?:\\.|[^"\\])*"', '""', code)      # string literals
    code = re.sub(r"'(?:\\.|[^'\\])'", "' '", code)      # char literals (lifetimes such as '_ stay)
    stack = []
    pairs = {")": "(", "]": "[", "}": "{"}
    for i, ch in enumerate(code):
        if ch in "([{":
            stack.append((ch, i))
        elif ch in ")]}":
            assert stack and stack[-1][0] == pairs[ch], f"unbalanced {ch!r} near: {code[max(0, i - 60):i + 20]!r}"
            stack.pop()
    assert not stack, f"unclosed {stack[-1][0]!r} near: {code[stack[-1][1]:stack[-1][1] + 80]!r}"
    assert "\t" not in rs
    assert rs.index("#![allow(non_camel_case_types)]") < rs.index("use std::ffi::CStr;")
    for m in re.finditer(r"\bfn\s+\w+[^;{]*([;{])", code):
        assert m.group(1) in ";{"
    # every extern fn is `pub fn name(args) -> ret;` or `pub fn name(args);`
    block = code[code.index('extern "" {'):]
    block = block[:block.index("\n}\n")]
    for line in re.findall(r"pub fn [^;]*;", block, flags=re.S):
        assert re.match(r"pub fn zj_\w+\([^)]*\)(\s*->\s*[^;]+)?;", line, flags=re.S), line
