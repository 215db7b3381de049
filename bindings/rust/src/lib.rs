//! FFI to `libzjhip.so` -- the MI355X arm of zune-jpeg's pixel pipeline -- plus a thin safe layer whose
//! names follow the reference crate (`Decoder`, `ZuneJpegOptions`, `ColorSpace`, the three fn-pointer types of
//! `src/decoder.rs:47,56` and `src/components.rs:14`).
//!
//! The C side is `include/zjhip.h` (ABI version 8, checked at run time by `Decoder::new_with_options`).  Output bytes equal the reference's *scalar* arms.
#![allow(non_camel_case_types)]
use std::ffi::CStr;
use std::os::raw::{c_char, c_int, c_void};

pub const ZJ_ABI_VERSION: c_int = 8;
pub const ZJ_SCATTER_MAX: usize = 32;
pub const ZJ_BACKEND_SCALAR: c_int = 0;
pub const ZJ_BACKEND_AVX2: c_int = 1;
pub const ZJ_BACKEND_HIP: c_int = 2;
pub const ZJ_OK: c_int = 0;
pub const ZJ_ERR_PANIC: c_int = -5;
pub const ZJ_FLAG_PLAIN_TAIL: u32 = 1;
pub const ZJ_FLAG_CLAMP_DC: u32 = 2;
pub const ZJ_FLAG_EDGE_REPLICATE: u32 = 4;
pub const ZJ_FLAG_CORRECTED: u32 = 7;
/// zj_options.flags only: AC values as the file codes them (the reference cuts some fast-AC values to six bits)
pub const ZJ_FLAG_FULL_AC_VALUES: u32 = 8;
pub const ZJ_LAYOUT_HWC: u32 = 0;
pub const ZJ_LAYOUT_CHW: u32 = 1;

/// `ColorSpace`, `src/misc.rs:88-106` (same discriminants).
#[repr(i32)]
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub enum ColorSpace { RGB = 0, GRAYSCALE = 1, YCbCr = 2, CMYK = 3, YCCK = 4, RGBA = 5, RGBX = 6 }

impl ColorSpace {
    /// `misc.rs:113-121`
    pub fn num_components(self) -> usize {
        match self { ColorSpace::RGB | ColorSpace::YCbCr => 3, ColorSpace::GRAYSCALE => 1, _ => 4 }
    }
}

#[repr(C)]
pub struct zj_component {            // <-> Components, src/components.rs:18-43
    pub horizontal_sample: usize,
    pub vertical_sample: usize,
    pub width_stride: usize,
    pub quantization_table: [i32; 64],
}

#[repr(C)]
#[derive(Clone)]
pub struct zj_frame_desc {
    pub width: u32, pub height: u32, pub h_max: u32, pub v_max: u32,
    pub in_components: u32, pub out_colorspace: i32,
    pub qt: [[i32; 64]; 3],
    pub flags: u32, pub out_layout: u32,
    pub out_pitch: u32,             // bytes between output rows, 0 = tight (device outputs only)
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct zj_options {              // <-> ZuneJpegOptions, src/options.rs:6-40 (zero = reference default)
    pub out_colorspace: i32, pub strict_mode: i32, pub max_width: i32, pub max_height: i32,
    pub max_scans: i32, pub num_threads: i32, pub pinned_planes: i32,
    pub flags: u32, pub out_layout: u32,
    pub entropy: i32,                // 0 CPU walker, 1 baseline scans of 32 KB and more on the GPU, 2 every eligible scan
}

#[repr(C)]
#[derive(Clone, Copy, Default, Debug)]
pub struct zj_image_info {           // <-> ImageInfo, src/decoder.rs:652-668
    pub width: u16, pub height: u16,
    pub components: u8, pub progressive: u8, pub h_max: u8, pub v_max: u8,
    pub scans: u16, pub restart_interval: u16,
}

#[repr(C)] pub struct zj_ctx { _private: [u8; 0] }
#[repr(C)] pub struct zj_decoder { _private: [u8; 0] }
#[repr(C)] pub struct zj_pool { _private: [u8; 0] }
#[repr(C)] pub struct zj_multi { _private: [u8; 0] }

/// the C fn-pointer types `zj_choose_*` hand out (`include/zjhip.h`)
pub type zj_idct_fn = unsafe extern "C" fn(*mut zj_ctx, *const i16, usize, *const i32, usize, usize, usize, *mut i16) -> c_int;
pub type zj_upsample_fn = unsafe extern "C" fn(*mut zj_ctx, *const i16, usize, *mut i16, usize) -> c_int;
pub type zj_color_convert16_fn = unsafe extern "C" fn(*mut zj_ctx, *const i16, *const i16, *const i16, *mut u8, usize, *mut usize) -> c_int;

extern "C" {
    pub fn zj_abi_version() -> c_int;
    pub fn zj_device_count() -> c_int;
    pub fn zj_ctx_create(backend: c_int, device: c_int, status: *mut c_int) -> *mut zj_ctx;
    pub fn zj_ctx_destroy(ctx: *mut zj_ctx);
    pub fn zj_default_ctx() -> *mut zj_ctx;
    pub fn zj_strerror(status: c_int) -> *const c_char;
    pub fn zj_last_error(ctx: *const zj_ctx) -> *const c_char;
    pub fn zj_idct_strip(ctx: *mut zj_ctx, coeff: *const i16, n: usize, qt: *const i32, stride: usize,
                         samp_factors: usize, v_samp: usize, out: *mut i16) -> c_int;
    pub fn zj_upsample_h(ctx: *mut zj_ctx, inp: *const i16, n: usize, out: *mut i16, out_len: usize) -> c_int;
    pub fn zj_upsample_v(ctx: *mut zj_ctx, inp: *const i16, n: usize, out: *mut i16, out_len: usize) -> c_int;
    pub fn zj_upsample_hv(ctx: *mut zj_ctx, inp: *const i16, n: usize, out: *mut i16, out_len: usize) -> c_int;
    pub fn zj_ycbcr_to_rgb16(ctx: *mut zj_ctx, y: *const i16, cb: *const i16, cr: *const i16,
                             out: *mut u8, out_len: usize, pos: *mut usize) -> c_int;
    pub fn zj_post_process_strip(ctx: *mut zj_ctx, coeff: *const *const i16, len: *const usize,
                                 comps: *const zj_component, in_cs: c_int, out_cs: c_int,
                                 out: *mut u8, out_len: usize, width: usize) -> c_int;
    pub fn zj_plane_len(d: *const zj_frame_desc, comp: c_int) -> usize;
    pub fn zj_out_len(d: *const zj_frame_desc) -> usize;
    pub fn zj_num_components(colorspace: c_int) -> c_int;
    pub fn zj_decode_planes(ctx: *mut zj_ctx, d: *const zj_frame_desc, y: *const i16, cb: *const i16,
                            cr: *const i16, out: *mut u8) -> c_int;
    pub fn zj_decode_planes_batch(ctx: *mut zj_ctx, d: *const zj_frame_desc, nframes: usize, y: *const i16,
                                  cb: *const i16, cr: *const i16, out: *mut u8) -> c_int;
    pub fn zj_decode_planes_device(ctx: *mut zj_ctx, d: *const zj_frame_desc, nframes: usize, d_y: *const i16,
                                   d_cb: *const i16, d_cr: *const i16, d_out: *mut u8, stream: *mut c_void) -> c_int;
    pub fn zj_alloc_pinned(bytes: usize) -> *mut c_void;
    pub fn zj_free_pinned(p: *mut c_void);
    pub fn zj_set_thread_device(device: c_int) -> c_int;
    pub fn zj_decoder_new(opt: *const zj_options) -> *mut zj_decoder;
    pub fn zj_decoder_free(d: *mut zj_decoder);
    pub fn zj_decoder_error(d: *const zj_decoder) -> *const c_char;
    pub fn zj_decoder_read_headers(d: *mut zj_decoder, buf: *const u8, len: usize, info: *mut zj_image_info) -> c_int;
    pub fn zj_decoder_decode_coefficients(d: *mut zj_decoder, buf: *const u8, len: usize, desc: *mut zj_frame_desc,
                                          planes: *mut *const i16, plane_len: *mut usize, info: *mut zj_image_info) -> c_int;
    pub fn zj_decoder_prepare(d: *mut zj_decoder, buf: *const u8, len: usize, desc: *mut zj_frame_desc,
                              info: *mut zj_image_info) -> c_int;
    pub fn zj_decoder_finish_pixels(d: *mut zj_decoder, ctx: *mut zj_ctx, out: *mut u8, out_cap: usize,
                                    out_len: *mut usize) -> c_int;
    pub fn zj_decoder_finish_pixels_device(d: *mut zj_decoder, ctx: *mut zj_ctx, d_out: *mut u8, out_cap: usize,
                                           out_len: *mut usize) -> c_int;
    pub fn zj_decode_scans(ctx: *mut zj_ctx, n: usize, descs: *const zj_frame_desc, blobs: *const *const c_void,
                           blob_bytes: *const usize, outs: *const *mut u8, outs_on_device: c_int, rcs: *mut c_int,
                           status_bits: *mut u32) -> c_int;
    pub fn zj_decoder_finish_pixels_batch(ds: *const *mut zj_decoder, n: usize, ctx: *mut zj_ctx, outs: *const *mut u8,
                                          out_caps: *const usize, out_lens: *mut usize, outs_on_device: c_int,
                                          rcs: *mut c_int) -> c_int;
    pub fn zj_decode_scan(ctx: *mut zj_ctx, d: *const zj_frame_desc, blob: *const c_void, blob_bytes: usize,
                          out: *mut u8, out_on_device: c_int, status_bits: *mut u32) -> c_int;
    pub fn zj_decoder_decode_buffer(d: *mut zj_decoder, ctx: *mut zj_ctx, buf: *const u8, len: usize, out: *mut u8,
                                    out_cap: usize, out_len: *mut usize, info: *mut zj_image_info) -> c_int;
    pub fn zj_pool_create(device: c_int, threads: c_int, opt: *const zj_options, status: *mut c_int) -> *mut zj_pool;
    pub fn zj_pool_destroy(pool: *mut zj_pool);
    pub fn zj_pool_error(pool: *const zj_pool) -> *const c_char;
    pub fn zj_pool_decode_files(pool: *mut zj_pool, nfiles: usize, bufs: *const *const u8, lens: *const usize,
                                outs: *const *mut u8, out_caps: *const usize, out_lens: *mut usize,
                                infos: *mut zj_image_info, statuses: *mut c_int) -> c_int;
    pub fn zj_pool_decode_files_device(pool: *mut zj_pool, nfiles: usize, bufs: *const *const u8, lens: *const usize,
                                       d_outs: *const *mut u8, out_caps: *const usize, out_lens: *mut usize,
                                       infos: *mut zj_image_info, statuses: *mut c_int) -> c_int;
    pub fn zj_choose_idct_func(backend: c_int) -> Option<zj_idct_fn>;
    pub fn zj_choose_upsample_func(backend: c_int, h_max: c_int, v_max: c_int) -> Option<zj_upsample_fn>;
    pub fn zj_choose_ycbcr_to_rgb_convert_func(backend: c_int, out_cs: c_int) -> Option<zj_color_convert16_fn>;
    pub fn zj_frame_begin(ctx: *mut zj_ctx, d: *const zj_frame_desc, y: *const i16, cb: *const i16, cr: *const i16,
                          out: *mut u8, out_on_device: c_int) -> c_int;
    pub fn zj_frame_rows_ready(ctx: *mut zj_ctx, mcu_rows: usize) -> c_int;
    pub fn zj_frame_end(ctx: *mut zj_ctx) -> c_int;
    pub fn zj_frame_abort(ctx: *mut zj_ctx) -> c_int;
    pub fn zj_decode_planes_to_device(ctx: *mut zj_ctx, d: *const zj_frame_desc, y: *const i16, cb: *const i16,
                                      cr: *const i16, d_out: *mut u8) -> c_int;
    pub fn zj_time_decode_device(ctx: *mut zj_ctx, d: *const zj_frame_desc, nframes: usize, d_y: *const i16,
                                 d_cb: *const i16, d_cr: *const i16, d_out: *mut u8, stream: *mut c_void, iters: c_int,
                                 ms_total: *mut f32, ms_each: *mut f32, kernel_name: *mut *const c_char) -> c_int;
    pub fn zj_decoder_gpu_status(d: *const zj_decoder) -> u32;
    pub fn zj_decoder_parallel_segments(d: *const zj_decoder) -> c_int;
    pub fn zj_decoder_parallel_mcus(d: *const zj_decoder) -> i64;
    pub fn zj_decoder_set_num_threads(d: *mut zj_decoder, threads: c_int) -> c_int;
    pub fn zj_decoder_scan_blob(d: *const zj_decoder, blob: *mut *const c_void, len: *mut usize) -> c_int;
    pub fn zj_scan_planes(ctx: *mut zj_ctx, y: *mut i16, cb: *mut i16, cr: *mut i16, len: *mut usize) -> c_int;
    pub fn zj_scan_stats(ctx: *const zj_ctx, rounds: *mut c_int, ms: *mut f32) -> c_int;
    pub fn zj_pool_threads(pool: *const zj_pool) -> c_int;
    pub fn zj_pool_stats(pool: *mut zj_pool, entropy_seconds: *mut f64, gpu_seconds: *mut f64, files: *mut usize) -> c_int;
    pub fn zj_device_alloc(ctx: *mut zj_ctx, bytes: usize) -> *mut c_void;
    pub fn zj_device_free(ctx: *mut zj_ctx, p: *mut c_void);
    pub fn zj_memcpy_h2d(ctx: *mut zj_ctx, dst: *mut c_void, src: *const c_void, bytes: usize) -> c_int;
    pub fn zj_memcpy_d2h(ctx: *mut zj_ctx, dst: *mut c_void, src: *const c_void, bytes: usize) -> c_int;
    pub fn zj_device_memset(ctx: *mut zj_ctx, d_ptr: *mut c_void, value: c_int, bytes: usize) -> c_int;
    pub fn zj_sync(ctx: *mut zj_ctx) -> c_int;
    pub fn zj_set_variant(ctx: *mut zj_ctx, variant: c_int) -> c_int;
    pub fn zj_variant_available(variant: c_int) -> c_int;
    pub fn zj_set_pipeline(ctx: *mut zj_ctx, on: c_int) -> c_int;
    pub fn zj_decode_frames(ctx: *mut zj_ctx, d: *const zj_frame_desc, nframes: usize, y: *const *const i16,
                            cb: *const *const i16, cr: *const *const i16, out: *const *mut u8) -> c_int;
    pub fn zj_decode_planes_device_strided(ctx: *mut zj_ctx, d: *const zj_frame_desc, nframes: usize, d_y: *const i16,
                                           d_cb: *const i16, d_cr: *const i16, d_out: *mut u8, y_stride: usize,
                                           c_stride: usize, out_stride: usize, stream: *mut c_void) -> c_int;
    pub fn zj_decode_frames_device(ctx: *mut zj_ctx, d: *const zj_frame_desc, nframes: usize, d_y: *const *const i16,
                                   d_cb: *const *const i16, d_cr: *const *const i16, d_out: *const *mut u8,
                                   stream: *mut c_void) -> c_int;
    pub fn zj_pointer_device(p: *const c_void) -> c_int;
    pub fn zj_device_pci_bus_id(device: c_int, buf: *mut c_char, cap: usize) -> c_int;
    pub fn zj_device_numa_node(device: c_int) -> c_int;
    pub fn zj_bind_thread_to_numa_node(node: c_int) -> c_int;
    pub fn zj_bind_thread_near_device(device: c_int) -> c_int;
    pub fn zj_thread_numa_node() -> c_int;
    pub fn zj_pool_slot_numa(pool: *mut zj_pool, slot: c_int, device_node: *mut c_int, threads_bound: *mut c_int,
                             threads: *mut c_int) -> c_int;
    pub fn zj_multi_slot_numa(m: *mut zj_multi, slot: c_int, device_node: *mut c_int, thread_node: *mut c_int,
                              bound: *mut c_int) -> c_int;
    pub fn zj_pool_create_multi(devices: *const c_int, ndev: c_int, threads_per_device: c_int, opt: *const zj_options,
                                status: *mut c_int) -> *mut zj_pool;
    pub fn zj_pool_devices(pool: *const zj_pool) -> c_int;
    pub fn zj_pool_device_stats(pool: *mut zj_pool, slot: c_int, device: *mut c_int, gpu_seconds: *mut f64,
                                files: *mut usize) -> c_int;
    pub fn zj_shard_range(nframes: usize, slot: c_int, nslots: c_int, lo: *mut usize, hi: *mut usize);
    pub fn zj_multi_create(devices: *const c_int, ndev: c_int, status: *mut c_int) -> *mut zj_multi;
    pub fn zj_multi_destroy(m: *mut zj_multi);
    pub fn zj_multi_devices(m: *const zj_multi) -> c_int;
    pub fn zj_multi_ctx(m: *mut zj_multi, slot: c_int) -> *mut zj_ctx;
    pub fn zj_multi_slot_stats(m: *mut zj_multi, slot: c_int, device: *mut c_int, frames: *mut usize) -> c_int;
    pub fn zj_multi_decode_planes_batch(m: *mut zj_multi, d: *const zj_frame_desc, nframes: usize, y: *const i16,
                                        cb: *const i16, cr: *const i16, out: *mut u8, statuses: *mut c_int) -> c_int;
    pub fn zj_multi_decode_frames(m: *mut zj_multi, d: *const zj_frame_desc, nframes: usize, y: *const *const i16,
                                  cb: *const *const i16, cr: *const *const i16, out: *const *mut u8,
                                  statuses: *mut c_int) -> c_int;
    pub fn zj_multi_decode_frames_device(m: *mut zj_multi, d: *const zj_frame_desc, nframes: usize, d_y: *const *const i16,
                                         d_cb: *const *const i16, d_cr: *const *const i16, d_out: *const *mut u8,
                                         statuses: *mut c_int) -> c_int;
}

fn check(rc: c_int, what: &str) {
    // the reference's pixel functions cannot fail; where it would panic, so do we
    if rc != ZJ_OK {
        let msg = unsafe { CStr::from_ptr(zj_strerror(rc)) }.to_string_lossy().into_owned();
        panic!("{}: zjhip status {} ({})", what, rc, msg);
    }
}

/// `Aligned32<[i32;64]>` of the reference (`src/misc.rs:70-72`); the HIP arm has no alignment requirement.
#[repr(align(32))]
pub struct Aligned32<T>(pub T);

/// Drop-in for `IDCTPtr` (`src/decoder.rs:56`): `dequantize_and_idct_int` / `_avx2`.
pub fn dequantize_and_idct_hip(vector: &[i16], qt: &Aligned32<[i32; 64]>, stride: usize, samp_factors: usize,
                               v_samp: usize) -> Vec<i16> {
    let mut out = vec![0i16; vector.len()];
    check(unsafe { zj_idct_strip(zj_default_ctx(), vector.as_ptr(), vector.len(), qt.0.as_ptr(), stride,
                                 samp_factors, v_samp, out.as_mut_ptr()) }, "zj_idct_strip");
    out
}

macro_rules! upsampler {
    ($name:ident, $ffi:ident) => {
        /// Drop-in for `UpSampler` (`src/components.rs:14`).
        pub fn $name(input: &[i16], output_len: usize) -> Vec<i16> {
            let mut out = vec![0i16; output_len];
            check(unsafe { $ffi(zj_default_ctx(), input.as_ptr(), input.len(), out.as_mut_ptr(), output_len) },
                  stringify!($ffi));
            out
        }
    };
}
upsampler!(upsample_horizontal_hip, zj_upsample_h);
upsampler!(upsample_vertical_hip, zj_upsample_v);
upsampler!(upsample_hv_hip, zj_upsample_hv);

/// Drop-in for `ColorConvert16Ptr` (`src/decoder.rs:47`).
pub fn ycbcr_to_rgb_hip_16(y: &[i16; 16], cb: &[i16; 16], cr: &[i16; 16], out: &mut [u8], pos: &mut usize) {
    check(unsafe { zj_ycbcr_to_rgb16(zj_default_ctx(), y.as_ptr(), cb.as_ptr(), cr.as_ptr(), out.as_mut_ptr(),
                                     out.len(), pos as *mut usize) }, "zj_ycbcr_to_rgb16");
}

/// The C side reads `zj_plane_len` elements behind every plane pointer it is given: a safe function must not hand it a
/// shorter slice.  The chroma planes are read whenever the frame has three components (an empty slice is too short then).
fn check_plane_lengths(d: &zj_frame_desc, frames: &[[&[i16]; 3]]) {
    let (ylen, clen) = unsafe { (zj_plane_len(d, 0), if d.in_components == 3 { zj_plane_len(d, 1) } else { 0 }) };
    assert!(ylen > 0, "not a decodable frame descriptor");
    for f in frames {
        assert!(f[0].len() >= ylen, "luma plane too short");
        assert!(f[1].len() >= clen && f[2].len() >= clen, "chroma plane too short");
    }
}

/// Frames the caller owns as independent `Vec`s -- the shape the reference's own callers have (a fresh `Vec` per strip,
/// `src/mcu.rs:238-250`; one `Vec<u8>` per decode, `src/decoder.rs:178`) -- decoded together: one pipelined pass over the
/// GPU instead of one launch per frame (`zj_decode_frames`).  `frames[f]` = `[y, cb, cr]` coefficient planes of frame `f`.
pub fn decode_frames(d: &zj_frame_desc, frames: &[[&[i16]; 3]]) -> Vec<Vec<u8>> {
    let n_out = unsafe { zj_out_len(d) };
    let mut outs: Vec<Vec<u8>> = frames.iter().map(|_| vec![0u8; n_out]).collect();
    let y: Vec<*const i16> = frames.iter().map(|f| f[0].as_ptr()).collect();
    let cb: Vec<*const i16> = frames.iter().map(|f| f[1].as_ptr()).collect();
    let cr: Vec<*const i16> = frames.iter().map(|f| f[2].as_ptr()).collect();
    let o: Vec<*mut u8> = outs.iter_mut().map(|v| v.as_mut_ptr()).collect();
    check_plane_lengths(d, frames);
    check(unsafe { zj_decode_frames(zj_default_ctx(), d, frames.len(), y.as_ptr(), cb.as_ptr(), cr.as_ptr(), o.as_ptr()) },
          "zj_decode_frames");
    outs
}

/// Image-level sharding over the GPUs of one node (`zj_multi_*`): frames are independent (the reference's own unit of
/// independence is the strip, `src/mcu.rs:225-226,356-368`), so N frames over D devices are D contiguous shards and no
/// collective; one context and one host thread per device slot, all slots at once.
pub struct Multi { m: *mut zj_multi }
unsafe impl Send for Multi {}
impl Multi {
    /// `devices[k]` = HIP device of slot k (a device may fill several slots); the 8-GPU node: `&[0, 1, 2, 3, 4, 5, 6, 7]`
    pub fn new(devices: &[c_int]) -> Result<Multi, DecodeErrors> {
        let mut st: c_int = 0;
        let m = unsafe { zj_multi_create(devices.as_ptr(), devices.len() as c_int, &mut st) };
        if m.is_null() {
            let msg = unsafe { CStr::from_ptr(zj_strerror(st)) }.to_string_lossy().into_owned();
            return Err(DecodeErrors { status: st, message: msg });
        }
        Ok(Multi { m })
    }
    pub fn slots(&self) -> usize { unsafe { zj_multi_devices(self.m) as usize } }
    /// frames `[lo, hi)` that slot `slot` of `nslots` decodes out of `nframes` (`zj_shard_range`)
    pub fn shard_range(nframes: usize, slot: usize, nslots: usize) -> (usize, usize) {
        let (mut lo, mut hi) = (0usize, 0usize);
        unsafe { zj_shard_range(nframes, slot as c_int, nslots as c_int, &mut lo, &mut hi) };
        (lo, hi)
    }
    /// `frames[f]` = `[y, cb, cr]` coefficient planes of frame `f`, each an allocation of its own; returns the frames' pixels
    pub fn decode_frames(&self, d: &zj_frame_desc, frames: &[[&[i16]; 3]]) -> Result<Vec<Vec<u8>>, DecodeErrors> {
        let n_out = unsafe { zj_out_len(d) };
        let mut outs: Vec<Vec<u8>> = frames.iter().map(|_| vec![0u8; n_out]).collect();
        let y: Vec<*const i16> = frames.iter().map(|f| f[0].as_ptr()).collect();
        let cb: Vec<*const i16> = frames.iter().map(|f| f[1].as_ptr()).collect();
        let cr: Vec<*const i16> = frames.iter().map(|f| f[2].as_ptr()).collect();
        let o: Vec<*mut u8> = outs.iter_mut().map(|v| v.as_mut_ptr()).collect();
        check_plane_lengths(d, frames);
        let mut statuses = vec![0 as c_int; self.slots()];
        let rc = unsafe { zj_multi_decode_frames(self.m, d, frames.len(), y.as_ptr(), cb.as_ptr(), cr.as_ptr(), o.as_ptr(),
                                                 statuses.as_mut_ptr()) };
        if rc != ZJ_OK {
            let msg = unsafe { CStr::from_ptr(zj_strerror(rc)) }.to_string_lossy().into_owned();
            return Err(DecodeErrors { status: rc, message: format!("{} (per slot: {:?})", msg, statuses) });
        }
        Ok(outs)
    }
}
impl Drop for Multi { fn drop(&mut self) { unsafe { zj_multi_destroy(self.m) } } }

/// `IDCTPtr` (`src/decoder.rs:56`), `UpSampler` (`src/components.rs:14`), `ColorConvert16Ptr` (`src/decoder.rs:47`).
pub type IDCTPtr = fn(&[i16], &Aligned32<[i32; 64]>, usize, usize, usize) -> Vec<i16>;
pub type UpSampler = fn(&[i16], usize) -> Vec<i16>;
pub type ColorConvert16Ptr = fn(&[i16; 16], &[i16; 16], &[i16; 16], &mut [u8], &mut usize);

/// Which arm of the reference's dispatch a decoder uses.  `Scalar` and `Simd` are the reference crate's own arms
/// (`use_unsafe = false / true`) and stay there; this crate implements `Hip`.
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub enum Backend { Scalar, Simd, Hip }

/// `choose_idct_func` (`src/idct.rs:40`) with the hip arm: `None` for the arms that live in the reference crate
/// (the maintainer's patched `choose_idct_func` falls through to them).
pub fn choose_idct_func(backend: Backend) -> Option<IDCTPtr> {
    match backend { Backend::Hip => Some(dequantize_and_idct_hip), _ => None }
}
/// `choose_horizontal_samp_function` (`src/upsampler.rs:82`)
pub fn choose_horizontal_samp_function(backend: Backend) -> Option<UpSampler> {
    match backend { Backend::Hip => Some(upsample_horizontal_hip), _ => None }
}
/// `choose_hv_samp_function` (`src/upsampler.rs:97`)
pub fn choose_hv_samp_function(backend: Backend) -> Option<UpSampler> {
    match backend { Backend::Hip => Some(upsample_hv_hip), _ => None }
}
/// the (1,2) case of `Decoder::set_upsampling` (`src/decoder.rs:492-501`; the reference has a scalar arm only)
pub fn choose_v_samp_function(backend: Backend) -> Option<UpSampler> {
    match backend { Backend::Hip => Some(upsample_vertical_hip), _ => None }
}
/// `choose_ycbcr_to_rgb_convert_func` (`src/color_convert.rs:61`): the reference only ever asks for `ColorSpace::RGB`
pub fn choose_ycbcr_to_rgb_convert_func(type_need: ColorSpace, backend: Backend) -> Option<ColorConvert16Ptr> {
    match (backend, type_need) { (Backend::Hip, ColorSpace::RGB) => Some(ycbcr_to_rgb_hip_16), _ => None }
}

/// `DecodeErrors` (`src/errors.rs:16-43`) flattened: the status code of `include/zjhip.h` and the reference's text.
#[derive(Debug)]
pub struct DecodeErrors { pub status: i32, pub message: String }
impl std::fmt::Display for DecodeErrors {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result { write!(f, "{}", self.message) }
}
impl std::error::Error for DecodeErrors {}

/// `ImageInfo` (`src/decoder.rs:652-668`): the fields the C side reports (`zj_image_info`).
pub type ImageInfo = zj_image_info;

/// `ZuneJpegOptions` (`src/options.rs:6-160`; defaults `:26-40`: use_unsafe, RGB, 4 threads, 16384 x 16384, 64 scans,
/// not strict) plus the knobs of this library.
#[derive(Clone, Copy)]
pub struct ZuneJpegOptions { raw: zj_options, use_unsafe: bool, backend: Backend }
impl Default for ZuneJpegOptions {
    fn default() -> Self {
        let raw = zj_options { out_colorspace: ColorSpace::RGB as i32, strict_mode: 0, max_width: 1 << 14, max_height: 1 << 14,
                               max_scans: 64, num_threads: 4, ..zj_options::default() };
        ZuneJpegOptions { raw, use_unsafe: true, backend: Backend::Hip }
    }
}
impl ZuneJpegOptions {
    pub fn new() -> Self { Self::default() }
    pub fn get_out_colorspace(&self) -> ColorSpace { colorspace_from(self.raw.out_colorspace) }
    pub fn set_out_colorspace(mut self, cs: ColorSpace) -> Self { self.raw.out_colorspace = cs as i32; self }
    /// `options.rs:66-73`: in the reference this selects the SIMD arms; the hip arm ignores it (kept so that callers
    /// compile unchanged; `Backend::Scalar` / `Backend::Simd` are the reference crate's business)
    pub fn get_use_unsafe(&self) -> bool { self.use_unsafe }
    pub fn set_use_unsafe(mut self, choice: bool) -> Self { self.use_unsafe = choice; self }
    pub fn get_threads(&self) -> u32 { self.raw.num_threads as u32 }
    pub fn set_num_threads(mut self, count: std::num::NonZeroU32) -> Self { self.raw.num_threads = count.get() as i32; self }
    pub fn get_max_width(&self) -> u16 { self.raw.max_width as u16 }
    pub fn set_max_width(mut self, w: u16) -> Self { self.raw.max_width = w as i32; self }
    pub fn get_max_height(&self) -> u16 { self.raw.max_height as u16 }
    pub fn set_max_height(mut self, h: u16) -> Self { self.raw.max_height = h as i32; self }
    pub fn get_max_scans(&self) -> usize { self.raw.max_scans as usize }
    pub fn set_max_scans(mut self, n: usize) -> Self { self.raw.max_scans = n as i32; self }
    pub fn get_strict_mode(&self) -> bool { self.raw.strict_mode != 0 }
    pub fn set_strict_mode(mut self, yes: bool) -> Self { self.raw.strict_mode = yes as i32; self }
    // ---- this library's knobs --------------------------------------------------------------------------------------
    pub fn get_backend(&self) -> Backend { self.backend }
    /// only `Backend::Hip` can be served by this crate; `Decoder::new_with_options` rejects the others
    pub fn set_backend(mut self, b: Backend) -> Self { self.backend = b; self }
    /// ZJ_FLAG_* (0 = the reference's bytes; ZJ_FLAG_CORRECTED = the non-quirk mode)
    pub fn set_flags(mut self, flags: u32) -> Self { self.raw.flags = flags; self }
    /// 0 = Huffman on the CPU (the default), 1 = baseline scans of 32 KB and more on the GPU, 2 = every eligible scan
    pub fn set_entropy(mut self, mode: i32) -> Self { self.raw.entropy = mode; self }
    /// ZJ_LAYOUT_HWC (the reference's interleaved bytes) or ZJ_LAYOUT_CHW (planar u8, tensor consumers)
    pub fn set_out_layout(mut self, layout: u32) -> Self { self.raw.out_layout = layout; self }
    /// coefficient planes in pinned host memory (faster uploads)
    pub fn set_pinned_planes(mut self, yes: bool) -> Self { self.raw.pinned_planes = yes as i32; self }
}
fn colorspace_from(v: i32) -> ColorSpace {
    match v { 1 => ColorSpace::GRAYSCALE, 2 => ColorSpace::YCbCr, 3 => ColorSpace::CMYK, 4 => ColorSpace::YCCK,
              5 => ColorSpace::RGBA, 6 => ColorSpace::RGBX, _ => ColorSpace::RGB }
}

/// `Decoder` (`src/decoder.rs:60`): CPU entropy decode (or the device stage, `set_entropy`), GPU pixel path.
pub struct Decoder { d: *mut zj_decoder, ctx: *mut zj_ctx, info: Option<ImageInfo>, options: ZuneJpegOptions, stale: bool }

impl Decoder {
    /// `decoder.rs:186`
    pub fn new() -> Decoder { Decoder::new_with_options(ZuneJpegOptions::new()) }
    /// `decoder.rs:462`
    pub fn new_with_options(o: ZuneJpegOptions) -> Decoder {
        assert_eq!(unsafe { zj_abi_version() }, ZJ_ABI_VERSION, "libzjhip.so and this crate disagree about the ABI");
        assert!(o.backend == Backend::Hip, "this crate serves Backend::Hip; the scalar / SIMD arms are the reference crate's");
        Decoder { d: unsafe { zj_decoder_new(&o.raw) }, ctx: unsafe { zj_default_ctx() }, info: None, options: o, stale: false }
    }
    fn handle(&mut self) -> *mut zj_decoder {
        if self.stale { // a deprecated setter changed the options: the C decoder is configured at creation
            unsafe { zj_decoder_free(self.d); self.d = zj_decoder_new(&self.options.raw); }
            self.stale = false;
        }
        self.d
    }
    fn err(&self, rc: c_int) -> DecodeErrors {
        let m = unsafe { CStr::from_ptr(zj_decoder_error(self.d)) }.to_string_lossy().into_owned();
        DecodeErrors { status: rc, message: m }
    }
    /// `decoder.rs:452`
    pub fn read_headers(&mut self, buf: &[u8]) -> Result<(), DecodeErrors> {
        let mut info = zj_image_info::default();
        let rc = unsafe { zj_decoder_read_headers(self.handle(), buf.as_ptr(), buf.len(), &mut info) };
        if rc != ZJ_OK { return Err(self.err(rc)); }
        self.info = Some(info);
        Ok(())
    }
    /// `decoder.rs:210`
    pub fn info(&self) -> Option<ImageInfo> { self.info }
    /// `decoder.rs:561,570`
    pub fn width(&self) -> u16 { self.info.map_or(0, |i| i.width) }
    pub fn height(&self) -> u16 { self.info.map_or(0, |i| i.height) }
    /// `decoder.rs:415`
    pub fn get_output_colorspace(&self) -> ColorSpace { self.options.get_out_colorspace() }
    /// `decoder.rs:178`
    pub fn decode_buffer(&mut self, buf: &[u8]) -> Result<Vec<u8>, DecodeErrors> {
        self.read_headers(buf)?;
        let i = self.info.unwrap();
        let nc = if i.components == 1 { 1 } else { self.options.get_out_colorspace().num_components() };
        let mut out = vec![0u8; i.width as usize * i.height as usize * nc];
        let (mut n, mut info) = (0usize, zj_image_info::default());
        let rc = unsafe { zj_decoder_decode_buffer(self.handle(), self.ctx, buf.as_ptr(), buf.len(), out.as_mut_ptr(),
                                                   out.len(), &mut n, &mut info) };
        if rc != ZJ_OK { return Err(self.err(rc)); }
        out.truncate(n);
        self.info = Some(info);
        Ok(out)
    }
    /// `decoder.rs:193`
    pub fn decode_file<P: AsRef<std::path::Path> + Clone>(&mut self, file: P) -> Result<Vec<u8>, DecodeErrors> {
        let buf = std::fs::read(file).map_err(|e| DecodeErrors { status: -1, message: e.to_string() })?;
        self.decode_buffer(&buf)
    }
    // ---- the reference's deprecated setters (`decoder.rs:535-600`) ---------------------------------------------------
    pub fn rgba(&mut self) { self.options = self.options.set_out_colorspace(ColorSpace::RGBA); self.stale = true; }
    pub fn set_limits(&mut self, width: u16, height: u16) {
        self.options = self.options.set_max_width(width).set_max_height(height);
        self.stale = true;
    }
    pub fn set_output_colorspace(&mut self, colorspace: ColorSpace) {
        self.options = self.options.set_out_colorspace(colorspace);
        self.stale = true;
    }
    pub fn set_num_threads(&mut self, threads: usize) -> Result<(), DecodeErrors> {
        match std::num::NonZeroU32::new(threads as u32) {
            None => Err(DecodeErrors { status: -1, message: "Cannot set zero threads to decode image".into() }),
            Some(n) => { self.options = self.options.set_num_threads(n); self.stale = true; Ok(()) }
        }
    }
}
impl Drop for Decoder { fn drop(&mut self) { unsafe { zj_decoder_free(self.d) } } }
