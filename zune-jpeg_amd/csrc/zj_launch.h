// zj_launch.h -- launcher prototypes shared by zj_kernels.hip and zj_api.cpp
#pragma once
#include <hip/hip_runtime.h>

#include "zj_device.h"
#include "zj_huff.h"

namespace zj {
hipError_t launch_fused(int hs, int vs, int out, int variant, int fast, const Params& p, hipStream_t s);
bool fused_has_wide(); // built with VARIANTS=all
#if defined(ZJ_ABLATION)
void set_pad_lds(int bytes);
int fused_occupancy_420_rgb(int pad_lds);
#endif
// The rows of a batch's frames that the strips never reach (Q6: zeros in the reference), all frames in ONE launch: a
// memset per frame costs a 60-frame launch of 720-row 4:2:0 frames (45 MCU rows, the odd one dropped) twice its kernel time.
struct ZeroRows {
    uint8_t* out;                 // contiguous / strided batch: first frame, frames `frame_stride` bytes apart; or null:
    long long frame_stride;
    uint64_t fptr[SCATTER_MAX];   // ... the frames' own addresses
    unsigned long long off[3], len[3]; // zj_plan.h: uncovered_ranges
    int nr, nframes;
};
hipError_t launch_zero_rows(const ZeroRows& z, hipStream_t s);
int fused_slots_per_cu(int hs, int vs, int out, int variant, int fast, const Params& p); // 0: unknown
const char* fused_kernel_name(int hs, int vs, int out, int variant, int fast, const Params& p);
hipError_t launch_idct_strip(const int16_t* coeff, const int32_t qt[64], int16_t* out, long long nblocks,
                             long long chunks, long long bpc, long long stride, hipStream_t s);
hipError_t launch_upsample_h(const int16_t* in, long long n, int16_t* out, long long out_len, long long m, hipStream_t s);
hipError_t launch_upsample_v(const int16_t* in, long long stride, int16_t* out, long long out_len, hipStream_t s);
hipError_t launch_rgb16(const int16_t* ycc, uint8_t* out, hipStream_t s);
// b: the working sets of njobs scans; max_nsub: the largest scan's sub-sequence count
hipError_t launch_huff_sync(const HuffBatch& b, int njobs, uint32_t max_nsub, int round, bool periodic, hipStream_t s); // one synchronisation round (+ the periodic-run pass in front of it)
hipError_t launch_huff_finish(const HuffBatch& b, int njobs, uint32_t max_nsub, hipStream_t s);         // prefix sums, write pass, EOI cut
} // namespace zj
