// zj_lab_launch.h -- launchers of the micro-benchmark / lab kernels (libzjlab.so: tools only, never in libzjhip.so)
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace zj {
int ubench2_count();
const char* ubench2_name(int op);
hipError_t launch_ubench2(int op, int* out, int blocks, int iters, int seed, hipStream_t s);
int labmem_count();
const char* labmem_name(int i);
hipError_t launch_labmem(int i, const void* in, void* out, long long bytes, hipStream_t s);
int lab_count();
const char* lab_name(int i);
hipError_t launch_lab(int i, const int32_t qt[3][64], int* out, int blocks, int iters, hipStream_t s);
hipError_t launch_lab_rd_check(int mode, const void* in, uint32_t* sums, long long ntiles, hipStream_t s);
hipError_t launch_ub_clock(unsigned long long* out, int blocks, int iters, hipStream_t s);
struct Params;
hipError_t launch_persist(int mode, const Params& p, int groups, hipStream_t s);   // lab/zj_persist.hip
int persist_occupancy(int mode);
} // namespace zj
