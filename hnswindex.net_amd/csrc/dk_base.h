// dk_base.h -- device code, part of device_kernels.h: includes, launch-shape macros, wave-level synchronisation, metric ids.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "device_backend.h"

#ifdef EXP_LAT_REGS // experiment: no occupancy target for the traversal kernels (every register the wave can have: no spills)
#define HNSW_WAVES(x) 1
#else
#define HNSW_WAVES(x) (x)
#endif

#ifndef HNSW_I8_WAVES // waves per SIMD of the int8 search kernels with up to two register sets (build experiment: -DHNSW_I8_WAVES=4 / 6)
#define HNSW_I8_WAVES 5
#endif

namespace hnsw {

// Every block of the kernels below that stage data through LDS is ONE wavefront working on its own job (the latency
// variants add a second wave with a role of its own, which never meets the first at a barrier): what the phases of
// such a wave need between a write and the reads of other lanes is that its own memory operations have completed and
// that the compiler keeps the order -- what __syncthreads() does in front of its s_barrier, without the barrier.
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_s_waitcnt(0); // vmcnt(0) expcnt(0) lgkmcnt(0)
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
}

// LDS ordering inside ONE wave (every traversal block is one wave): the wave's LDS instructions execute in order, so
// all a write-then-read by other lanes needs is that the compiler keeps them in order -- not wave_sync(), whose
// s_waitcnt also drains the vector-memory counter and with it every load still in flight.
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// ------------------------------------------------------------------------------------
// device code
// ------------------------------------------------------------------------------------
// Forms of the two traversal kernels (their fourth template argument).  PLAIN: every launch flag is read at run time.  LAT: the
// latency variant, two waves per job (dk_team.h, dk_pool_top.h).  LEAN: what a launch without visited sets (flags 9: the
// default wherever every adjacency list has at most 64 entries and a row at most 1 KB) needs and nothing else -- "no visited
// set" and "rows of all listed neighbours in one go" are compile-time facts, the code of the other two ways through an
// expansion is not there, and the arguments that only the start and the end of a job use are read from the kernarg segment
// where they are used (kernarg_load below): 316 -> 105 spilled scalars in the int8 two-set form, HALF the vector instructions
// per launch (the spills' v_readlane / v_writelane sat in the expansion loop).  Measured (round 5, profiles/r5_lean_ab.log):
// int8 records 12 500-query launches 3.73 -> 2.93 ms, 65 536-query launches 12.9 -> 9.2 ms; f32 rows within +-1.5 %.
constexpr int kFormPlain = 0, kFormLat = 1, kFormLean = 2;

// A kernel argument read again from the kernarg segment at the point of use (a scalar load that hits the constant cache) instead of
// being carried in SGPRs from the kernel's first instruction to its last: the lean search kernel does this for the two dozen
// arguments that only the start and the end of a job look at (result arrays, job list, scratch of the exact traversal), so that the
// registers belong to the expansion loop.  `volatile`: the load stays where it is written (hoisted out of the job loop it would be the
// long live range again).
template <class T>
__device__ __forceinline__ T kernarg_load(unsigned byte_offset)
{
    typedef const char __attribute__((address_space(4))) *cptr;
    typedef const volatile T __attribute__((address_space(4))) *tptr;
    return *(tptr)((cptr)__builtin_amdgcn_kernarg_segment_ptr() + byte_offset);
}
enum { M_SQ = HNSWDEV_SQ_EUCLID, M_COS = HNSWDEV_COSINE, M_UCOS = HNSWDEV_UCOSINE, M_I8 = HNSWDEV_SQ_EUCLID_I8 };

} // namespace hnsw
