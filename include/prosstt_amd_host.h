/*
 * prosstt_amd_host.h -- host-side helpers of the drop-in path (libprosstt_amd_host.so: plain C++ threads, no HIP).
 *
 * The device library (prosstt_amd.h) returns counts as the int32 it computes in; the reference returns
 * int64 (/root/reference/prosstt/simulation.py:651: the ndarray that scipy's nbinom(...).rvs() fills).  Widening on the
 * device doubles the bytes that cross PCIe (8 GB for 50 000 x 20 000); these helpers widen on the host's cores instead,
 * a chunk at a time under the transfer of the next (prosstt_amd/device.py, _to_host_widened), so that the copy stays
 * bound by 4 bytes per count.
 */
#ifndef PROSSTT_AMD_HOST_H
#define PROSSTT_AMD_HOST_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* dst[i] = (int64_t)src[i], i < count, on `threads` worker threads (a pool kept by the library; clamped to [1, 64]).
 * The ranges may not overlap.  Returns 0, or -1 for a NULL pointer with count > 0.  Thread-safe (calls are serialised). */
int prosstt_amd_host_widen_i32_i64(const int32_t* src, int64_t* dst, uint64_t count, int32_t threads);

/* The same from a uint16 source: the transfer format of prosstt_amd/device.py -- the low 16 bits of every count, 2 bytes over
 * PCIe; the few counts above 65 535 travel beside them as (position, value) pairs and are written over the widened matrix. */
int prosstt_amd_host_widen_u16_i64(const uint16_t* src, int64_t* dst, uint64_t count, int32_t threads);
int prosstt_amd_host_widen_u16_i32(const uint16_t* src, int32_t* dst, uint64_t count, int32_t threads);

/* ... and from a uint8 source: the low 8 bits of every count, 1 byte over PCIe (two thirds of a count matrix are zeros and
 * one count in a thousand of the headline workload is above 255); the others travel beside them as (position, value) pairs. */
int prosstt_amd_host_widen_u8_i64(const uint8_t* src, int64_t* dst, uint64_t count, int32_t threads);
int prosstt_amd_host_widen_u8_i32(const uint8_t* src, int32_t* dst, uint64_t count, int32_t threads);

/* dst[positions[i]] = values[i], i < count, for an int64 (dst_itemsize 8) or int32 (4) destination: the counts that did not fit
 * the wire, written over the widened matrix.  Positions must be distinct and inside the destination (not checked). */
int prosstt_amd_host_scatter_i32(void* dst, int32_t dst_itemsize, const int64_t* positions, const int32_t* values,
                                 uint64_t count, int32_t threads);

/* 1 if the widening loop runs its AVX2 form on this machine, 0 for the portable loop. */
int prosstt_amd_host_has_avx2(void);

#ifdef __cplusplus
}
#endif
#endif
