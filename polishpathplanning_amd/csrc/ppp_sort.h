/* ppp_sort.h -- the one library sort of the engine (rocPRIM radix sort), kept in its own translation unit */
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
/* stable ascending sort of (key, value) pairs on bits [0, end_bit) of the key.  tmp == nullptr: only *tmp_bytes is set. */
hipError_t ppp_sort_pairs_u32(void *tmp, size_t *tmp_bytes, const unsigned *key_in, unsigned *key_out, const int *val_in, int *val_out,
                              size_t n, int end_bit, hipStream_t stream);
