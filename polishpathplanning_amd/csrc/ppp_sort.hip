/* ppp_sort.hip -- rocPRIM device radix sort behind one plain function (voxel_down orders points by voxel id with it) */
#include <cstring>
#include "ppp_sort.h"
#include <rocprim/rocprim.hpp>

hipError_t ppp_sort_pairs_u32(void *tmp, size_t *tmp_bytes, const unsigned *key_in, unsigned *key_out, const int *val_in, int *val_out,
                              size_t n, int end_bit, hipStream_t stream)
{
    return rocprim::radix_sort_pairs(tmp, *tmp_bytes, key_in, key_out, val_in, val_out, n, 0u, (unsigned)end_bit, stream);
}
