// sort_util.hip — a STABLE device sort of (32-bit key, index) pairs (hipCUB's LSD radix sort), kept in its own translation unit so
// that the template instantiation costs one object file.  Used where an order inside equal keys has to be the INPUT order: the
// voxel grid accumulates each voxel's centroid in float over its points in ascending input index (cloud_kernels.hip), which is what
// makes the filter's output independent of scheduling and equal to the sequential restatement bit for bit.
#include <hipcub/hipcub.hpp>

#include "lvf_internal.hpp"

namespace lvf {

int device_sort_pairs_u32(lvf_ctx* ctx, const unsigned* keys_in, unsigned* keys_out, const int* vals_in, int* vals_out, int n, int key_bits) {
  if (n <= 0) return LVF_OK;
  hipStream_t s = ctx->stream;
  size_t bytes = 0;
  LVF_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, keys_in, keys_out, vals_in, vals_out, n, 0, key_bits, s));
  DevBuf<unsigned char> tmp;
  LVF_TRY(tmp.alloc(bytes ? bytes : 1));
  LVF_HIP(hipcub::DeviceRadixSort::SortPairs(tmp.p, bytes, keys_in, keys_out, vals_in, vals_out, n, 0, key_bits, s));
  LVF_HIP(hipStreamSynchronize(s));          // (tmp is released on return; the pool hands it out again in stream order, so this wait is only belt and braces)
  return LVF_OK;
}

}  // namespace lvf
