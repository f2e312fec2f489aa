// Raw-sample history of a decimator: new = last hist_len samples of [old | x], + zero the raw-peak buffer the NEXT call
// will accumulate into (api.hip keeps two and flips).  Rounds 1-3 ran this as a one-workgroup launch behind every
// decimator kernel: 4.3-4.9 us of stream time for 8-16 KB.  Round 4: ONE workgroup of the decimator kernel itself does
// it while its first tile's copies are in flight (MixDecArgs / MixMfmaArgs::hist_new); the launch remains for calls
// that start no decimator kernel.
#pragma once
#include "common.h"

namespace pysdr {

__device__ __forceinline__ void roll_history(const float2* __restrict__ x, const float2* __restrict__ hist_old,
                                             float2* __restrict__ hist_new, int hist_len, uint32_t n_total,
                                             unsigned* __restrict__ zero, int zero_n, int tid, int nth) {
  for (int j = tid; j < hist_len; j += nth) {
    const long long rel = (long long)n_total - hist_len + j;
    hist_new[j] = (rel >= 0) ? x[rel] : hist_old[hist_len + rel];
  }
  for (int j = tid; j < zero_n; j += nth) zero[j] = 0u;
}

}  // namespace pysdr
