// Numeric back-end of the waterfall display (three_box_plot.plot, Plotting.py:536-626, and
// shift_waterfall :689-695) -- SURVEY.md 8(f) row N1.  The reference re-allocates the
// [nfft][100] history with np.concatenate on every 20 Hz tick and reduces it on the host;
// here the history is a device ring of columns, a retune is an index offset, and one call
// produces the image that gets blitted:
//   push   wf = concat(wf[:,1:], line)             line shorter than nfft padded with -1e38
//   roll   wf = np.roll(wf, -nbins, axis=0)
//   image  bkgnd = median(mean(wf[:, -cnt:], 1));  zz = wf[0:npsd, :] - bkgnd  (npsd = length of the
//          line just pushed);  img = max(zz, nanmax(zz) - PAN_DR)
#include "common.h"

struct pysdr_waterfall {
  int device = 0, nfft = 0, ncols = 0;
  int head = 0;        // slot that receives the next line (= oldest column)
  int cnt = 0;         // valid columns (wf_cnt, Plotting.py:545-546)
  int shift = 0;       // accumulated retune roll: logical bin i lives at (i + shift) mod nfft
  float* d_wf = nullptr;     // [ncols][nfft]
  float* d_line = nullptr;   // staging for host lines
  float* d_mean = nullptr;   // [nfft]
  float* d_stat = nullptr;   // [0] bkgnd, [1] max(wf)
  float* d_image = nullptr;  // [ncols][nfft]
  int* d_pk = nullptr;       // peak pick scratch: [3][nfft / 2 + 2] positions, states, kept indices; + [1] the count
  hipStream_t stream = nullptr;
};

namespace pysdr {
namespace {

constexpr float kFill = -1.0e38f;    // Plotting.py:385

__global__ __launch_bounds__(256) void wf_fill_kernel(float* p, size_t n, float v) {
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = v;
}

__global__ __launch_bounds__(256) void wf_push_kernel(const float* __restrict__ line, int n, int nfft,
                                                      int shift, float* __restrict__ slot) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= nfft) return;
  int p = i + shift;
  if (p >= nfft) p -= nfft;
  slot[p] = (i < n) ? line[i] : kFill;
}

// mean over the newest `cnt` columns, per logical bin
__global__ __launch_bounds__(256) void wf_mean_kernel(const float* __restrict__ wf, int nfft, int ncols,
                                                      int head, int cnt, int shift,
                                                      float* __restrict__ mean) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= nfft) return;
  int p = i + shift;
  if (p >= nfft) p -= nfft;
  const float inv = 1.0f / (float)cnt;
  float acc = 0.f;
  for (int k = 1; k <= cnt; ++k) {
    int s = head - k;
    if (s < 0) s += ncols;
    acc += wf[(size_t)s * nfft + p] * inv;     // scaled first: a column of -1e38 fills must not overflow
  }
  mean[i] = acc;
}

__device__ __forceinline__ unsigned f2key(float f) {
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key2f(unsigned k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

// exact k-th smallest (0-based) of x[0..n) by 4 passes of 8-bit radix selection; one workgroup
__device__ unsigned select_kth(const float* x, int n, unsigned k, unsigned* hist /*[256] LDS*/) {
  unsigned prefix = 0u, mask = 0u;
  for (int pass = 3; pass >= 0; --pass) {
    const int sh = 8 * pass;
    for (int b = threadIdx.x; b < 256; b += blockDim.x) hist[b] = 0u;
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
      const unsigned key = f2key(x[i]);
      if ((key & mask) == prefix) atomicAdd(&hist[(key >> sh) & 255u], 1u);
    }
    __syncthreads();
    // every thread walks the 256 bins identically (cheap, avoids a broadcast)
    unsigned acc = 0u, bin = 0u;
    for (unsigned b = 0; b < 256u; ++b) {
      const unsigned h = hist[b];
      if (k < acc + h) { bin = b; break; }
      acc += h;
    }
    k -= acc;
    prefix |= bin << sh;
    mask |= 255u << sh;
    __syncthreads();
  }
  return prefix;
}

// np.median: middle element, or the mean of the two middle elements for even n
__global__ __launch_bounds__(1024) void wf_median_kernel(const float* __restrict__ x, int n,
                                                         float* __restrict__ stat) {
  __shared__ unsigned hist[256];
  const unsigned klo = (unsigned)((n - 1) / 2), khi = (unsigned)(n / 2);
  const float a = key2f(select_kth(x, n, klo, hist));
  const float b = (khi == klo) ? a : key2f(select_kth(x, n, khi, hist));
  if (threadIdx.x == 0) stat[0] = 0.5f * (a + b);
}

// max over the logical rows [0, npsd) of every column (Plotting.py:618-619: zz = wf[0:npsd,:] - med,
// zmax = nanmax(zz) -- the rows past the CURRENT line's length are not looked at)
__global__ __launch_bounds__(256) void wf_max_kernel(const float* __restrict__ wf, int nfft, int ncols, int shift,
                                                     int npsd, unsigned* __restrict__ out_key) {
  float m = -3.0e38f;
  const size_t n = (size_t)nfft * ncols;
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    int l = (int)(i % (size_t)nfft) - shift;           // physical bin -> logical row
    if (l < 0) l += nfft;
    if (l < npsd) m = fmaxf(m, wf[i]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) atomicMax(out_key, f2key(m));
}

// image[c][i] = max(wf_L[i][c] - bkgnd, (max(wf) - bkgnd) - pan_dr), c = 0 oldest column
__global__ __launch_bounds__(256) void wf_image_kernel(const float* __restrict__ wf, int nfft, int ncols,
                                                       int head, int shift, const float* __restrict__ stat,
                                                       float pan_dr, float* __restrict__ img) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int c = blockIdx.y;
  if (i >= nfft) return;
  int p = i + shift;
  if (p >= nfft) p -= nfft;
  int s = head + c;
  if (s >= ncols) s -= ncols;
  const float bk = stat[0];
  const float zmax = key2f(reinterpret_cast<const unsigned*>(stat)[1]) - bk;
  img[(size_t)c * nfft + i] = fmaxf(wf[(size_t)s * nfft + p] - bk, zmax - pan_dr);
}

// ---- scipy.signal.find_peaks(x, height = h, distance = d) on the device (Plotting.py:594-602: the peak pick on the averaged
// PSD), ONE workgroup.  The three steps of SciPy's implementation, each in its parallel form:
//  (1) local maxima with flat tops (_local_maxima_1d): a peak is a FALL x[i] < x[i-1] whose previous change point p (largest
//      p < i with x[p] != x[p-1]) was a RISE; the plateau is [p, i-1], the peak its midpoint (p + i - 1) / 2 (integer
//      division).  Flat stretches touching either end of the line have no rise / no fall and are no peaks, as there.  The
//      previous change point is an exclusive prefix maximum: per-thread chunks, a scan of the 1024 chunk results in LDS.
//  (2) height: x[peak] >= h, compared in double (h = bkgnd + 10 is a double there).
//  (3) distance (_select_by_peak_distance): greedily by priority = height, every kept peak removes all peaks closer than
//      ceil(d).  Rounds: an undecided peak that outranks every not-yet-removed peak closer than d is KEPT (the greedy order
//      would reach it before anything could remove it); then every undecided peak closer than d to a kept one is REMOVED;
//      until nothing is undecided.  Same result as the sequential greedy walk whenever the heights differ.  TIES: SciPy
//      ranks equal heights by an UNSTABLE np.argsort, i.e. which of two equal peaks closer than d survives there depends
//      on NumPy's sort of the day; here the one with the higher index outranks (what a stable sort would give).
__global__ __launch_bounds__(1024) void wf_peaks_kernel(const float* __restrict__ x, int n, double height, int dist,
                                                        int* __restrict__ pos, int* __restrict__ state,
                                                        int* __restrict__ kept, int* __restrict__ count) {
  __shared__ int sh[1024];
  __shared__ int sh2[1024];
  __shared__ int flag;
  const int t = threadIdx.x;
  const int C = (n + 1023) / 1024;
  const int lo = t * C, hi = (lo + C < n) ? lo + C : n;
  // (1a) last change point of every chunk, exclusive prefix maximum over the chunks
  int last = -1;
  for (int i = (lo > 1 ? lo : 1); i < hi; ++i)
    if (x[i] != x[i - 1]) last = i;
  sh[t] = last;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {
    const int v = (t >= o) ? sh[t - o] : -1;
    __syncthreads();
    sh[t] = max(sh[t], v);
    __syncthreads();
  }
  const int carry = (t > 0) ? sh[t - 1] : -1;
  __syncthreads();
  // (1b + 2) this chunk's peaks: counted, then (with the chunks' prefix sum) written in order of position
  auto walk = [&](int* out) {
    int prev = carry, cnt = 0;
    for (int i = (lo > 1 ? lo : 1); i < hi; ++i) {
      const float a = x[i - 1], b = x[i];
      if (a == b) continue;
      if (b < a && prev >= 1 && x[prev] > x[prev - 1]) {
        const int mid = (prev + i - 1) >> 1;
        if ((double)x[mid] >= height) {
          if (out) out[cnt] = mid;
          ++cnt;
        }
      }
      prev = i;
    }
    return cnt;
  };
  const int mine = walk(nullptr);
  sh2[t] = mine;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {
    const int v = (t >= o) ? sh2[t - o] : 0;
    __syncthreads();
    sh2[t] += v;
    __syncthreads();
  }
  const int P = sh2[1023];
  (void)walk(pos + (sh2[t] - mine));
  for (int j = t; j < P; j += 1024) state[j] = 0;
  __threadfence_block();
  __syncthreads();
  // (3) the distance rule in rounds; 0 undecided, 1 kept, 2 removed
  auto outranks = [&](int k, int j) {             // peak k before peak j in the greedy order
    const float a = x[pos[k]], b = x[pos[j]];
    return a > b || (a == b && k > j);
  };
  if (dist > 1) {
    for (;;) {
      if (t == 0) flag = 0;
      __syncthreads();
      for (int j = t; j < P; j += 1024) {
        if (state[j] != 0) continue;
        bool best = true;
        for (int k = j - 1; best && k >= 0 && pos[j] - pos[k] < dist; --k)
          if (((volatile int*)state)[k] != 2 && outranks(k, j)) best = false;
        for (int k = j + 1; best && k < P && pos[k] - pos[j] < dist; ++k)
          if (((volatile int*)state)[k] != 2 && outranks(k, j)) best = false;
        if (best) ((volatile int*)state)[j] = 1;
      }
      __threadfence_block();
      __syncthreads();
      for (int j = t; j < P; j += 1024) {
        if (state[j] != 0) continue;
        bool gone = false;
        for (int k = j - 1; !gone && k >= 0 && pos[j] - pos[k] < dist; --k) gone = (state[k] == 1);
        for (int k = j + 1; !gone && k < P && pos[k] - pos[j] < dist; ++k) gone = (state[k] == 1);
        if (gone) state[j] = 2;
        else flag = 1;                             // still undecided: another round
      }
      __threadfence_block();
      __syncthreads();
      const int again = flag;
      __syncthreads();
      if (!again) break;
    }
  } else {
    for (int j = t; j < P; j += 1024) state[j] = 1;
    __threadfence_block();
    __syncthreads();
  }
  // kept peaks, in order of position
  const int PC = (P + 1023) / 1024;
  const int plo = t * PC, phi = (plo + PC < P) ? plo + PC : P;
  int kc = 0;
  for (int j = plo; j < phi; ++j) kc += (state[j] == 1);
  sh[t] = kc;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {
    const int v = (t >= o) ? sh[t - o] : 0;
    __syncthreads();
    sh[t] += v;
    __syncthreads();
  }
  int w = sh[t] - kc;
  for (int j = plo; j < phi; ++j)
    if (state[j] == 1) kept[w++] = pos[j];
  if (t == 0) *count = sh[1023];
}

}  // namespace
}  // namespace pysdr

using namespace pysdr;

extern "C" {

int pysdr_waterfall_create(int device, int nfft, int ncols, pysdr_waterfall** out) {
  if (!out || nfft < 2 || ncols < 1) return PYSDR_ERR_ARG;
  hipError_t e0 = hipSetDevice(device);
  if (e0 != hipSuccess) { set_last_error("hipSetDevice(%d): %s", device, hipGetErrorString(e0)); return PYSDR_ERR_NO_DEVICE; }
  pysdr_waterfall* w = new pysdr_waterfall();
  w->device = device; w->nfft = nfft; w->ncols = ncols;
  const size_t n = (size_t)nfft * ncols;
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { set_last_error("pysdr_waterfall_create: %s -> %s", #e, hipGetErrorString(_e)); pysdr_waterfall_destroy(w); return PYSDR_ERR_HIP; } } while (0)
  CK(hipStreamCreateWithFlags(&w->stream, hipStreamNonBlocking));
  CK(hipMalloc(&w->d_wf, n * sizeof(float)));
  CK(hipMalloc(&w->d_image, n * sizeof(float)));
  CK(hipMalloc(&w->d_line, (size_t)nfft * sizeof(float)));
  CK(hipMalloc(&w->d_mean, (size_t)nfft * sizeof(float)));
  CK(hipMalloc(&w->d_stat, 4 * sizeof(float)));
  CK(hipMalloc(&w->d_pk, (3 * ((size_t)nfft / 2 + 2) + 1) * sizeof(int)));
#undef CK
  hipLaunchKernelGGL(wf_fill_kernel, dim3(1024), dim3(256), 0, w->stream, w->d_wf, n, kFill);
  if (hipStreamSynchronize(w->stream) != hipSuccess) { pysdr_waterfall_destroy(w); return PYSDR_ERR_HIP; }
  *out = w;
  return PYSDR_OK;
}

void pysdr_waterfall_destroy(pysdr_waterfall* w) {
  if (!w) return;
  (void)hipSetDevice(w->device);
  if (w->stream) (void)hipStreamSynchronize(w->stream);
  if (w->d_wf) (void)hipFree(w->d_wf);
  if (w->d_image) (void)hipFree(w->d_image);
  if (w->d_line) (void)hipFree(w->d_line);
  if (w->d_mean) (void)hipFree(w->d_mean);
  if (w->d_stat) (void)hipFree(w->d_stat);
  if (w->d_pk) (void)hipFree(w->d_pk);
  if (w->stream) (void)hipStreamDestroy(w->stream);
  delete w;
}

int pysdr_waterfall_push(pysdr_waterfall* w, const float* line, int n, int on_device) {
  if (!w || !line || n < 0 || n > w->nfft) return PYSDR_ERR_ARG;
  PYSDR_HIP_CHECK(hipSetDevice(w->device));
  const float* src = line;
  if (!on_device) {
    PYSDR_HIP_CHECK(hipMemcpyAsync(w->d_line, line, (size_t)n * sizeof(float), hipMemcpyHostToDevice, w->stream));
    src = w->d_line;
  }
  hipLaunchKernelGGL(wf_push_kernel, dim3((w->nfft + 255) / 256), dim3(256), 0, w->stream, src, n, w->nfft,
                     w->shift, w->d_wf + (size_t)w->head * w->nfft);
  PYSDR_HIP_CHECK(hipGetLastError());
  if (!on_device) PYSDR_HIP_CHECK(hipStreamSynchronize(w->stream));   // the caller may reuse `line`
  w->head = (w->head + 1) % w->ncols;
  if (w->cnt < w->ncols) w->cnt++;
  return PYSDR_OK;
}

int pysdr_waterfall_roll(pysdr_waterfall* w, int nbins) {
  if (!w) return PYSDR_ERR_ARG;
  long s = ((long)w->shift + nbins) % w->nfft;
  if (s < 0) s += w->nfft;
  w->shift = (int)s;
  return PYSDR_OK;
}

int pysdr_waterfall_image_rows(pysdr_waterfall* w, float pan_dr, int npsd, float* image_out, float* mean_out,
                               float* bkgnd_out) {
  if (!w || npsd < 1 || npsd > w->nfft) return PYSDR_ERR_ARG;
  if (w->cnt < 1) { set_last_error("pysdr_waterfall_image: no line pushed yet"); return PYSDR_ERR_STATE; }
  PYSDR_HIP_CHECK(hipSetDevice(w->device));
  const int gx = (w->nfft + 255) / 256;
  const size_t n = (size_t)w->nfft * w->ncols;
  hipLaunchKernelGGL(wf_mean_kernel, dim3(gx), dim3(256), 0, w->stream, w->d_wf, w->nfft, w->ncols, w->head,
                     w->cnt, w->shift, w->d_mean);
  hipLaunchKernelGGL(wf_median_kernel, dim3(1), dim3(1024), 0, w->stream, w->d_mean, w->nfft, w->d_stat);
  PYSDR_HIP_CHECK(hipMemsetAsync(w->d_stat + 1, 0, sizeof(float), w->stream));
  hipLaunchKernelGGL(wf_max_kernel, dim3(512), dim3(256), 0, w->stream, w->d_wf, w->nfft, w->ncols, w->shift, npsd,
                     reinterpret_cast<unsigned*>(w->d_stat + 1));
  hipLaunchKernelGGL(wf_image_kernel, dim3(gx, w->ncols), dim3(256), 0, w->stream, w->d_wf, w->nfft, w->ncols,
                     w->head, w->shift, w->d_stat, pan_dr, w->d_image);
  PYSDR_HIP_CHECK(hipGetLastError());
  if (image_out) PYSDR_HIP_CHECK(hipMemcpyAsync(image_out, w->d_image, n * sizeof(float), hipMemcpyDeviceToHost, w->stream));
  if (mean_out) PYSDR_HIP_CHECK(hipMemcpyAsync(mean_out, w->d_mean, (size_t)w->nfft * sizeof(float), hipMemcpyDeviceToHost, w->stream));
  if (bkgnd_out) PYSDR_HIP_CHECK(hipMemcpyAsync(bkgnd_out, w->d_stat, sizeof(float), hipMemcpyDeviceToHost, w->stream));
  PYSDR_HIP_CHECK(hipStreamSynchronize(w->stream));
  return PYSDR_OK;
}

int pysdr_waterfall_peaks(pysdr_waterfall* w, const float* line, int n, double height, int distance, int* idx_out, int cap,
                          int* n_out) {
  if (!w || !n_out || n < 0 || n > w->nfft || distance < 1 || cap < 0 || (cap > 0 && !idx_out)) return PYSDR_ERR_ARG;
  PYSDR_HIP_CHECK(hipSetDevice(w->device));
  const float* x = w->d_mean;
  if (line) {
    PYSDR_HIP_CHECK(hipMemcpyAsync(w->d_line, line, (size_t)n * sizeof(float), hipMemcpyHostToDevice, w->stream));
    x = w->d_line;
  } else if (w->cnt < 1) {
    set_last_error("pysdr_waterfall_peaks: no averaged line yet (pysdr_waterfall_image first, or pass a line)");
    return PYSDR_ERR_STATE;
  }
  const size_t half = (size_t)w->nfft / 2 + 2;
  int* pos = w->d_pk, *state = w->d_pk + half, *kept = w->d_pk + 2 * half, *count = w->d_pk + 3 * half;
  hipLaunchKernelGGL(wf_peaks_kernel, dim3(1), dim3(1024), 0, w->stream, x, n, height, distance, pos, state, kept, count);
  PYSDR_HIP_CHECK(hipGetLastError());
  int np_ = 0;
  PYSDR_HIP_CHECK(hipMemcpyAsync(&np_, count, sizeof(int), hipMemcpyDeviceToHost, w->stream));
  PYSDR_HIP_CHECK(hipStreamSynchronize(w->stream));
  *n_out = np_;
  const int m = np_ < cap ? np_ : cap;
  if (m > 0) {
    PYSDR_HIP_CHECK(hipMemcpyAsync(idx_out, kept, (size_t)m * sizeof(int), hipMemcpyDeviceToHost, w->stream));
    PYSDR_HIP_CHECK(hipStreamSynchronize(w->stream));
  }
  return PYSDR_OK;
}

int pysdr_waterfall_image(pysdr_waterfall* w, float pan_dr, float* image_out, float* mean_out,
                          float* bkgnd_out) {
  if (!w) return PYSDR_ERR_ARG;
  return pysdr_waterfall_image_rows(w, pan_dr, w->nfft, image_out, mean_out, bkgnd_out);
}

}  // extern "C"
