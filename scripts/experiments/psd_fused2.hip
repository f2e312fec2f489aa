// EXPERIMENT (not part of libpysdr_hip.so): the 64k PSD as ONE persistent kernel whose
// 512 KB/frame four-step intermediate is meant to stay in the XCD's L2 instead of going
// through HBM.  Workgroups read HW_REG_XCC_ID and pull tickets from their XCD's queue (ticket
// -> local frame, unit; units 0..7 = 32 columns each, 8..15 = 32 rows each); one 64-bit word per
// (XCD, slot) carries tag | frame | rows done | columns done, so a unit needs one spin-load; the
// publisher of frame lf waits for the rows of lf - ring before it re-labels the slot.  Every
// wait is on a smaller ticket of the same queue, tickets are only held by running workgroups,
// so there is no deadlock whatever the residency (and spins are bounded).  Rows read the
// intermediate with agent-scope loads (miss L1, hit L2).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off [-DNO_PREFETCH] [-DINLINE_UNITS] \
//         -o /tmp/psd_fused2 scripts/experiments/psd_fused2.hip
//   /tmp/psd_fused2 <nframes> <wgs_per_cu> <ring> <reps>
// Results on MI355X (round 1), 2048 frames, bit-exact against the two-kernel path in every run:
//   two-kernel, groups of 256 frames            0.59 ms
//   fused, ticket prefetched one unit ahead     1.7-1.9 ms  (a held ticket delays a unit others wait for)
//   fused, -DNO_PREFETCH                        1.15 ms     (1 or 2 workgroups per CU alike)
// rocprofv3 --pmc says why it cannot win: the fused kernel fetches 740 MB (input 512 MB + 22 %
// of the intermediate: the L2 does serve the re-reads) but WRITES 1.54 GB = output + the whole
// intermediate -- global stores go through the L2 to memory on this part, so only the
// intermediate's read (1 GB of the two-kernel path's 3.1 GB) can be saved, and the two-kernel
// path already gets that read from the Infinity Cache.  Two compiler/ISA lessons are kept in
// DESIGN.md 7 (single-thread work in ONE block per loop trip; non-kernel functions need explicit
// address-space casts or they emit flat_load/flat_store).
#include "../../pysdr_amd/csrc/psdfft.hip"
#include <chrono>
#include <thread>
#include <unistd.h>
#include <cstdarg>
namespace pysdr {
void set_last_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
namespace {

struct Ctl2 {
  unsigned gframe, error, pad[14];
  struct Xcd { unsigned ticket; unsigned pad0[15]; unsigned long long slot[16]; } xcd[8];
};
// slot word: tag (lf+1, 16 bits) << 48 | frame (32 bits) << 16 | rows done << 8 | columns done
__device__ __forceinline__ unsigned w_tag(unsigned long long w) { return (unsigned)(w >> 48); }
__device__ __forceinline__ unsigned w_frame(unsigned long long w) { return (unsigned)(w >> 16); }
__device__ __forceinline__ unsigned w_rows(unsigned long long w) { return (unsigned)(w >> 8) & 255u; }
__device__ __forceinline__ unsigned w_cols(unsigned long long w) { return (unsigned)w & 255u; }

__device__ __forceinline__ unsigned xcc_id2() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 7u;
}

#ifdef INLINE_UNITS
#define UNIT_ATTR __forceinline__
#else
#define UNIT_ATTR __noinline__
#endif
__device__ UNIT_ATTR void cols_unit_ni(const float2* xf, const float* win, float2* yf, int u, int tid) {
  extern __shared__ __attribute__((aligned(16))) float2 lds_dyn[];
  const int half = tid >> 8;
  cols_unit(xf, win, yf, 2 * u + half, tid & 255, lds_dyn + half * kColLds);
}
__device__ UNIT_ATTR void rows_unit_ni(const float2* yf, float* of, int rb, int db, int tid) {
  extern __shared__ __attribute__((aligned(16))) float2 lds_dyn[];
  rows_unit<true>(yf, of, rb, db, tid, lds_dyn);
}

constexpr unsigned kSpin2 = 1u << 16;

#ifdef INLINE_UNITS
#define WPE
#else
#define WPE __attribute__((amdgpu_waves_per_eu(4, 4)))
#endif
__global__ __launch_bounds__(512) WPE
void psd_fused2_kernel(const float2* __restrict__ x, size_t hop, int nframes, const float* __restrict__ win,
                       float2* work, float* __restrict__ out, int db, int ring, Ctl2* ctl) {
  __shared__ unsigned sh[4];
  const int tid = threadIdx.x;
  const unsigned xc = xcc_id2();
  Ctl2::Xcd* q = &ctl->xcd[xc];
  float2* const xwork = work + (size_t)xc * ring * kN;

  unsigned long long* done = nullptr;   // slot word to bump for the unit finished last
  unsigned long long done_inc = 0;
  bool leaving = false, have_next = false;
  unsigned next_t = 0;
  for (;;) {
    if (tid == 0) {
      if (done) atomicAdd(done, done_inc);
      unsigned f = 0xffffffffu, u = 0, slot = 0, ok = 1;
      if (leaving && !have_next) {
        ok = 2;                                       // nothing left to do
      } else {
        const unsigned t = have_next ? next_t : atomicAdd(&q->ticket, 1u);
        have_next = false;
#ifndef NO_PREFETCH
        if (!leaving) { next_t = atomicAdd(&q->ticket, 1u); have_next = true; }
#endif
        const unsigned lf = t >> 4;
        u = t & 15u;
        slot = lf % (unsigned)ring;
        unsigned long long* w = &q->slot[slot];
        const unsigned tag = (lf + 1u) & 0xffffu;
        if (u == 0) {
          f = atomicAdd(&ctl->gframe, 1u);
          if (lf >= (unsigned)ring) {
            const unsigned ptag = (lf + 1u - (unsigned)ring) & 0xffffu;
            unsigned n = 0;
            for (;;) {
              const unsigned long long v = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              if (w_tag(v) == ptag && w_rows(v) == 8u) break;
              if (++n > kSpin2 || __hip_atomic_load(&ctl->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { ok = 0; break; }
              __builtin_amdgcn_s_sleep(4);
            }
          }
          if (ok) __hip_atomic_store(w, ((unsigned long long)tag << 48) | ((unsigned long long)f << 16), __ATOMIC_RELAXED,
                                     __HIP_MEMORY_SCOPE_AGENT);
        } else {
          const unsigned need_cols = (u >= 8u) ? 8u : 0u;
          unsigned n = 0;
          for (;;) {
            const unsigned long long v = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (w_tag(v) == tag && w_cols(v) >= need_cols) { f = w_frame(v); break; }
            if (++n > kSpin2 || __hip_atomic_load(&ctl->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { ok = 0; break; }
            __builtin_amdgcn_s_sleep(4);
          }
        }
        if (!ok) atomicCAS(&ctl->error, 0u, 1u + (t << 4) + xc);
      }
      sh[0] = f; sh[1] = u; sh[2] = slot; sh[3] = ok;
    }
    __syncthreads();
    const unsigned f = __builtin_amdgcn_readfirstlane(sh[0]), u = __builtin_amdgcn_readfirstlane(sh[1]);
    const unsigned slot = __builtin_amdgcn_readfirstlane(sh[2]), ok = __builtin_amdgcn_readfirstlane(sh[3]);
    __syncthreads();
    if (ok != 1u) return;
    done = &q->slot[slot];
    done_inc = (u < 8u) ? 1ull : 256ull;
    if (f >= (unsigned)nframes) {       // past the end: count the unit as done, take no new tickets
      leaving = true;
      continue;
    }
    float2* const yf = xwork + (size_t)slot * kN;
    if (u < 8u) {
      cols_unit_ni(x + (size_t)f * hop, win, yf, (int)u, tid);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      rows_unit_ni(yf, out + (size_t)f * kN, (int)u - 8, db, tid);
    }
    __syncthreads();
  }
}

}  // namespace
}  // namespace pysdr

using namespace pysdr;
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { fprintf(stderr, "%s -> %s\n", #e, hipGetErrorString(_e)); return 1; } } while (0)
int main(int argc, char** argv) {
  const int nframes = argc > 1 ? atoi(argv[1]) : 2048, wpc = argc > 2 ? atoi(argv[2]) : 2, ring = argc > 3 ? atoi(argv[3]) : 6;
  const int reps = argc > 4 ? atoi(argv[4]) : 5;
  int ncu = 256; CK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0));
  const int grid = ncu * wpc;
  float2 *x, *work; float *win, *out, *out2;
  CK(hipMalloc(&x, (size_t)nframes * kM * 8)); CK(hipMalloc(&work, (size_t)256 * kN * 8));
  CK(hipMalloc(&win, kM * 4)); CK(hipMalloc(&out, (size_t)nframes * kN * 4)); CK(hipMalloc(&out2, (size_t)nframes * kN * 4));
  std::vector<float2> hx((size_t)2048 * 64); std::vector<float> hw(kM);
  unsigned s = 1; for (auto& v : hx) { s = s * 1664525u + 1013904223u; v.x = (float)(s >> 8) / 16777216.f - 0.5f; s = s * 1664525u + 1013904223u; v.y = (float)(s >> 8) / 16777216.f - 0.5f; }
  for (int i = 0; i < kM; ++i) hw[i] = (1.0f + 0.5f * sinf(i * 1e-3f)) / kM;
  for (size_t o = 0; o < (size_t)nframes * kM; o += hx.size())
    CK(hipMemcpy(x + o, hx.data(), std::min(hx.size(), (size_t)nframes * kM - o) * 8, hipMemcpyHostToDevice));
  { // make frames differ: scale frame f by (1 + f/nframes) on the host side pattern -- cheap: add the frame index into sample 0
    std::vector<float2> first(nframes);
    for (int f = 0; f < nframes; ++f) first[f] = make_float2(0.25f + f * 1e-3f, -0.125f);
    for (int f = 0; f < nframes; ++f) CK(hipMemcpy(x + (size_t)f * kM + 7, &first[f], 8, hipMemcpyHostToDevice));
  }
  CK(hipMemcpy(win, hw.data(), kM * 4, hipMemcpyHostToDevice));
  hipStream_t st, sa; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9f;
  for (int r = 0; r < reps; ++r) {
    CK(hipEventRecord(e0, st));
    for (int f0 = 0; f0 < nframes; f0 += 256) {
      const int nf = std::min(256, nframes - f0);
      if (launch_psd64k(x + (size_t)f0 * kM, kM, nf, win, work, out + (size_t)f0 * kN, 1, st)) return 1;
    }
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
  }
  printf("two-kernel (groups of 256): %.3f ms per %d frames\n", best, nframes);
  std::vector<float> a((size_t)nframes * kN), b(a.size());
  CK(hipMemcpy(a.data(), out, a.size() * 4, hipMemcpyDeviceToHost));
  Ctl2* ctl; CK(hipMalloc(&ctl, sizeof(Ctl2)));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(psd_fused2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kRowLds * 8));
  best = 1e9f;
  size_t worst_bad = 0;
  for (int r = 0; r < reps; ++r) {
    CK(hipMemsetAsync(ctl, 0, sizeof(Ctl2), st));
    CK(hipMemsetAsync(out2, 0, (size_t)nframes * kN * 4, st));
    CK(hipEventRecord(e0, st));
    hipLaunchKernelGGL(psd_fused2_kernel, dim3(grid), dim3(512), kRowLds * 8, st, x, (size_t)kM, nframes, win, work, out2, 1, ring, ctl);
    CK(hipGetLastError());
    CK(hipEventRecord(e1, st));
    int waited = 0;
    while (hipEventQuery(e1) != hipSuccess) {
      std::this_thread::sleep_for(std::chrono::milliseconds(20));
      if (++waited == 250) { fprintf(stderr, "kernel still running after 5 s: aborting through ctl->error\n"); static unsigned one = 0xdead; (void)hipMemcpyAsync(&ctl->error, &one, 4, hipMemcpyHostToDevice, sa); }
      if (waited > 500) { fprintf(stderr, "giving up\n"); _exit(3); }
    }
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
    Ctl2 h; CK(hipMemcpy(&h, ctl, sizeof(h), hipMemcpyDeviceToHost));
    CK(hipMemcpy(b.data(), out2, b.size() * 4, hipMemcpyDeviceToHost));
    size_t bad = 0; for (size_t i = 0; i < a.size(); ++i) bad += (a[i] != b[i]);
    worst_bad = std::max(worst_bad, bad);
    if (h.error || bad) printf("  rep %d: error=0x%x mismatches=%zu gframe=%u\n", r, h.error, bad, h.gframe);
  }
  printf("fused2 wgs/cu=%d ring=%d: %.3f ms per %d frames, worst mismatches %zu\n", wpc, ring, best, nframes, worst_bad);
  return 0;
}
