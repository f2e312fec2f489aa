// EXPERIMENT (not part of libpysdr_hip.so): the 64k PSD as ONE persistent kernel whose
// 512 KB/frame four-step intermediate stays in the XCD's L2 instead of going through HBM.
// Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o /tmp/psd_fused scripts/experiments/psd_fused.hip
//   /tmp/psd_fused <nframes> <grid> <ring>
// Result on MI355X (round 1): bit-exact against the two-kernel path, but 1.03 ms per 2048
// frames against 0.66 ms: one 512-thread workgroup per CU (204 VGPRs) has nothing to hide the
// load -> FFT -> LDS -> FFT -> store chain behind, and every 32-column / 32-row unit pays four
// serial agent-scope atomics (~1 us each).  Forcing 128 VGPRs (2 workgroups per CU) spills and
// is slower still (1.28-1.5 ms) and showed one silent mismatch at ring = 4.  See DESIGN.md 7.
#define PSD_MARK(slot_, v_) __hip_atomic_store(&q->pad0[(slot_) % 15u], (unsigned)(v_), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)
#include "../../pysdr_amd/csrc/psdfft.hip"
namespace pysdr {
namespace {
// ---- fused form: ONE persistent kernel, the intermediate never leaves the XCD's L2.
// Every workgroup reads the id of the XCD it runs on and pulls tickets from that XCD's
// queue: ticket t -> local frame lf = t/16, unit u = t%16; u < 8 = columns [32u, 32u+32)
// (two 16-column half-units, one per half workgroup), u >= 8 = rows [32(u-8), +32).  The
// frame number itself comes from one global counter (claimed by the holder of unit 0), so
// the split of frames over XCDs follows their actual speed.  Frame lf of an XCD lives in
// slot lf % ring of that XCD's work area; the rows of a frame wait for its 8 column units,
// the columns of frame lf wait for the rows of frame lf - ring.  Every wait is on a SMALLER
// ticket of the same queue, and tickets are only ever held by running workgroups, so the
// scheme cannot deadlock whatever the residency; spins are bounded anyway and raise
// ctl->error instead of hanging the GPU.
struct PsdFusedCtl {
  unsigned gframe;                  // next frame to claim
  unsigned error;
  unsigned pad[14];
  struct Xcd {
    unsigned ticket;
    unsigned pad0[15];
    unsigned long long frame_of[16];   // (lf + 1) << 32 | frame
    unsigned cols_done[16];
    unsigned rows_done[16];
  } xcd[8];
};

__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 7u;
}

constexpr unsigned kSpinLimit = 1u << 14;

__global__ __launch_bounds__(512) void psd_fused_kernel(const float2* __restrict__ x, size_t hop, int nframes,
                                                        const float* __restrict__ win, float2* work,
                                                        float* __restrict__ out, int db, int ring,
                                                        PsdFusedCtl* ctl) {
  __shared__ __attribute__((aligned(16))) float2 lds[kRowLds];
  __shared__ unsigned sh[4];
  static_assert(kRowLds >= 2 * kColLds, "LDS of a rows unit holds two column half-units");
  const int tid = threadIdx.x;
  const unsigned xc = xcc_id();
  PsdFusedCtl::Xcd* q = &ctl->xcd[xc];
  float2* const xwork = work + (size_t)xc * ring * kN;

  // All single-thread work of an iteration sits in ONE block at the top of the loop (publish
  // the previous unit, take the next ticket, wait for its dependencies).  A second
  // `if (tid == 0)` block behind the unit's barrier made hipcc build a loop whose other lanes
  // ran ahead into the next barrier while lane 0 was parked: a livelock.
  unsigned* done = nullptr;      // counter to bump for the unit finished in the last iteration
  bool leaving = false;
  for (;;) {
    if (tid == 0) {
      if (done) atomicAdd(done, 1u);
      unsigned f = 0xffffffffu, u = 0, slot = 0, ok = leaving ? 0u : 1u;
      if (!leaving) {
        const unsigned t = atomicAdd(&q->ticket, 1u);
        const unsigned lf = t >> 4, gen = lf / (unsigned)ring;
        u = t & 15u;
        slot = lf % (unsigned)ring;
        PSD_MARK(t, 1);
        if (u == 0) {
          f = atomicAdd(&ctl->gframe, 1u);
          __hip_atomic_store(&q->frame_of[slot], ((unsigned long long)(lf + 1u) << 32) | f, __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
        } else {
          unsigned n = 0;
          for (;;) {
            const unsigned long long v = __hip_atomic_load(&q->frame_of[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((unsigned)(v >> 32) == lf + 1u) { f = (unsigned)v; break; }
            if (++n > kSpinLimit || __hip_atomic_load(&ctl->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { ok = 0; break; }
            __builtin_amdgcn_s_sleep(8);
          }
        }
        PSD_MARK(t, 2);
        if (ok && f < (unsigned)nframes) {
          // wait for what this unit depends on
          const unsigned* cnt = (u < 8u) ? &q->rows_done[slot] : &q->cols_done[slot];
          const unsigned need = (u < 8u) ? 8u * gen : 8u * (gen + 1u);
          unsigned n = 0;
          while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
            if (++n > kSpinLimit || __hip_atomic_load(&ctl->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { ok = 0; break; }
            __builtin_amdgcn_s_sleep(8);
          }
        }
        if (!ok) atomicCAS(&ctl->error, 0u, 1u + (t << 4) + (f < (unsigned)nframes ? 0u : 8u) + xc);
        PSD_MARK(t, 3);
      }
      sh[0] = f; sh[1] = u; sh[2] = slot; sh[3] = ok;
    }
    __syncthreads();
    // wave-uniform by construction: keep them in SGPRs so the branches below are scalar
    const unsigned f = __builtin_amdgcn_readfirstlane(sh[0]), u = __builtin_amdgcn_readfirstlane(sh[1]);
    const unsigned slot = __builtin_amdgcn_readfirstlane(sh[2]), ok = __builtin_amdgcn_readfirstlane(sh[3]);
    __syncthreads();
    if (!ok) return;
    done = (u < 8u) ? &q->cols_done[slot] : &q->rows_done[slot];
    if (f >= (unsigned)nframes) {
      // past the end: nothing to compute, but later tickets may still count on this unit;
      // one more trip through the top block publishes it, then everybody leaves
      leaving = true;
      continue;
    }
    float2* const yf = xwork + (size_t)slot * kN;
    if (u < 8u) {
      const int half = tid >> 8;
      cols_unit(x + (size_t)f * hop, win, yf, 2 * (int)u + half, tid & 255, lds + half * kColLds);
      // the unit's stores have reached L2 (vmcnt counts them until then) before thread 0
      // publishes the unit at the top of the next iteration
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      rows_unit<true>(yf, out + (size_t)f * kN, (int)u - 8, db, tid, lds);
    }
    __syncthreads();
  }
}




// human-readable dump of a host copy of the control block (diagnostics of a timed-out wait)
static std::string psd_fused_ctl_dump(const void* host_copy) {
  const PsdFusedCtl* c = reinterpret_cast<const PsdFusedCtl*>(host_copy);
  char buf[256];
  std::string out;
  snprintf(buf, sizeof(buf), "gframe=%u error=0x%x;", c->gframe, c->error);
  out += buf;
  for (int x = 0; x < 8; ++x) {
    snprintf(buf, sizeof(buf), " xcd%d: marks=%u%u%u%u%u%u%u%u ticket=%u cols=[%u %u %u %u] rows=[%u %u %u %u] fo0=%llx;", x, c->xcd[x].pad0[0], c->xcd[x].pad0[1], c->xcd[x].pad0[2], c->xcd[x].pad0[3], c->xcd[x].pad0[4], c->xcd[x].pad0[5], c->xcd[x].pad0[6], c->xcd[x].pad0[7], c->xcd[x].ticket,
             c->xcd[x].cols_done[0], c->xcd[x].cols_done[1], c->xcd[x].cols_done[2], c->xcd[x].cols_done[3],
             c->xcd[x].rows_done[0], c->xcd[x].rows_done[1], c->xcd[x].rows_done[2], c->xcd[x].rows_done[3],
             (unsigned long long)c->xcd[x].frame_of[0]);
    out += buf;
  }
  return out;
}

}  // namespace
}  // namespace pysdr
#include <chrono>
#include <thread>
#include <unistd.h>
#include <cstdarg>
namespace pysdr { void set_last_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); } }
using namespace pysdr;

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { fprintf(stderr, "%s -> %s\n", #e, hipGetErrorString(_e)); return 1; } } while (0)
int main(int argc, char** argv) {
  const int nframes = argc > 1 ? atoi(argv[1]) : 64, grid = argc > 2 ? atoi(argv[2]) : 64, ring = argc > 3 ? atoi(argv[3]) : 4;
  float2 *x, *work; float *win, *out, *out2;
  CK(hipMalloc(&x, (size_t)nframes * kM * 8)); CK(hipMalloc(&work, (size_t)std::max(nframes, 8 * ring) * kN * 8));
  CK(hipMalloc(&win, kM * 4)); CK(hipMalloc(&out, (size_t)nframes * kN * 4)); CK(hipMalloc(&out2, (size_t)nframes * kN * 4));
  std::vector<float2> hx((size_t)nframes * kM); std::vector<float> hw(kM);
  unsigned s = 1; for (auto& v : hx) { s = s * 1664525u + 1013904223u; v.x = (float)(s >> 8) / 16777216.f - 0.5f; s = s * 1664525u + 1013904223u; v.y = (float)(s >> 8) / 16777216.f - 0.5f; }
  for (int i = 0; i < kM; ++i) hw[i] = 1.0f / kM;
  CK(hipMemcpy(x, hx.data(), hx.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(win, hw.data(), kM * 4, hipMemcpyHostToDevice));
  hipStream_t st; CK(hipStreamCreate(&st));
  if (launch_psd64k(x, kM, nframes, win, work, out, 1, st)) return 1;
  CK(hipStreamSynchronize(st));
  PsdFusedCtl* ctl; CK(hipHostMalloc(&ctl, sizeof(PsdFusedCtl), hipHostMallocCoherent));
  memset(ctl, 0, sizeof(*ctl));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, st));
  hipLaunchKernelGGL(psd_fused_kernel, dim3(grid), dim3(512), 0, st, x, (size_t)kM, nframes, win, work, out2, 1, ring, ctl);
  CK(hipGetLastError());
  CK(hipEventRecord(e1, st));
  for (int i = 0; i < 40; ++i) {
    if (hipEventQuery(e1) == hipSuccess) break;
    std::this_thread::sleep_for(std::chrono::milliseconds(50));
    if (i % 10 == 9 || i < 3) fprintf(stderr, "[%d] %s\n", i, psd_fused_ctl_dump(ctl).c_str());
    if (i == 30) { fprintf(stderr, "aborting kernel\n"); __atomic_store_n(&ctl->error, 0xdeadu, __ATOMIC_SEQ_CST); }
  }
  hipError_t q = hipEventQuery(e1);
  fprintf(stderr, "final query: %s\n%s\n", hipGetErrorString(q), psd_fused_ctl_dump(ctl).c_str());
  if (q != hipSuccess) { fprintf(stderr, "kernel still running: giving up\n"); _exit(3); }
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<float> a((size_t)nframes * kN), b(a.size());
  CK(hipMemcpy(a.data(), out, a.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), out2, b.size() * 4, hipMemcpyDeviceToHost));
  size_t bad = 0; for (size_t i = 0; i < a.size(); ++i) bad += (a[i] != b[i]);
  printf("fused %.3f ms, mismatches %zu of %zu\n", ms, bad, a.size());
  return 0;
}
