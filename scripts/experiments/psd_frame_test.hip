// Test + timing harness of the register-resident 64k PSD (scripts/experiments/psd_frame.hip) against the
// two-kernel four-step path (psdfft.hip) and a double-precision DFT of a few bins.
//   F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off"; C=pysdr_amd/csrc
//   hipcc $F -c $C/psdfft.hip -o /tmp/psdfft.o; hipcc $F -fno-slp-vectorize -c $C/psdreg.hip -o /tmp/psdreg.o
//   hipcc $F -c scripts/experiments/psd_frame_test.hip -o /tmp/t.o; hipcc --offload-arch=gfx950 /tmp/t.o /tmp/psdfft.o /tmp/psdreg.o -o scripts/experiments/psd_frame_test.bin
//   /tmp/psd_frame_test <nframes> <reps>
#include "../../pysdr_amd/csrc/common.h"
namespace pysdr {
int launch_psd64k_frames(const float2* x, size_t hop, int nframes, const float* win, const float2* cwin1,
                         float* out, int db, hipStream_t st);
#ifdef PSD_STAMP
int psd_read_stamps(unsigned long long* out);
#endif
}
#include <cmath>
#include <cstdarg>
#include <vector>
namespace pysdr {
void set_last_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
}
using namespace pysdr;
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { fprintf(stderr, "%s -> %s\n", #e, hipGetErrorString(_e)); return 1; } } while (0)
int main(int argc, char** argv) {
  const int nframes = argc > 1 ? atoi(argv[1]) : 2048, reps = argc > 2 ? atoi(argv[2]) : 5;
  const int kM = 32768, kN = 65536;
  float2 *x, *work, *cwin; float *win, *out, *out2;
  CK(hipMalloc(&x, (size_t)nframes * kM * 8)); CK(hipMalloc(&work, (size_t)448 * kN * 8));
  CK(hipMalloc(&win, kM * 4)); CK(hipMalloc(&cwin, kM * 8));
  CK(hipMalloc(&out, (size_t)nframes * kN * 4)); CK(hipMalloc(&out2, (size_t)nframes * kN * 4));
  std::vector<float2> hx((size_t)kM * 8); std::vector<float> hw(kM); std::vector<float2> hc(kM);
  unsigned s = 1; for (auto& v : hx) { s = s * 1664525u + 1013904223u; v.x = ((float)(s >> 8) / 16777216.f - 0.5f) * 0.01f; s = s * 1664525u + 1013904223u; v.y = ((float)(s >> 8) / 16777216.f - 0.5f) * 0.01f; }
  for (size_t i = 0; i < hx.size(); ++i) { hx[i].x += 0.3f * cosf(0.37f * (float)(i % kM)); hx[i].y += 0.3f * sinf(0.37f * (float)(i % kM)); }
  double wsum = 0; for (int i = 0; i < kM; ++i) { hw[i] = 1.0f + 0.5f * sinf(i * 1e-3f); wsum += hw[i]; }
  for (int i = 0; i < kM; ++i) { hw[i] = (float)(hw[i] / wsum); hc[i] = make_float2((float)(hw[i] * cos(M_PI * i / kM)), (float)(-hw[i] * sin(M_PI * i / kM))); }
  for (size_t o = 0; o < (size_t)nframes * kM; o += hx.size())
    CK(hipMemcpy(x + o, hx.data(), std::min(hx.size(), (size_t)nframes * kM - o) * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(win, hw.data(), kM * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(cwin, hc.data(), kM * 8, hipMemcpyHostToDevice));
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9f;
  for (int r = 0; r < reps; ++r) {
    CK(hipEventRecord(e0, st));
    for (int f0 = 0; f0 < nframes; f0 += 448) {
      const int nf = std::min(448, nframes - f0);
      if (launch_psd64k(x + (size_t)f0 * kM, kM, nf, win, work, out + (size_t)f0 * kN, 1, st)) return 1;
    }
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
  }
  printf("two-kernel (groups of 448): %.3f ms per %d frames = %.1f ns/frame\n", best, nframes, best * 1e6 / nframes);
  best = 1e9f;
  for (int r = 0; r < reps; ++r) {
    CK(hipEventRecord(e0, st));
    if (launch_psd64k_frames(x, kM, nframes, win, cwin, out2, 1, st)) return 1;
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
  }
  const int ncmp = std::min(nframes, 8);
  std::vector<float> a((size_t)ncmp * kN), b(a.size());
  CK(hipMemcpy(a.data(), out, a.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), out2, b.size() * 4, hipMemcpyDeviceToHost));
  double maxlin = 0, errlin = 0, errdb = 0; size_t worst = 0;
  for (size_t i = 0; i < a.size(); ++i) maxlin = std::max(maxlin, pow(10.0, a[i] / 10.0));
  for (size_t i = 0; i < a.size(); ++i) {
    const double la = pow(10.0, a[i] / 10.0), lb = pow(10.0, b[i] / 10.0);
    if (fabs(la - lb) > errlin) { errlin = fabs(la - lb); worst = i; }
    if (a[i] > 10 * log10(maxlin) - 60) errdb = std::max(errdb, (double)fabsf(a[i] - b[i]));
  }
  printf("register-resident: %.3f ms per %d frames = %.1f ns/frame; vs two-kernel: max lin err %.3g of max (bin %zu: %.4f vs %.4f dB), max dB err (top 60 dB) %.3g\n",
         best, nframes, best * 1e6 / nframes, errlin / maxlin, worst % kN, a[worst], b[worst], errdb);
#ifdef PSD_STAMP
  {
    static const char* names[13] = {"start", "x loads", "window", "dft64 #1", "twiddle #1", "exchange 1", "dft64 #2", "twiddle #2",
                                    "exchange 2", "dft8 x8", "power+dB", "stores/p0", "barrier"};
    unsigned long long t[2 * 16 * 8];
    if (psd_read_stamps(t)) return 1;
    for (int e = 0; e < 2; ++e) {
      printf("pass %d (cycles since the pass began; wave 0 / slowest wave / per-phase of wave 0)\n", e);
      for (int i = 1; i <= 12; ++i) {
        unsigned long long w0 = t[(e * 16 + i) * 8] - t[(e * 16) * 8], mx = 0;
        for (int w = 0; w < 8; ++w) mx = std::max(mx, t[(e * 16 + i) * 8 + w] - t[(e * 16) * 8 + w]);
        printf("  %-11s %8llu %8llu   +%llu\n", names[i], w0, mx, t[(e * 16 + i) * 8] - t[(e * 16 + i - 1) * 8]);
      }
    }
    printf("pass 0 start -> pass 1 end: %llu cycles\n", t[(16 + 12) * 8] - t[0]);
  }
#endif
  // a few bins of frame 0 in double precision
  double worst_d = 0;
  for (int K : {0, 1, 2, 3, 777, 32767, 32768, 32769, 36627, 36628, 36629, 65535, 4097, 513, 66}) {
    double re = 0, im = 0;
    for (int n = 0; n < kM; ++n) {
      const double ang = -2.0 * M_PI * (double)(((long long)n * K) % kN) / kN;
      const double xr = hx[n].x * (double)hw[n], xi = hx[n].y * (double)hw[n];
      re += xr * cos(ang) - xi * sin(ang); im += xr * sin(ang) + xi * cos(ang);
    }
    const double pd = 10 * log10(re * re + im * im + 1e-30);
    const float got = b[(K + kM) & (kN - 1)];
    printf("  bin %5d: exact %.5f dB, register-resident %.5f dB, two-kernel %.5f dB\n", K, pd, got, a[(K + kM) & (kN - 1)]);
    if (pd > -80) worst_d = std::max(worst_d, fabs(pd - got));
  }
  printf("max |dB - exact| over the sampled bins above -80 dB: %.3g\n", worst_d);
  return 0;
}
