// Minimal repro attempt of the round-4 stale read (round 5 follow-up).
//
// The product symptom: pp_argmax_kernel read values of a 419 MB volume that pp_mask256_kernel had written ONE LAUNCH EARLIER ON THE
// SAME STREAM that pre-dated that write (first megabytes of the buffer, ~25 % of B = 8 forwards), only with the network's other chains
// running on five more streams (six streams on the runtime's four hardware queues).  This program has no torch in it: one stream
// runs   pollute (old content of the buffer's head, written and re-read so that every XCD's L2 / every CU's L1 may hold it)
//        -> producer (fills the whole volume with this iteration's pattern) -> consumer (gathers like the argmax kernel and checks
//        every value against the pattern, plain / sc1 / sc0 sc1 loads)
// while 0..7 other streams replay graphs of streaming kernels.  Per configuration it reports: mismatching reads, what they held (the
// polluting pattern, the previous iteration's pattern, something else), where (offset in the volume), and from device timestamps
// (s_memrealtime) whether any consumer workgroup started before the last producer workgroup ended (= an ordering failure, as opposed
// to a cache-visibility one).
//
//   hipcc --offload-arch=gfx950 -O2 tools/probes/stale_probe.hip -o /tmp/stale_probe && /tmp/stale_probe [iters] [volume MiB]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#define CK(x)                                                                                 \
  do {                                                                                        \
    hipError_t e_ = (x);                                                                      \
    if (e_ != hipSuccess) {                                                                   \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));       \
      exit(2);                                                                                \
    }                                                                                         \
  } while (0)

__host__ __device__ inline uint32_t pat(uint32_t j, uint32_t it, uint32_t tag) {
  uint32_t h = j * 2654435761u ^ (it * 40503u + tag * 0x9e3779b9u);
  h ^= h >> 15;
  h *= 2246822519u;
  h ^= h >> 13;
  return h | 1u;  // never 0
}

__device__ inline uint64_t now() { return wall_clock64(); }  // s_memrealtime: one 100 MHz counter for the whole chip

// old content of the head of the buffer: written ...
__global__ void pollute_write(uint32_t* p, size_t n, uint32_t it) {
  size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n) p[j] = pat((uint32_t)j, it, 0xA);
}
// ... and re-read by OTHER workgroups (a rotated block map), so that lines sit in L1s / L2s that did not write them
__global__ void pollute_read(const uint32_t* p, size_t n, uint32_t rot, uint32_t* sink) {
  size_t b = (blockIdx.x + rot) % gridDim.x;
  size_t j = b * blockDim.x + threadIdx.x;
  uint32_t v = j < n ? p[j] : 0;
  if (v == 0xdeadbeefu) sink[0] = v;
}

__global__ void producer(uint32_t* p, size_t n, uint32_t it, uint64_t* t_end) {
  size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n) p[j] = pat((uint32_t)j, it, 0xB);
  if (threadIdx.x == 0 && (blockIdx.x & 63) == 0) t_end[blockIdx.x >> 6] = now();  // (time of issue of the stores, a lower bound of "done")
}

template <int FL>
__device__ inline uint32_t ld(const uint32_t* p) {
  if (FL == 0) return *p;
  if (FL == 1) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// the argmax kernel's access shape: grid (npix / 256, B); a thread = one output pixel of a (T, H, W) map, reads the four bilinear
// neighbours in the (T, MS, MS, Q) slab of its item for nk of the Q channels
template <int FL>
__global__ __launch_bounds__(256) void consumer(const uint32_t* p, uint32_t it, int T, int H, int W, int MS, int Q, int nk, unsigned long long* stats,
                                                uint32_t* samples, uint64_t* t_start) {
  const int b = blockIdx.y;
  if (threadIdx.x == 0 && (blockIdx.x & 15) == 0) t_start[b * (gridDim.x >> 4) + (blockIdx.x >> 4)] = now();
  const int64_t pix = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t npix = (int64_t)T * H * W;
  if (pix >= npix) return;
  const int x = (int)(pix % W), y = (int)((pix / W) % H), t = (int)(pix / ((int64_t)W * H));
  const int y0 = y * MS / H, x0 = x * MS / W;
  const int y1 = y0 + (y0 < MS - 1), x1 = x0 + (x0 < MS - 1);
  const int64_t base = ((int64_t)b * T + t) * MS * MS;
  for (int k = 0; k < nk; ++k) {
    const int q = (k * 7 + 3) % Q;
    const int64_t idx[4] = {(base + (int64_t)y0 * MS + x0) * Q + q, (base + (int64_t)y0 * MS + x1) * Q + q, (base + (int64_t)y1 * MS + x0) * Q + q,
                            (base + (int64_t)y1 * MS + x1) * Q + q};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const uint32_t v = ld<FL>(p + idx[c]);
      const uint32_t want = pat((uint32_t)idx[c], it, 0xB);
      if (v != want) {
        const int kind = v == pat((uint32_t)idx[c], it, 0xA) ? 1 : (v == pat((uint32_t)idx[c], it - 1, 0xB) ? 2 : (v == pat((uint32_t)idx[c], it - 1, 0xA) ? 3 : 4));
        const unsigned long long n = atomicAdd(&stats[0], 1ull);
        atomicAdd(&stats[kind], 1ull);
        atomicMin(&stats[6], (unsigned long long)idx[c]);
        atomicMax(&stats[7], (unsigned long long)idx[c]);
        if (n < 16) {
          samples[n * 4 + 0] = (uint32_t)idx[c];
          samples[n * 4 + 1] = v;
          samples[n * 4 + 2] = want;
          samples[n * 4 + 3] = (uint32_t)blockIdx.x | ((uint32_t)b << 24);
        }
      }
    }
  }
}

// background load: a streaming read-modify-write over a private buffer
__global__ void bg_kernel(float* a, size_t n, float s) {
  size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; j < n; j += stride) a[j] = a[j] * s + 1.0f;
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 60;
  const size_t vol_mib = argc > 2 ? (size_t)atoi(argv[2]) : 400;
  const int B = 8, T = 2, MS = 256, Q = 100, H = 512, W = 512, NK = 5;
  size_t n = (size_t)B * T * MS * MS * Q;  // 104 857 600 words = 400 MiB
  if (vol_mib != 400) n = vol_mib * 1024 * 1024 / 4 / ((size_t)B * T * Q) * ((size_t)B * T * Q);
  const int ms_eff = vol_mib == 400 ? MS : (int)__builtin_sqrt((double)(n / ((size_t)B * T * Q)));
  const size_t head = std::min<size_t>(n, (size_t)32 * 1024 * 1024 / 4);  // 32 MiB of "old content" at the head of the buffer
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("# %s, %d CUs; volume %.1f MiB, head %zu MiB polluted; %d iterations per configuration; GPU_MAX_HW_QUEUES=%s\n", prop.gcnArchName, prop.multiProcessorCount,
         n * 4.0 / 1048576, head * 4 / 1048576, iters, getenv("GPU_MAX_HW_QUEUES") ? getenv("GPU_MAX_HW_QUEUES") : "(default)");

  uint32_t *P, *sink, *samples;
  unsigned long long* stats;
  uint64_t *t_end, *t_start;
  const unsigned pgrid = (unsigned)((n + 255) / 256);
  const unsigned cgx = (unsigned)(((size_t)T * H * W + 255) / 256);
  CK(hipMalloc(&P, n * 4));
  CK(hipMalloc(&sink, 64));
  CK(hipMalloc(&samples, 16 * 4 * 4));
  CK(hipMalloc(&stats, 8 * 8));
  CK(hipMalloc(&t_end, ((size_t)pgrid / 64 + 1) * 8));
  CK(hipMalloc(&t_start, ((size_t)cgx / 16 + 1) * B * 8));
  const int NBG = 7;
  const size_t bgn = (size_t)64 * 1024 * 1024;  // 256 MiB per background stream
  float* bgbuf[NBG];
  hipStream_t s0, bgs[NBG], other;
  CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&other, hipStreamNonBlocking));
  hipGraphExec_t bgexec[NBG];
  for (int i = 0; i < NBG; ++i) {
    CK(hipMalloc(&bgbuf[i], bgn * 4));
    CK(hipMemset(bgbuf[i], 0, bgn * 4));
    CK(hipStreamCreateWithFlags(&bgs[i], hipStreamNonBlocking));
    hipGraph_t g;
    CK(hipStreamBeginCapture(bgs[i], hipStreamCaptureModeThreadLocal));
    for (int k = 0; k < 12; ++k) hipLaunchKernelGGL(bg_kernel, dim3(256 + 32 * i), dim3(256), 0, bgs[i], bgbuf[i], bgn / 4, 0.5f);
    CK(hipStreamEndCapture(bgs[i], &g));
    CK(hipGraphInstantiate(&bgexec[i], g, nullptr, nullptr, 0));
  }
  CK(hipDeviceSynchronize());
  std::vector<uint64_t> he(pgrid / 64 + 1), hs((size_t)(cgx / 16 + 1) * B);
  uint32_t hsamp[64];
  unsigned long long hst[8];
  hipEvent_t ev;
  CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));

  // configurations: load flavour x background streams x who pollutes (same stream | another stream behind an event)
  for (int pollute_other = 0; pollute_other < 2; ++pollute_other)
    for (int nbg : {0, 3, 5, 7})
      for (int fl = 0; fl < 3; ++fl) {
        unsigned long long tot[8] = {0};
        int bad_iters = 0, order_viol = 0;
        long long min_gap = 1ll << 60;
        uint32_t first_samp[8] = {0};
        for (int it = 1; it <= iters; ++it) {
          CK(hipMemsetAsync(stats, 0, 6 * 8, s0));
          {
            unsigned long long init[2] = {~0ull, 0ull};
            CK(hipMemcpyAsync(stats + 6, init, 16, hipMemcpyHostToDevice, s0));
          }
          for (int i = 0; i < nbg; ++i) CK(hipGraphLaunch(bgexec[i], bgs[i]));
          hipStream_t ps = pollute_other ? other : s0;
          if (pollute_other) {
            CK(hipEventRecord(ev, s0));
            CK(hipStreamWaitEvent(other, ev, 0));
          }
          hipLaunchKernelGGL(pollute_write, dim3((unsigned)((head + 255) / 256)), dim3(256), 0, ps, P, head, (uint32_t)it);
          hipLaunchKernelGGL(pollute_read, dim3((unsigned)((head + 255) / 256)), dim3(256), 0, ps, P, head, (uint32_t)(it * 37 + 5), sink);
          if (pollute_other) {
            CK(hipEventRecord(ev, other));
            CK(hipStreamWaitEvent(s0, ev, 0));
          }
          hipLaunchKernelGGL(producer, dim3(pgrid), dim3(256), 0, s0, P, n, (uint32_t)it, t_end);
          if (fl == 0) hipLaunchKernelGGL(consumer<0>, dim3(cgx, B), dim3(256), 0, s0, P, (uint32_t)it, T, H, W, ms_eff, Q, NK, stats, samples, t_start);
          if (fl == 1) hipLaunchKernelGGL(consumer<1>, dim3(cgx, B), dim3(256), 0, s0, P, (uint32_t)it, T, H, W, ms_eff, Q, NK, stats, samples, t_start);
          if (fl == 2) hipLaunchKernelGGL(consumer<2>, dim3(cgx, B), dim3(256), 0, s0, P, (uint32_t)it, T, H, W, ms_eff, Q, NK, stats, samples, t_start);
          CK(hipMemcpyAsync(hst, stats, 64, hipMemcpyDeviceToHost, s0));
          CK(hipMemcpyAsync(hsamp, samples, 256, hipMemcpyDeviceToHost, s0));
          CK(hipMemcpyAsync(he.data(), t_end, he.size() * 8, hipMemcpyDeviceToHost, s0));
          CK(hipMemcpyAsync(hs.data(), t_start, hs.size() * 8, hipMemcpyDeviceToHost, s0));
          CK(hipStreamSynchronize(s0));
          const uint64_t last_end = *std::max_element(he.begin(), he.begin() + (pgrid + 63) / 64);
          uint64_t first_start = ~0ull;
          for (int b = 0; b < B; ++b)
            for (unsigned k = 0; k < (cgx + 15) / 16; ++k) first_start = std::min(first_start, hs[(size_t)b * (cgx >> 4) + k]);
          const long long gap = (long long)(first_start - last_end);
          min_gap = std::min(min_gap, gap);
          order_viol += gap < 0;
          if (hst[0]) {
            if (!bad_iters) memcpy(first_samp, hsamp, 32);
            ++bad_iters;
            for (int k = 0; k < 6; ++k) tot[k] += hst[k];
            tot[6] = bad_iters == 1 ? hst[6] : std::min(tot[6], hst[6]);
            tot[7] = std::max(tot[7], hst[7]);
          }
        }
        CK(hipDeviceSynchronize());
        printf("pollute=%s bg_streams=%d loads=%-7s: %3d / %d iterations with stale reads; reads stale %llu (pollute pattern %llu, previous iteration %llu, "
               "previous pollute %llu, other %llu); word offsets %llu .. %llu; consumer first start - producer last store issue: min %lld ticks (100 MHz), %d iterations < 0\n",
               pollute_other ? "other-stream" : "same-stream", nbg, fl == 0 ? "plain" : (fl == 1 ? "sc1" : "sc0sc1"), bad_iters, iters, tot[0], tot[1], tot[2], tot[3], tot[4],
               bad_iters ? tot[6] : 0ull, tot[7], min_gap, order_viol);
        if (bad_iters)
          printf("    first samples: idx %u got %08x want %08x (block %u item %u); idx %u got %08x want %08x\n", first_samp[0], first_samp[1], first_samp[2],
                 first_samp[3] & 0xffffff, first_samp[3] >> 24, first_samp[4], first_samp[5], first_samp[6]);
        fflush(stdout);
      }
  return 0;
}
